set -o pipefail
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r5
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -x -q -m gpu > gpurun_out/r5/p48_tests.log 2>&1 || { tail -30 gpurun_out/r5/p48_tests.log; exit 1; }
tail -4 gpurun_out/r5/p48_tests.log
timeout -k 10 600 python tools/ab_x2.py p40=ablibs/pitch40.so p48=default --rounds 3 --shape 8 1080 1920 --precision bf16 > gpurun_out/r5/ab_p48_bf16.txt 2>&1 || { cat gpurun_out/r5/ab_p48_bf16.txt; exit 1; }
cat gpurun_out/r5/ab_p48_bf16.txt
timeout -k 10 600 python tools/ab_x2.py p40=ablibs/pitch40.so p48=default --rounds 3 > gpurun_out/r5/ab_p48_x2.txt 2>&1 || exit 1
cat gpurun_out/r5/ab_p48_x2.txt
