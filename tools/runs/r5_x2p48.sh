set -o pipefail
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r5
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r5/x2p48_tests.log 2>&1 || { tail -40 gpurun_out/r5/x2p48_tests.log; exit 1; }
tail -3 gpurun_out/r5/x2p48_tests.log
timeout -k 10 600 python tools/ab_x2.py p40=ablibs/x2_fused_stem_p40.so p48=default --rounds 3 > gpurun_out/r5/ab_x2_p48_only.txt 2>&1 || { cat gpurun_out/r5/ab_x2_p48_only.txt; exit 1; }
cat gpurun_out/r5/ab_x2_p48_only.txt
timeout -k 10 600 python tools/ab_x2.py p40=ablibs/x2_fused_stem_p40.so p48=default --rounds 2 --shape 16 256 256 > gpurun_out/r5/ab_x2_p48_only_cfg2.txt 2>&1 || { cat gpurun_out/r5/ab_x2_p48_only_cfg2.txt; exit 1; }
head -6 gpurun_out/r5/ab_x2_p48_only_cfg2.txt
