set -o pipefail
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r6e
O=gpurun_out/r6e
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > $O/tests_a.log 2>&1 || { tail -40 $O/tests_a.log; exit 1; }
tail -2 $O/tests_a.log
timeout -k 10 300 python tools/cfg_sweep.py 1 256 256 bf16 > $O/cfg_sweep_b1_256_bf16.txt 2>&1 || { tail $O/cfg_sweep_b1_256_bf16.txt; exit 1; }
timeout -k 10 300 python tools/cfg_sweep.py 1 256 256 fp32 10 > $O/cfg_sweep_b1_256_fp32.txt 2>&1 || { tail $O/cfg_sweep_b1_256_fp32.txt; exit 1; }
python tools/latency.py > $O/latency.txt 2>&1 || { tail $O/latency.txt; exit 1; }
cat $O/latency.txt
