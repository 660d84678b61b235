set -o pipefail
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r6d
O=gpurun_out/r6d
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -x -q -m gpu > $O/tests_a.log 2>&1 || { tail -40 $O/tests_a.log; exit 1; }
tail -3 $O/tests_a.log
python tools/latency.py > $O/latency.txt 2>&1 || { tail $O/latency.txt; exit 1; }
cat $O/latency.txt
timeout -k 10 300 python tools/cfg_sweep.py 1 256 256 bf16 > $O/cfg_sweep_b1_256_bf16.txt 2>&1 || { tail $O/cfg_sweep_b1_256_bf16.txt; exit 1; }
timeout -k 10 300 python tools/cfg_sweep.py 1 256 256 fp32 10 > $O/cfg_sweep_b1_256_fp32.txt 2>&1 || { tail $O/cfg_sweep_b1_256_fp32.txt; exit 1; }
timeout -k 10 600 python tools/ab_bench.py base=ablibs/lib_base.so no_wstream=ablibs/lib_no_wstream.so --rounds 2 --steps 10 -- --video-frames 0 --no-fp32 --no-tile4k > $O/ab_no_wstream.txt 2>&1 || { tail $O/ab_no_wstream.txt; exit 1; }
cat $O/ab_no_wstream.txt
