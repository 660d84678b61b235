set -o pipefail
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r4
O=$PWD/gpurun_out/r4
timeout -k 10 500 python tools/ab_bench.py noskip=abl/lib_noskip.so skip=default --rounds 4 --steps 10 -- --video-frames 0 --no-fp32 > $O/ab_fragskip.txt 2>&1
echo "ab rc $?" >> $O/ab_fragskip.txt
python -m pytest tests -x -q -m gpu > $O/gpu_tests3.log 2>&1; echo "pytest rc $?" >> $O/gpu_tests3.log
tail -28 $O/ab_fragskip.txt; tail -4 $O/gpu_tests3.log
