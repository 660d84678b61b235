set -o pipefail
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r4
O=$PWD/gpurun_out/r4
timeout -k 10 600 python tools/ab_bench.py base=abl/lib_base.so noskip=abl/lib_noskip.so skip=default --rounds 3 --steps 10 -- --video-frames 0 --no-fp32 > $O/ab_fragskip2.txt 2>&1
echo "ab rc $?" >> $O/ab_fragskip2.txt
tail -28 $O/ab_fragskip2.txt
