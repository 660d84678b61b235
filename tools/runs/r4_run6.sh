set -o pipefail
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r4
O=$PWD/gpurun_out/r4
timeout -k 10 800 python tools/ab_bench.py base=abl/lib_base.so o1=abl/lib_o1.so o2=abl/lib_o2.so o3=abl/lib_o3.so o4=abl/lib_o4.so --rounds 3 --steps 10 -- --video-frames 0 --no-fp32 > $O/ab_mma_order.txt 2>&1
echo "ab rc $?" >> $O/ab_mma_order.txt
tail -30 $O/ab_mma_order.txt
