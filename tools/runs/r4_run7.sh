set -o pipefail
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r4
O=$PWD/gpurun_out/r4
timeout -k 10 600 python tools/ab_bench.py two=abl/lib_ldspad.so one=abl/lib_ldspad.so,FIUNET_LDS_PAD=8192 --rounds 2 --steps 10 -- --video-frames 0 --no-fp32 > $O/ab_one_wg_per_cu.txt 2>&1
echo "ab rc $?" >> $O/ab_one_wg_per_cu.txt
tail -26 $O/ab_one_wg_per_cu.txt
