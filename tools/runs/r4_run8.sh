set -o pipefail
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r4
O=$PWD/gpurun_out/r4
L=abl/lib_ldspad.so
timeout -k 10 900 python tools/ab_bench.py p0=$L p1k=$L,FIUNET_LDS_PAD=1024 p2k=$L,FIUNET_LDS_PAD=2048 p3k=$L,FIUNET_LDS_PAD=3072 p4k=$L,FIUNET_LDS_PAD=4096 p5k=$L,FIUNET_LDS_PAD=5120 p6k=$L,FIUNET_LDS_PAD=6144 p7k=$L,FIUNET_LDS_PAD=7168 --rounds 2 --steps 10 -- --video-frames 0 --no-fp32 > $O/ab_lds_pad_sweep.txt 2>&1
echo "ab rc $?" >> $O/ab_lds_pad_sweep.txt
tail -30 $O/ab_lds_pad_sweep.txt
