cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r4
timeout -k 10 500 python tools/ab_bench.py pre=abl/lib_pre_x2.so now=default --rounds 3 --steps 10 -- --video-frames 0 --no-fp32 > gpurun_out/r4/ab_pre_x2.txt 2>&1
tail -24 gpurun_out/r4/ab_pre_x2.txt
