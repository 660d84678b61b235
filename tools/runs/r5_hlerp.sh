cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r5
timeout -k 10 600 python tools/ab_x2.py shipped=default no_hlerp=ablibs/no_hlerp.so --rounds 3 --shape 8 1080 1920 --precision bf16 > gpurun_out/r5/ab_no_hlerp.txt 2>&1
cat gpurun_out/r5/ab_no_hlerp.txt
