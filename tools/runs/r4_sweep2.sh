cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r4
timeout -k 10 500 python tools/shape_sweep.py 60 > gpurun_out/r4/shape_sweep2.txt 2>&1; echo "sweep rc $?"; tail -4 gpurun_out/r4/shape_sweep2.txt
