cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r4
timeout -k 10 500 python tools/shape_sweep.py 60 > gpurun_out/r4/shape_sweep.txt 2>&1; echo "sweep rc $?"
timeout -k 10 300 python tools/stress.py 20 > gpurun_out/r4/stress.txt 2>&1; echo "stress rc $?"
tail -4 gpurun_out/r4/shape_sweep.txt; tail -2 gpurun_out/r4/stress.txt
