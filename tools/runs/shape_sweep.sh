set -o pipefail
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/sweep
timeout -k 10 1100 python tools/shape_sweep.py 40 > gpurun_out/sweep/shape_sweep.txt 2>&1 || { tail -5 gpurun_out/sweep/shape_sweep.txt; exit 1; }
tail -3 gpurun_out/sweep/shape_sweep.txt
