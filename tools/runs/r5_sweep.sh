set -o pipefail
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r5
timeout -k 10 1000 python tools/shape_sweep.py > gpurun_out/r5/shape_sweep.txt 2>&1 || { tail -20 gpurun_out/r5/shape_sweep.txt; exit 1; }
tail -5 gpurun_out/r5/shape_sweep.txt; grep -c FAIL gpurun_out/r5/shape_sweep.txt || true
