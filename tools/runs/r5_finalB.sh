set -o pipefail
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r5f
bash tools/profile_all.sh r5final > gpurun_out/r5f/profile_all.log 2>&1 || { tail -20 gpurun_out/r5f/profile_all.log; exit 1; }
tail -6 gpurun_out/r5f/profile_all.log
bash tools/runs/r5_x2prof.sh > gpurun_out/r5f/x2prof.log 2>&1 || { tail -20 gpurun_out/r5f/x2prof.log; exit 1; }
tail -22 gpurun_out/r5f/x2prof.log
