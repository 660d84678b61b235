cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r5
timeout -k 10 600 python tools/ab_x2.py $AB_ARGS > gpurun_out/r5/$AB_OUT 2>&1
tail -40 gpurun_out/r5/$AB_OUT
