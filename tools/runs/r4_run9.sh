cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r4
timeout -k 10 300 python tools/power_sample.py --steps 800 > gpurun_out/r4/power_sample.txt 2>&1
cat gpurun_out/r4/power_sample.txt
