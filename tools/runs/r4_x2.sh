cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r4
timeout -k 10 500 python tools/x2_check.py > gpurun_out/r4/x2_check.txt 2>&1; echo "rc $?"; grep -v amdgpu gpurun_out/r4/x2_check.txt | tail -24
