cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r4
python tools/gap_probe.py 2>&1 | grep -v amdgpu | tee gpurun_out/r4/gap_probe.txt
