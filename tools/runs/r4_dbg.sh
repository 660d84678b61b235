cd $GRAFT_REPO_ROOT && python bench.py --no-cpu-baseline --video-frames 0 --no-fp32 --no-power 2>&1 | grep -v amdgpu | tail -8 | cut -c1-400
