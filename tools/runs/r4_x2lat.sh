cd $GRAFT_REPO_ROOT
for p in fp32 bf16x2 bf16; do python tools/stage_times.py 1 256 256 $p 1 50 2>&1 | grep -v amdgpu | head -1; done
for p in fp32 bf16x2 bf16; do python tools/stage_times.py 1 1080 1920 $p 1 10 2>&1 | grep -v amdgpu | head -1; done
