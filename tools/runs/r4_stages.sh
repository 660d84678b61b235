cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r4
{ python tools/stage_times.py 4 1080 1920 fp32; python tools/stage_times.py 16 256 256 fp32 1 20; python tools/stage_times.py 16 256 256 bf16 1 20; python tools/stage_times.py 1 1080 1920 bf16 1 20; python tools/stage_times.py 1 256 256 bf16 1 50; python tools/stage_times.py 1 256 256 fp32 1 50; python tools/stage_times.py 4 1080 1920 fp32 3 3; } > gpurun_out/r4/stage_times.txt 2>&1
cat gpurun_out/r4/stage_times.txt | grep -v amdgpu.ids
