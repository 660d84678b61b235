cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r4
python -m pytest tests -x -q -m gpu > gpurun_out/r4/gpu_tests6.log 2>&1; echo "pytest rc $?"; tail -3 gpurun_out/r4/gpu_tests6.log
{ python tools/stage_times.py 8 1080 1920 bf16 1 5 convt; python tools/stage_times.py 2 1080 1920 fp32 1 3 convt; } > gpurun_out/r4/stage_times_convt.txt 2>&1; grep -v amdgpu.ids gpurun_out/r4/stage_times_convt.txt
