cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r4
python tools/stage_times.py 4 1080 1920 bf16x2 1 5 > gpurun_out/r4/stage_times_x2.txt 2>&1; python tools/stage_times.py 4 1080 1920 bf16 1 5 >> gpurun_out/r4/stage_times_x2.txt 2>&1
grep -v amdgpu gpurun_out/r4/stage_times_x2.txt
