cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r4
timeout -k 10 600 python -m pytest tests/test_gpu_configs.py -q -m gpu -k "convtranspose" > gpurun_out/r4/gpu_tests_ct.log 2>&1; echo "pytest rc $?"; tail -3 gpurun_out/r4/gpu_tests_ct.log
python tools/stage_times.py 8 1080 1920 bf16 1 5 convt > gpurun_out/r4/stage_times_convt.txt 2>&1; grep -v amdgpu gpurun_out/r4/stage_times_convt.txt | head -20
