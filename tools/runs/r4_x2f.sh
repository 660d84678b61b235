cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r4
python -m pytest tests/test_gpu_configs.py -x -q -m gpu -k "bf16x2" > gpurun_out/r4/gpu_tests_x2.log 2>&1; echo "pytest rc $?"; tail -3 gpurun_out/r4/gpu_tests_x2.log
for p in fp32 bf16x2 bf16; do python tools/stage_times.py 1 256 256 $p 1 50 2>&1 | grep -v amdgpu | head -1; done
python tools/stage_times.py 16 256 256 bf16x2 1 30 2>&1 | grep -v amdgpu | head -1
python tools/shape_sweep.py 30 2>&1 | tail -1
