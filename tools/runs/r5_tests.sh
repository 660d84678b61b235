cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r5
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/r5/gpu_tests.log 2>&1
echo "rc $?" >> gpurun_out/r5/gpu_tests.log
tail -25 gpurun_out/r5/gpu_tests.log
