cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r4
python -m pytest tests -x -q -m gpu > gpurun_out/r4/gpu_tests7.log 2>&1; echo "pytest rc $?"; tail -3 gpurun_out/r4/gpu_tests7.log
