cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r4
timeout -k 10 600 python -m pytest tests/test_gpu_configs.py -q -m gpu -k "convtranspose" > gpurun_out/r4/gpu_tests_ct.log 2>&1; echo "pytest rc $?"; tail -30 gpurun_out/r4/gpu_tests_ct.log
