cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r4
python -m pytest tests -q -m gpu -k "rgb or RGB or channel" > gpurun_out/r4/gpu_tests_rgb.log 2>&1; echo "pytest rc $?"; tail -5 gpurun_out/r4/gpu_tests_rgb.log
bash tools/runs/r4_rgb.sh
