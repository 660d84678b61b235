cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r4
python -m pytest tests -x -q -m gpu > gpurun_out/r4/gpu_tests4.log 2>&1; echo "pytest rc $?"; tail -3 gpurun_out/r4/gpu_tests4.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
