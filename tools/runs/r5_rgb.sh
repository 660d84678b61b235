set -o pipefail
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r5
timeout -k 10 300 python tools/stage_times.py 8 1080 1920 bf16 3 10 > gpurun_out/r5/rgb_stage_times.txt 2>&1 || exit 1
cat gpurun_out/r5/rgb_stage_times.txt
timeout -k 10 600 python -m pytest tests -x -q -m gpu -k "rgb or u8_read" > gpurun_out/r5/rgb_tests.log 2>&1 || exit 1; tail -3 gpurun_out/r5/rgb_tests.log
