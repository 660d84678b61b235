set -o pipefail
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r5
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r5/rgbx2_tests.log 2>&1 || { tail -40 gpurun_out/r5/rgbx2_tests.log; exit 1; }
tail -3 gpurun_out/r5/rgbx2_tests.log
timeout -k 10 300 python tools/stage_times.py 4 1080 1920 bf16x2 3 5 > gpurun_out/r5/rgb_x2_stage_times.txt 2>&1 || { tail gpurun_out/r5/rgb_x2_stage_times.txt; exit 1; }
head -4 gpurun_out/r5/rgb_x2_stage_times.txt
