set -o pipefail
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r5
timeout -k 10 400 python tools/stress.py 40 bf16x2 > gpurun_out/r5/stress_x2.txt 2>&1 || { tail gpurun_out/r5/stress_x2.txt; exit 1; }
tail -2 gpurun_out/r5/stress_x2.txt
timeout -k 10 400 python tools/stress.py 40 bf16 > gpurun_out/r5/stress_bf16.txt 2>&1 || { tail gpurun_out/r5/stress_bf16.txt; exit 1; }
tail -2 gpurun_out/r5/stress_bf16.txt
