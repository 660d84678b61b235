set -o pipefail
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r5
for b in 1 2 4 8 12 16; do
  timeout -k 10 200 python tools/stage_times.py $b 1080 1920 bf16 1 10 2>&1 | grep "frames/s" || exit 1
done > gpurun_out/r5/batch_sweep_1080p_bf16.txt
cat gpurun_out/r5/batch_sweep_1080p_bf16.txt
