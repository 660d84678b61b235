set -o pipefail
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r5
timeout -k 10 600 python -m pytest tests/test_gpu_dist.py -x -q -m gpu > gpurun_out/r5/dist_tests.log 2>&1 || { tail -30 gpurun_out/r5/dist_tests.log; exit 1; }
tail -3 gpurun_out/r5/dist_tests.log
FIUNET_BENCH_REHEARSE=1 timeout -k 10 600 python bench.py --gpus 2 --steps 5 --warmup 2 --video-frames 40 > gpurun_out/r5/rehearsal_2ranks_default.json 2> gpurun_out/r5/rehearsal_2ranks_default.err || { tail -30 gpurun_out/r5/rehearsal_2ranks_default.err; exit 1; }
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r5/rehearsal_2ranks_default.json").read().strip().splitlines()[-1])
print({k: d.get(k) for k in ("value", "n_gpus", "rccl_ranks_seen", "pairs_per_rank", "collective_backend")})
print("video", json.dumps(d.get("video_sharded"))[:600])
print("tile4k", json.dumps(d.get("tile4k"))[:600])
PY
