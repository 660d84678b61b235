cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r4
for i in 1 2 3; do python bench.py --no-cpu-baseline --video-frames 0 --no-fp32 --no-power > gpurun_out/r4/bench_gap$i.json 2>/dev/null; done
python bench.py > gpurun_out/r4/bench_final4.json 2>/dev/null
python - <<'PY'
import json
for f in ("bench_gap1", "bench_gap2", "bench_gap3", "bench_final4"):
    d = json.loads([l for l in open(f"gpurun_out/r4/{f}.json") if l.startswith("{")][-1])
    print(f, d["value"], d["ms_per_step"], d["roofline"]["whole_forward"]["sum_stage_ms"], d["roofline"]["events_forwards"])
PY
