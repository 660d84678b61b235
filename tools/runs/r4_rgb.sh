cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r4
python bench.py --no-cpu-baseline --video-frames 0 > gpurun_out/r4/bench_rgb.json 2> gpurun_out/r4/bench_rgb.err; echo "rc $?"
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r4/bench_rgb.json") if l.startswith("{")][-1])
print(d["value"], d.get("rgb_6to3"), d.get("power"))
PY
tail -3 gpurun_out/r4/bench_rgb.err
