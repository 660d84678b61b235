cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r4
python bench.py > gpurun_out/r4/bench_default.json 2> gpurun_out/r4/bench_default.err; echo "rc $?"
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r4/bench_default.json") if l.startswith("{")][-1])
print(d["value"], d["ms_per_step"], d.get("power"))
r = d["roofline"]; print({k: r[k] for k in r if k not in ("stages", "traffic_source")})
print(d.get("fp32", {}).get("config2_b16_256x256", {}).get("value"), d.get("video_sharded"))
print(d.get("parity", {}).get("psnr_delta_sweep"))
PY
