cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r5
timeout -k 10 900 python bench.py > gpurun_out/r5/bench_default.json 2> gpurun_out/r5/bench_default.err
echo "rc $?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r5/bench_default.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms", d["ms_per_step"], "frac", d["roofline"]["frac"])
for k in ("video_sharded", "tile4k", "rgb_6to3", "fp32_contract_on_bf16_pipe", "power"):
    v = d.get(k)
    if isinstance(v, dict):
        v = {a: b for a, b in v.items() if a not in ("stages", "rows")}
    print(k, json.dumps(v)[:1500])
print("fp32", json.dumps(d.get("fp32"))[:600])
print("cpu", json.dumps(d.get("cpu_baseline"))[:400])
PY
tail -5 gpurun_out/r5/bench_default.err
