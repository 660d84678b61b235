cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r4
python bench.py > gpurun_out/r4/bench_final3.json 2> gpurun_out/r4/bench_final3.err; echo "bench rc $?"
python bench.py --no-cpu-baseline --video-frames 0 --steps 5 > gpurun_out/r4/bench_nocpu.json 2>/dev/null; echo "rc $?"
python - <<'PY'
import json
for f in ("gpurun_out/r4/bench_final3.json", "gpurun_out/r4/bench_nocpu.json"):
    d = json.loads([l for l in open(f) if l.startswith("{")][-1])
    x = d["fp32_contract_on_bf16_pipe"]
    print(d["value"], d["rgb_6to3"]["value"], d["rgb_6to3"]["parity_135x240"], x["config2_b16_256x256"]["value"], x["b4_1080p"]["value"], x.get("parity_270x480"))
PY
