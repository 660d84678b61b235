cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r4
python -m pytest tests -x -q -m gpu > gpurun_out/r4/gpu_tests8.log 2>&1; echo "pytest rc $?"; tail -4 gpurun_out/r4/gpu_tests8.log
python bench.py --no-cpu-baseline --video-frames 0 > gpurun_out/r4/bench_x2.json 2> gpurun_out/r4/bench_x2.err; echo "bench rc $?"
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r4/bench_x2.json") if l.startswith("{")][-1])
print(d["value"], json.dumps(d.get("fp32_contract_on_bf16_pipe"), indent=1)[:1800])
print({k: v["value"] for k, v in d["fp32"].items()})
PY
