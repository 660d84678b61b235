cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r4
python -m pytest tests -x -q -m gpu > gpurun_out/r4/gpu_tests10.log 2>&1; echo "pytest rc $?"; tail -3 gpurun_out/r4/gpu_tests10.log
python bench.py > gpurun_out/r4/bench_final5.json 2>/dev/null; echo "bench rc $?"
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r4/bench_final5.json") if l.startswith("{")][-1])
x = d["fp32_contract_on_bf16_pipe"]
print(d["value"], d["ms_per_step"], d["roofline"]["whole_forward"]["sum_stage_ms"], d["rgb_6to3"]["value"], {k: v["value"] for k, v in d["fp32"].items()}, x["config2_b16_256x256"]["value"], x["b4_1080p"]["value"], x["parity_270x480"]["max_abs_vs_cpu_ref"])
PY
