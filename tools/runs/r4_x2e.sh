cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r4
python -m pytest tests -x -q -m gpu > gpurun_out/r4/gpu_tests9.log 2>&1; echo "pytest rc $?"; tail -3 gpurun_out/r4/gpu_tests9.log
python bench.py > gpurun_out/r4/bench_final2.json 2> gpurun_out/r4/bench_final2.err; echo "bench rc $?"
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r4/bench_final2.json") if l.startswith("{")][-1])
x = d["fp32_contract_on_bf16_pipe"]
print(d["value"], d["rgb_6to3"]["value"], {k: v["value"] for k, v in d["fp32"].items()}, x["config2_b16_256x256"]["value"], x["b4_1080p"]["value"], x["parity_270x480"], d["power"]["socket_w_median"], d["power"]["sclk_mhz_median"])
PY
