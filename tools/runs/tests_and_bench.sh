set -o pipefail
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/full
O=gpurun_out/full
timeout -k 10 1100 python -m pytest tests -q -m gpu -x > $O/gpu_tests.log 2>&1 || { tail -40 $O/gpu_tests.log; exit 1; }
tail -3 $O/gpu_tests.log
timeout -k 10 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err || { tail -20 $O/bench_default.err; exit 1; }
python - <<'PY'
import json
d = json.loads(open("gpurun_out/full/bench_default.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms", d["ms_per_step"], "frac", d["roofline"]["frac"], "x2", d["fp32_contract_on_bf16_pipe"]["b4_1080p"]["value"],
      d["fp32_contract_on_bf16_pipe"]["config2_b16_256x256"]["value"], "rgb", d["rgb_6to3"]["value"], "video",
      d["video_sharded"]["interpolated_frames_per_s"], d["video_sharded"]["host_resident_frames_per_s"], d.get("video_sharded_efficiency"))
l = d["latency_256"]
print("latency_256", {k: l[k]["ms_per_forward"] for k in ("bf16", "bf16x2", "fp32")}, "b16 bf16", l["b16_256x256_bf16"]["value"],
      "fp32 cfg2", d["fp32"]["config2_b16_256x256"]["value"])
PY
