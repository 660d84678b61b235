cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r4
python -m pytest tests -x -q -m gpu > gpurun_out/r4/gpu_tests5.log 2>&1; echo "pytest rc $?"; tail -3 gpurun_out/r4/gpu_tests5.log
python bench.py > gpurun_out/r4/bench_final.json 2> gpurun_out/r4/bench_final.err; echo "bench rc $?"
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r4/bench_final.json") if l.startswith("{")][-1])
print(d["value"], d["ms_per_step"], d["power"], d["rgb_6to3"]["value"], d["fp32"]["config2_b16_256x256"]["value"], d["video_sharded"]["interpolated_frames_per_s"], d["roofline"]["frac"], d["roofline"].get("rocprof_avg_launch_ms_committed_profile"), d["roofline"]["avg_launch_ms"])
PY
