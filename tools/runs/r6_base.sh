set -o pipefail
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r6a
O=$GRAFT_REPO_ROOT/gpurun_out/r6a
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for p in bf16 fp32; do
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $O/tr_$p -- python3 $R/tools/latency_trace.py $p > $O/lat_$p.log 2>&1 || { tail $O/lat_$p.log; exit 1; }
  tail -1 $O/lat_$p.log
done
cd $R
python tools/trace_forward.py $O/tr_bf16 40 > $O/trace_bf16.txt; python tools/trace_forward.py $O/tr_fp32 40 > $O/trace_fp32.txt
find $O -name "*.csv" -delete; find $O -name "*.db" -delete
python tools/latency.py > $O/latency.txt 2>&1 || { tail $O/latency.txt; exit 1; }
cat $O/latency.txt
timeout -k 10 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err || { tail -20 $O/bench_default.err; exit 1; }
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r6a/bench_default.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms", d["ms_per_step"], "frac", d["roofline"]["frac"], "x2", d["fp32_contract_on_bf16_pipe"]["b4_1080p"]["value"], d["fp32_contract_on_bf16_pipe"]["config2_b16_256x256"]["value"], "rgb", d["rgb_6to3"]["value"], "video", d["video_sharded"]["interpolated_frames_per_s"], d["video_sharded"]["host_resident_frames_per_s"])
PY
