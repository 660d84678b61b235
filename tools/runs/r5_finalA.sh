set -o pipefail
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r5f
[ -f ablibs/lib_clock.so ] || { mkdir -p ablibs && make -C ai_based_frame_interpolation_amd/csrc OUT=../../ablibs/lib_clock.so EXTRA=-DFIUNET_CLOCK > /dev/null; }   # diagnostic build (git-ignored): built on the box when absent
O=gpurun_out/r5f
timeout -k 10 1000 python -m pytest tests -q -m gpu > $O/gpu_tests.log 2>&1 || { tail -30 $O/gpu_tests.log; exit 1; }
tail -3 $O/gpu_tests.log
timeout -k 10 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err || { tail -20 $O/bench_default.err; exit 1; }
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r5f/bench_default.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms", d["ms_per_step"], "frac", d["roofline"]["frac"], "x2", d["fp32_contract_on_bf16_pipe"]["b4_1080p"]["value"], d["fp32_contract_on_bf16_pipe"]["config2_b16_256x256"]["value"], "rgb", d["rgb_6to3"]["value"], "video", d["video_sharded"]["interpolated_frames_per_s"], d["video_sharded"]["host_resident_frames_per_s"])
PY
export FIUNET_LIB=ablibs/lib_clock.so
timeout -k 10 300 python tools/inkernel_clock.py 8 1080 1920 bf16 $O/inkernel_clock.json > $O/inkernel_clock_bf16.txt 2>&1 || { tail $O/inkernel_clock_bf16.txt; exit 1; }
timeout -k 10 300 python tools/inkernel_clock.py 4 1080 1920 bf16x2 > $O/inkernel_clock_bf16x2.txt 2>&1 || { tail $O/inkernel_clock_bf16x2.txt; exit 1; }
tail -18 $O/inkernel_clock_bf16.txt
