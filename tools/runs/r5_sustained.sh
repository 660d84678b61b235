set -o pipefail
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r5
timeout -k 10 300 python bench.py --steps 400 --warmup 5 --no-cpu-baseline --no-fp32 --video-frames 0 --no-tile4k --no-power > gpurun_out/r5/bench_sustained_400.json 2> gpurun_out/r5/bench_sustained_400.err || { tail gpurun_out/r5/bench_sustained_400.err; exit 1; }
python -c "
import json; d=json.loads(open('gpurun_out/r5/bench_sustained_400.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('frac_of_peak_at_sustained_clock'), d['roofline'].get('rocprof_avg_launch_ms_committed_profile'))"
