set -o pipefail
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/clock
O=gpurun_out/clock
[ -f ablibs/lib_clock.so ] || { echo "build ablibs/lib_clock.so first (tools/runs/README.md: EXTRA=-DFIUNET_CLOCK)"; exit 1; }
export FIUNET_LIB=ablibs/lib_clock.so
timeout -k 10 300 python tools/inkernel_clock.py 8 1080 1920 bf16 $O/inkernel_clock.json > $O/inkernel_clock_bf16.txt 2>&1 || { tail $O/inkernel_clock_bf16.txt; exit 1; }
tail -20 $O/inkernel_clock_bf16.txt
unset FIUNET_LIB
timeout -k 10 600 python bench.py --steps 400 --warmup 5 --no-cpu-baseline --video-frames 0 --no-fp32 --no-tile4k > $O/bench_sustained_400.json 2> $O/bench_sustained_400.err || { tail $O/bench_sustained_400.err; exit 1; }
python -c "
import json; d=json.loads(open('gpurun_out/clock/bench_sustained_400.json').read().strip().splitlines()[-1]); print('sustained 400 steps:', d['value'], d['ms_per_step'])"
