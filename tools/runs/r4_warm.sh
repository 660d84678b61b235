cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r4
for rep in 1 2; do for w in 3 20 60; do
  python bench.py --warmup $w --steps 20 --no-cpu-baseline --video-frames 0 --no-fp32 --no-power 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('warmup', d['warmup'], 'steps', d['steps'], d['value'], d['ms_per_step'])"
  sleep 3
done; done | tee gpurun_out/r4/warmup_sweep.txt
