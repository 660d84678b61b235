cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r4
bash tools/profile_all.sh r4 > gpurun_out/r4/profile_all.log 2>&1; echo "profile rc $?"
tail -8 gpurun_out/r4/profile_all.log
python bench.py > gpurun_out/r4/bench_default2.json 2> gpurun_out/r4/bench_default2.err; echo "bench rc $?"
python bench.py --steps 400 --no-cpu-baseline --video-frames 0 --no-fp32 > gpurun_out/r4/bench_sustained400.json 2>/dev/null; echo "sustained rc $?"
