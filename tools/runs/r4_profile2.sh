cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r4
bash tools/profile_all.sh r4final > gpurun_out/r4/profile_all2.log 2>&1; echo "profile rc $?"
tail -6 gpurun_out/r4/profile_all2.log
python bench.py > gpurun_out/r4/bench_final6.json 2>/dev/null; echo "bench rc $?"
