set -o pipefail
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r6c
timeout -k 10 800 python tools/ab_bench.py base=ablibs/lib_base.so free_prologue=ablibs/lib_free_prologue.so no_wstream=ablibs/lib_no_wstream.so --rounds 3 --steps 10 -- --video-frames 0 --no-fp32 --no-tile4k > gpurun_out/r6c/ab_levers.txt 2>&1 || { tail gpurun_out/r6c/ab_levers.txt; exit 1; }
cat gpurun_out/r6c/ab_levers.txt
