set -o pipefail
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/ab
for l in base free_prologue no_wstream; do [ -f ablibs/lib_$l.so ] || { echo "build ablibs/lib_$l.so first (tools/runs/README.md: EXTRA=-DFIUNET_DIAG_FREE_PROLOGUE / -DFIUNET_DIAG_NO_WSTREAM)"; exit 1; }; done
timeout -k 10 900 python tools/ab_bench.py base=ablibs/lib_base.so free_prologue=ablibs/lib_free_prologue.so no_wstream=ablibs/lib_no_wstream.so \
    --rounds 3 --steps 10 -- --video-frames 0 --no-fp32 --no-tile4k > gpurun_out/ab/ab_levers.txt 2>&1 || { tail gpurun_out/ab/ab_levers.txt; exit 1; }
cat gpurun_out/ab/ab_levers.txt
