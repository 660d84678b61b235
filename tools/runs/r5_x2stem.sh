set -o pipefail
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r5
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r5/x2stem_tests.log 2>&1 || { tail -40 gpurun_out/r5/x2stem_tests.log; exit 1; }
tail -3 gpurun_out/r5/x2stem_tests.log
timeout -k 10 600 python tools/ab_x2.py unfused_stem=ablibs/pitch40.so fused_stem=default --rounds 3 > gpurun_out/r5/ab_x2_fused_stem.txt 2>&1 || { cat gpurun_out/r5/ab_x2_fused_stem.txt; exit 1; }
cat gpurun_out/r5/ab_x2_fused_stem.txt
timeout -k 10 600 python tools/ab_x2.py old=ablibs/pitch40.so new=default --rounds 2 --shape 8 1080 1920 --precision bf16 > gpurun_out/r5/ab_bf16_after_x2stem.txt 2>&1 || { cat gpurun_out/r5/ab_bf16_after_x2stem.txt; exit 1; }
head -8 gpurun_out/r5/ab_bf16_after_x2stem.txt
