set -o pipefail
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r5
timeout -k 10 1000 python -m pytest tests -x -q -m gpu -k "rgb or u8_read or RGB" > gpurun_out/r5/rgbstem2_tests.log 2>&1 || { tail -40 gpurun_out/r5/rgbstem2_tests.log; exit 1; }
tail -3 gpurun_out/r5/rgbstem2_tests.log
for v in ablibs/before_rgbstem.so default ablibs/before_rgbstem.so default; do
  if [ $v = default ]; then unset FIUNET_LIB; else export FIUNET_LIB=$v; fi
  echo "== $v"; timeout -k 10 300 python tools/stage_times.py 8 1080 1920 bf16 3 10 2>&1 | grep -E "frames/s|stem_rgb" || exit 1
done > gpurun_out/r5/rgb_stem_ab2.txt 2>&1
cat gpurun_out/r5/rgb_stem_ab2.txt
