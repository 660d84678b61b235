set -o pipefail
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r5
[ -f ablibs/lib_clock.so ] || { mkdir -p ablibs && make -C ai_based_frame_interpolation_amd/csrc OUT=../../ablibs/lib_clock.so EXTRA=-DFIUNET_CLOCK > /dev/null; }   # diagnostic build (git-ignored): built on the box when absent
export FIUNET_LIB=ablibs/lib_clock.so
timeout -k 10 300 python tools/inkernel_clock.py 8 1080 1920 bf16 > gpurun_out/r5/inkernel_clock_bf16.txt 2>&1 || exit 1
echo "rc $?" >> gpurun_out/r5/inkernel_clock_bf16.txt
timeout -k 10 300 python tools/inkernel_clock.py 4 1080 1920 bf16x2 > gpurun_out/r5/inkernel_clock_bf16x2.txt 2>&1 || exit 1
echo "rc $?" >> gpurun_out/r5/inkernel_clock_bf16x2.txt
cat gpurun_out/r5/inkernel_clock_bf16.txt gpurun_out/r5/inkernel_clock_bf16x2.txt
