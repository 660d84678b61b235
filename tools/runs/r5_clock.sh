set -o pipefail
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r5
export FIUNET_LIB=ablibs/lib_clock.so
timeout -k 10 300 python tools/inkernel_clock.py 8 1080 1920 bf16 > gpurun_out/r5/inkernel_clock_bf16.txt 2>&1 || exit 1
echo "rc $?" >> gpurun_out/r5/inkernel_clock_bf16.txt
timeout -k 10 300 python tools/inkernel_clock.py 4 1080 1920 bf16x2 > gpurun_out/r5/inkernel_clock_bf16x2.txt 2>&1 || exit 1
echo "rc $?" >> gpurun_out/r5/inkernel_clock_bf16x2.txt
cat gpurun_out/r5/inkernel_clock_bf16.txt gpurun_out/r5/inkernel_clock_bf16x2.txt
