set -o pipefail
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/stamps
O=gpurun_out/stamps
for lib in stamp stamp_prolog; do
  [ -f ablibs/lib_$lib.so ] || { echo "build ablibs/lib_$lib.so first (tools/runs/README.md)"; exit 1; }
  export FIUNET_LIB=ablibs/lib_$lib.so
  timeout -k 10 200 python tools/stamp_report.py 8 1080 1920 bf16 > $O/${lib}_b8_1080p_bf16.txt 2>&1 || { tail $O/${lib}_b8_1080p_bf16.txt; exit 1; }
  timeout -k 10 200 python tools/stamp_report.py 1 256 256 bf16 > $O/${lib}_b1_256_bf16.txt 2>&1 || { tail $O/${lib}_b1_256_bf16.txt; exit 1; }
done
export FIUNET_LIB=ablibs/lib_stamp.so
timeout -k 10 200 python tools/stamp_report.py 1 256 256 fp32 > $O/stamp_b1_256_fp32.txt 2>&1 || { tail $O/stamp_b1_256_fp32.txt; exit 1; }
cat $O/stamp_prolog_b8_1080p_bf16.txt
