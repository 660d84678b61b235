cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r5
[ -f ablibs/lib_stamp.so ] || { mkdir -p ablibs && make -C ai_based_frame_interpolation_amd/csrc OUT=../../ablibs/lib_stamp.so EXTRA=-DFIUNET_STAMP > /dev/null; }   # diagnostic build (git-ignored): built on the box when absent
FIUNET_LIB=ablibs/lib_stamp.so timeout -k 10 300 python tools/stamp_report.py > gpurun_out/r5/stamp_phases.txt 2>&1 || { tail gpurun_out/r5/stamp_phases.txt; exit 1; }
cat gpurun_out/r5/stamp_phases.txt
