cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r5
FIUNET_LIB=ablibs/lib_stamp.so timeout -k 10 300 python tools/stamp_report.py > gpurun_out/r5/stamp_phases.txt 2>&1 || { tail gpurun_out/r5/stamp_phases.txt; exit 1; }
cat gpurun_out/r5/stamp_phases.txt
