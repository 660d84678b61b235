set -o pipefail
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r5
R=$PWD; O=$R/gpurun_out/r5/prof_rgb; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/sq -- python3 $R/tools/stage_times.py 8 1080 1920 bf16 3 3 > $O/sq.log 2>&1 || { tail $O/sq.log; exit 1; }
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --kernel-trace --output-format csv -d $O/inst -- python3 $R/tools/stage_times.py 8 1080 1920 bf16 3 3 > $O/inst.log 2>&1 || { tail $O/inst.log; exit 1; }
python3 - $O > $R/gpurun_out/r5/rgb_stem_pmc.txt <<'PY'
import csv, glob, os, sys
from collections import defaultdict
for pas in ("sq", "inst"):
    f = sorted(glob.glob(os.path.join(sys.argv[1], pas, "*", "*_counter_collection.csv")), key=os.path.getmtime)[-1]
    disp = defaultdict(dict)
    for r in csv.DictReader(open(f)):
        d = disp[int(r["Dispatch_Id"])]
        d["name"] = r["Kernel_Name"]; d["t"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6
        d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    last = [d for i, d in sorted(disp.items()) if "stem_rgb" in d["name"]][-1]
    print(pas, {k: (round(v, 4) if isinstance(v, float) else v) for k, v in last.items() if k != "name"})
PY
find $O -name "*.csv" -delete
cat $R/gpurun_out/r5/rgb_stem_pmc.txt
