set -o pipefail
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r5
R=$PWD; O=$R/gpurun_out/r5/prof_x2; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/tools/stage_times.py 4 1080 1920 bf16x2 1 5 > $O/trace.log 2>&1 || exit 1; echo "trace rc $?"
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/sq -- python3 $R/tools/stage_times.py 4 1080 1920 bf16x2 1 3 > $O/sq.log 2>&1 || exit 1; echo "pmc rc $?"
find $O -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/r5/x2_kernel_stats.csv \;
python3 - $O/sq > $R/gpurun_out/r5/x2_pmc.txt <<'PY'
import csv, glob, os, sys
from collections import defaultdict
f = sorted(glob.glob(os.path.join(sys.argv[1], "*", "*_counter_collection.csv")), key=os.path.getmtime)[-1]
disp = defaultdict(dict)
for r in csv.DictReader(open(f)):
    d = disp[int(r["Dispatch_Id"])]
    d["name"] = r["Kernel_Name"]
    d["t"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6
    d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
ids = sorted(disp)
last = ids[-21:]   # one forward: 17 convs (the stem is fused into the first) + 4 x2_upsample launches
for i in last:
    d = disp[i]
    if "GRBM_GUI_ACTIVE" not in d: continue
    act = d["GRBM_GUI_ACTIVE"] / 8
    print(f"{d['t']:7.3f} ms  clock {act / (d['t'] * 1e-3) / 1e9:5.2f} GHz  mfma-busy {d.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / (act * 1024):5.3f}  {d['name'][:90]}")
PY
find $O -name "*.csv" -delete; cat $O/trace.log
head -12 $R/gpurun_out/r5/x2_kernel_stats.csv | cut -c1-160; cat $R/gpurun_out/r5/x2_pmc.txt
