set -o pipefail
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r4
O=$PWD/gpurun_out/r4
python -m pytest tests -x -q -m gpu > $O/gpu_tests2.log 2>&1; echo "pytest rc $?" >> $O/gpu_tests2.log
P=$PWD/tools/probes/wave512_probe
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_w512 -- $P > $O/pmc_w512.log 2>&1
python3 - $O/pmc_w512 > $O/pmc_w512.txt <<'PY'
import csv, glob, os, sys
from collections import defaultdict
f = sorted(glob.glob(os.path.join(sys.argv[1], "*", "*_counter_collection.csv")), key=os.path.getmtime)[-1]
disp = defaultdict(dict)
for r in csv.DictReader(open(f)):
    d = disp[int(r["Dispatch_Id"])]
    d["t"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6
    d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
for i in sorted(disp)[-3:]:
    d = disp[i]
    print(i, {k: (round(v, 4) if k == "t" else f"{v:.4g}") for k, v in d.items()})
    if d.get("SQ_BUSY_CYCLES"):
        print("   mfma_busy/busy", round(d["SQ_VALU_MFMA_BUSY_CYCLES"] / d["SQ_BUSY_CYCLES"] , 4), " gui_active cycles", d.get("GRBM_GUI_ACTIVE"), " clock GHz", round(d.get("GRBM_GUI_ACTIVE", 0) / (d["t"] * 1e-3) / 1e9, 3))
PY
find $O/pmc_w512 -name "*.csv" -delete
tail -4 $O/gpu_tests2.log; cat $O/pmc_w512.txt
