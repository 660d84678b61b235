set -o pipefail
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r4
O=$PWD/gpurun_out/r4
P=$PWD/tools/probes/wave512_probe
timeout -k 10 120 $P > $O/wave512_probe_v2.txt 2>&1; echo "probe rc $?" >> $O/wave512_probe_v2.txt
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_w512 -- $P > $O/pmc_w512.log 2>&1
python3 - $O/pmc_w512 > $O/pmc_w512_v2.txt <<'PY'
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
    act = d["GRBM_GUI_ACTIVE"] / 8
    print(f"dispatch {i}: {d['t']:.4f} ms  clock {act / (d['t'] * 1e-3) / 1e9:.3f} GHz  mfma-busy {d['SQ_VALU_MFMA_BUSY_CYCLES'] / (act * 1024):.4f}  "
          f"lds conflicts/active {d['SQ_LDS_BANK_CONFLICT'] / max(d['SQ_LDS_IDX_ACTIVE'], 1):.4f}  wait_inst/wave {d['SQ_WAIT_INST_ANY'] / d['SQ_WAVE_CYCLES']:.3f}")
PY
find $O/pmc_w512 -name "*.csv" -delete
cat $O/wave512_probe_v2.txt $O/pmc_w512_v2.txt
