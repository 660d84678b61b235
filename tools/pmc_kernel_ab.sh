#!/bin/bash
# A/B of SQ counters for ONE kernel between builds of libfiunet_hip.so (FIUNET_LIB), e.g. the fused-stem
# stage.  Run on the GPU box from the repo root:
#   bash tools/pmc_kernel_ab.sh ELi3ELi2E new=default r2=ab/lib_r2.so
# Prints, per build, the per-launch counters of the kernels whose mangled name contains the pattern.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
PAT=$1; shift
O=$R/gpurun_out/pmc_ab
rm -rf "$O"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --video-frames 0 --no-fp32 --no-power"
for spec in "$@"; do
  name=${spec%%=*}; lib=${spec#*=}
  if [ "$lib" != default ]; then export FIUNET_LIB=$R/$lib; else unset FIUNET_LIB; fi
  for grp in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
             "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAIT_INST_ANY"; do
    g=$(echo $grp | cut -c1-20 | tr ' ' '_')
    timeout -k 10 200 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$O/$name.$g" -- $B > "$O/$name.$g.log" 2>&1 || echo "pass $name $g failed"
  done
done
python3 - "$O" "$PAT" <<'PY'
import csv, glob, os, sys
from collections import defaultdict
o, pat = sys.argv[1], sys.argv[2]
for d in sorted(glob.glob(os.path.join(o, "*"))):
    if not os.path.isdir(d): continue
    acc, n = defaultdict(float), defaultdict(int)
    for f in glob.glob(os.path.join(d, "*", "*_counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            if pat in r["Kernel_Name"]:
                acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
    print(os.path.basename(d), " ".join(f"{k}={acc[k]/max(n[k],1):.4g}" for k in sorted(acc)))
PY
