#!/bin/bash
# The rocprofv3 passes behind profiles/: kernel trace + stats, then one PMC pass per counter group
# (separate passes, as /opt/skills/guides/MI355X_MICROARCH.md prescribes), all on the bench's default
# workload (or BENCH_ARGS="--batch 16 --height 256 --width 256 --precision fp32" for config 2).
# Run on the GPU box from the repo root:  bash tools/profile_all.sh [tag]
# Afterwards (anywhere): python tools/pmc_summary.py gpurun_out/prof_<tag> profiles/r03_pmc 4 <commit>
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
TAG=${1:-default}
O=$R/gpurun_out/prof_$TAG
rm -rf "$O"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --video-frames 0 --no-fp32 --no-power --no-tile4k ${BENCH_ARGS:-}"
run() { name=$1; shift; timeout -k 10 300 rocprofv3 "$@" --output-format csv -d "$O/$name" -- $B > "$O/$name.log" 2>&1 && echo "pass $name ok"; }
run trace --kernel-trace --stats &&
run fetch --pmc FETCH_SIZE --kernel-trace &&
run write --pmc WRITE_SIZE --kernel-trace &&
run sq --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace
rc=$?
find "$O" -name "*kernel_stats.csv" -exec cp {} "$O/kernel_stats.csv" \;
# keep only what the summary needs (the raw traces are large)
find "$O" -name "*.csv" ! -name "*counter_collection.csv" ! -name "kernel_stats.csv" -delete
du -sh "$O"
exit $rc
