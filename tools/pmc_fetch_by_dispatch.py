"""Per-dispatch HBM read bytes of the conv kernels from ONE rocprofv3 `--pmc FETCH_SIZE --kernel-trace` pass of
bench.py: python tools/pmc_fetch_by_dispatch.py <pass dir> [forwards]   (last forward only; FETCH_SIZE is in KiB
and counts 128-B requests as 64 B on gfx950 for wide streams: read bytes = 2 * FETCH_SIZE * 1024, the guide's rule)"""
import csv, glob, os, sys
from collections import defaultdict
src = sys.argv[1]
nfw = int(sys.argv[2]) if len(sys.argv) > 2 else 3
files = sorted(glob.glob(os.path.join(src, "*", "*_counter_collection.csv")), key=os.path.getmtime)[-1:]
disp = defaultdict(dict)
for r in csv.DictReader(open(files[0])):
    d = disp[int(r["Dispatch_Id"])]
    d["name"], d["grid"] = r["Kernel_Name"], int(r["Grid_Size"])
    d["t"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6
    d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
ids = sorted(k for k, v in disp.items() if "conv3x3" in v["name"] or "upsample_kernel" in v["name"])
per = len(ids) // nfw
for n, i in enumerate(ids[-per:]):
    d = disp[i]
    nm = d["name"].split("(")[0][-60:]
    print(f"{n:2d} grid {d['grid'] // 256:6d} wg  {d['t']:7.3f} ms  read {2 * d.get('FETCH_SIZE', 0) * 1024 / 1e9:6.3f} GB  {nm}")
