"""Per-dispatch durations and gaps of the LAST forward in a rocprofv3 --kernel-trace CSV.
usage: python tools/trace_forward.py <dir> <kernels per forward>"""
import csv, glob, os, sys
f = sorted(glob.glob(os.path.join(sys.argv[1], "**", "*_kernel_trace.csv"), recursive=True), key=os.path.getmtime)[-1]
per = int(sys.argv[2])
rows = [r for r in csv.DictReader(open(f)) if "fiunet" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
fw = rows[-per:]
t0 = int(fw[0]["Start_Timestamp"])
prev_end = None
for r in fw:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
    name = r["Kernel_Name"].replace("void fiunet::", "").split("(")[0]
    print(f"  +{(s - t0) / 1e3:8.1f} us  gap {gap:5.1f}  dur {(e - s) / 1e3:6.1f} us  grid {r.get('Grid_Size', '?'):>8}  {name[:90]}")
    prev_end = e
span = int(fw[-1]["End_Timestamp"]) - t0
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in fw)
print(f"span {span / 1e3:.1f} us, kernels {busy / 1e3:.1f} us, gaps {(span - busy) / 1e3:.1f} us, dispatches {per}")
