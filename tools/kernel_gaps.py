"""Idle time between the kernels of one forward, from a rocprofv3 --kernel-trace CSV.
usage: python tools/kernel_gaps.py <dir with *_kernel_trace.csv>"""
import csv, glob, os, sys
f = sorted(glob.glob(os.path.join(sys.argv[1], "**", "*_kernel_trace.csv"), recursive=True), key=os.path.getmtime)[-1]
rows = [r for r in csv.DictReader(open(f)) if "fiunet" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = len(rows)
per = 19  # kernels per bf16 gray forward: 17 convs + 2 upsample kernels
fw = rows[-per:]
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in fw)
span = int(fw[-1]["End_Timestamp"]) - int(fw[0]["Start_Timestamp"])
gaps = [int(fw[i + 1]["Start_Timestamp"]) - int(fw[i]["End_Timestamp"]) for i in range(per - 1)]
print(f"{f}: {n} fiunet dispatches; last forward: span {span/1e6:.3f} ms, kernels {busy/1e6:.3f} ms, "
      f"gaps {sum(gaps)/1e3:.1f} us total ({100.0*sum(gaps)/span:.2f} %), max {max(gaps)/1e3:.1f} us, "
      f"median {sorted(gaps)[len(gaps)//2]/1e3:.1f} us")
