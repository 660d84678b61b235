"""Summarise `make -C ai_based_frame_interpolation_amd/csrc asm` output (build/asm/resource_usage.txt):
one line per kernel with registers, spills, scratch, occupancy and LDS.
    python tools/resource_report.py [resource_usage.txt]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "build", "asm", "resource_usage.txt")
txt = open(path).read()
blocks = re.split(r"remark: [^\n]*Function Name: ", txt)[1:]
names = [b.split("\n")[0].split()[0] for b in blocks]
dem = subprocess.run(["c++filt"] + names, capture_output=True,
                     text=True).stdout.strip().split("\n")
KEYS = [("VGPR", r"VGPRs"), ("AGPR", r"AGPRs"), ("SGPR", r"SGPRs"), ("spillV", r"VGPRs? Spill"),
        ("spillS", r"SGPRs? Spill"), ("scratch", r"ScratchSize \[bytes/lane\]"),
        ("occ", r"Occupancy \[waves/SIMD\]"), ("LDS", r"LDS Size \[bytes/block\]")]
bad = 0
for b, d in zip(blocks, dem):
    vals = []
    for label, pat in KEYS:
        m = re.search(pat + r": (\d+)", b)
        vals.append((label, m.group(1) if m else "?"))
    d = re.sub(r"fiunet::|\(fiunet::ConvArgs\)|void ", "", d)
    d = d.replace("__hip_bfloat16", "bf16").replace("__bf16", "bf16")
    spill = dict(vals)
    flag = " <-- SPILL" if spill["spillV"] not in ("0", "?") or spill["scratch"] not in ("0", "?") else ""
    bad += bool(flag)
    print(f"{d[:64]:64s} " + " ".join(f"{k} {v:>4s}" for k, v in vals) + flag)
print(f"{len(blocks)} kernels, {bad} with VGPR spills / scratch")
