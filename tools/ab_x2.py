"""A/B builds of libfiunet_hip.so in precision bf16x2 on ONE box: interleaved rounds, separate processes.
usage: python tools/ab_x2.py name=path.so ... [--rounds 3] [--shape B H W] (path `default` = the in-tree library)"""
import os, re, statistics, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
libs = [a.split("=", 1) for a in sys.argv[1:] if "=" in a]
rounds = int(sys.argv[sys.argv.index("--rounds") + 1]) if "--rounds" in sys.argv else 3
shape = sys.argv[sys.argv.index("--shape") + 1:sys.argv.index("--shape") + 4] if "--shape" in sys.argv else ["4", "1080", "1920"]
prec = sys.argv[sys.argv.index("--precision") + 1] if "--precision" in sys.argv else "bf16x2"
fps = {n: [] for n, _ in libs}
last = {}
for r in range(rounds):
    for n, p in libs:
        env = dict(os.environ)
        if p != "default":
            env["FIUNET_LIB"] = os.path.join(ROOT, p)
        run = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "stage_times.py"), *shape, prec, "1", "8"],
                             env=env, capture_output=True, text=True)
        out = run.stdout
        if run.returncode != 0 or "core dump" in (run.stdout + run.stderr).lower():
            print(n, "FAILED rc", run.returncode, (run.stdout + run.stderr)[-400:]); sys.exit(1)
        m = re.search(r"([\d.]+) frames/s", out)
        if not m:   # a failed arm (possibly a GPU fault): stop at once, never launch another GPU process after it
            print(n, "FAILED", out[-300:]); sys.exit(1)
        fps[n].append(float(m.group(1))); last[n] = out
        print(f"round {r} {n}: {m.group(1)} frames/s", flush=True)
for n in fps:
    if fps[n]:
        print(f"{n:16s} median {statistics.median(fps[n]):.1f} min {min(fps[n]):.1f} max {max(fps[n]):.1f}")
rows = {n: [l for l in last[n].splitlines() if " ms " in l and "TFLOP/s" in l] for n in last}
names = list(rows)
for i in range(min(len(v) for v in rows.values()) if rows else 0):
    print("  " + " | ".join(rows[n][i].split("TFLOP/s")[0].strip() for n in names) + "  " + rows[names[-1]][i].split("TFLOP/s")[1].strip())
