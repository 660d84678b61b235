"""A/B several builds of libfiunet_hip.so on ONE box: interleaved rounds, separate processes.
usage: python tools/ab_bench.py name=path.so[,ENV=VALUE...] ... [--rounds 3] [--steps 10] [-- extra bench args]
(`default` as the path = the in-tree library; ENV=VALUE pairs are set for that variant only)"""
import json, os, subprocess, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = [a for a in sys.argv[1:] if "=" in a and not a.startswith("--")]
rounds = int(sys.argv[sys.argv.index("--rounds") + 1]) if "--rounds" in sys.argv else 3
steps = sys.argv[sys.argv.index("--steps") + 1] if "--steps" in sys.argv else "10"
extra = sys.argv[sys.argv.index("--") + 1:] if "--" in sys.argv else []
libs = [a.split("=", 1) for a in args]
res = {n: [] for n, _ in libs}
stages = {}
for r in range(rounds):
    for n, p in libs:
        env = dict(os.environ)
        p, *kv = p.split(",")
        for item in kv:
            k, v = item.split("=", 1)
            env[k] = v
        if p != "default":
            env["FIUNET_LIB"] = os.path.join(ROOT, p)
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", steps,
                              "--no-cpu-baseline", "--no-power"] + extra, env=env, capture_output=True, text=True)
        line = [l for l in out.stdout.splitlines() if l.startswith("{")]
        if not line:
            print(n, "FAILED", out.stderr[-400:]); continue
        d = json.loads(line[-1])
        res[n].append(d["value"]); stages[n] = d["roofline"]["stages"]
        print(f"round {r} {n}: {d['value']:.1f} fps", flush=True)
for n in res:
    if res[n]:
        print(f"{n:12s} median {statistics.median(res[n]):.1f} min {min(res[n]):.1f} max {max(res[n]):.1f}")
names = [n for n in res if n in stages]
if len(names) >= 2:
    for i in range(18):
        print("  " + " | ".join(f"{stages[n][i]['ms']:.3f} {stages[n][i]['tflops']:7.1f}" for n in names) + "  " + stages[names[0]][i]["kernel"])
