"""A/B of option flags in ONE process on ONE box (interleaved rounds; cdna guide rule 24):
legacy two-workgroup tiles vs the tile-pair kernel, per stage.
usage: python tools/ab_flags.py [--rounds 4] [--steps 8] [--batch 8] [--height 1080] [--width 1920] [--precision bf16]"""
import argparse, os, statistics, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import bench
ap = argparse.ArgumentParser()
ap.add_argument("--rounds", type=int, default=4); ap.add_argument("--steps", type=int, default=8)
ap.add_argument("--batch", type=int, default=8); ap.add_argument("--height", type=int, default=1080)
ap.add_argument("--width", type=int, default=1920); ap.add_argument("--precision", default="bf16")
a = ap.parse_args()
dev = torch.device("cuda:0")
m = bench.make_bench_model(a.precision).to(dev).eval()
gen = torch.Generator(device=dev).manual_seed(1)
f1 = torch.rand(a.batch, 1, a.height, a.width, device=dev, generator=gen) * 2 - 1
f2 = torch.rand(a.batch, 1, a.height, a.width, device=dev, generator=gen) * 2 - 1
arms = {"legacy": dict(), "new": dict(pair_tiles=True)}
fps = {k: [] for k in arms}; stages = {k: None for k in arms}
for k, opt in arms.items():
    m.set_options(**opt)
    for _ in range(3): m(f1, f2)
torch.cuda.synchronize()
for r in range(a.rounds):
    for k, opt in arms.items():
        m.set_options(**opt)
        m(f1, f2)
        m._ctx.profile_enable(True)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(a.steps): m(f1, f2)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        n, rows = m._ctx.profile_read(); m._ctx.profile_enable(False)
        fps[k].append(a.batch * a.steps / dt)
        if stages[k] is None: stages[k] = [[] for _ in rows]
        for i, (nm, ms, fl) in enumerate(rows): stages[k][i].append((nm, ms, fl))
        print(f"round {r} {k}: {fps[k][-1]:.1f} fps", flush=True)
for k in arms:
    print(f"{k:8s} median {statistics.median(fps[k]):.1f}  min {min(fps[k]):.1f}  max {max(fps[k]):.1f} frames/s")
print(f"{'stage':>5} {'legacy ms':>10} {'TF/s':>7} | {'new ms':>10} {'TF/s':>7} | ratio  kernel (new)")
for i in range(len(stages["legacy"])):
    l = stages["legacy"][i]; n = stages["new"][i]
    lm = statistics.median(x[1] for x in l); nm = statistics.median(x[1] for x in n)
    fl = l[0][2]
    tf = lambda ms: fl / (ms * 1e-3) / 1e12 if ms > 0 else 0
    print(f"{i:5d} {lm:10.4f} {tf(lm):7.1f} | {nm:10.4f} {tf(nm):7.1f} | {lm / nm if nm > 0 else 0:5.3f}  {n[0][0]}")
