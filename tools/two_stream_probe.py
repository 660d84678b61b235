"""Experiment: does running two half-batches on two HIP streams (kernels of different layers
overlapping, filling each other's last partial round of workgroups) beat one full batch?
usage: python tools/two_stream_probe.py [--rounds 3] [--steps 10] [--batch 8]"""
import argparse, os, statistics, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import bench
ap = argparse.ArgumentParser()
ap.add_argument("--rounds", type=int, default=3); ap.add_argument("--steps", type=int, default=10)
ap.add_argument("--batch", type=int, default=8); ap.add_argument("--height", type=int, default=1080)
ap.add_argument("--width", type=int, default=1920); ap.add_argument("--parts", type=int, default=2); ap.add_argument("--sizes", default="")
a = ap.parse_args()
dev = torch.device("cuda:0")
models = [bench.make_bench_model("bf16").to(dev).eval() for _ in range(5)]
gen = torch.Generator(device=dev).manual_seed(1)
f1 = torch.rand(a.batch, 1, a.height, a.width, device=dev, generator=gen) * 2 - 1
f2 = torch.rand(a.batch, 1, a.height, a.width, device=dev, generator=gen) * 2 - 1
streams = [torch.cuda.Stream(device=dev) for _ in range(4)]
sizes = [int(x) for x in a.sizes.split(",")] if a.sizes else [a.batch // a.parts] * a.parts
a.parts = len(sizes); assert sum(sizes) == a.batch
offs = [sum(sizes[:i]) for i in range(a.parts)]
parts = [(f1[o:o + n].contiguous(), f2[o:o + n].contiguous()) for o, n in zip(offs, sizes)]

def one():
    return models[-1](f1, f2)

def split():
    outs = []
    cur = torch.cuda.current_stream()
    for i, s in enumerate(streams[:a.parts]):
        s.wait_stream(cur)
        with torch.cuda.stream(s):
            outs.append(models[i](*parts[i]))
    for s in streams[:a.parts]:
        cur.wait_stream(s)
    return outs

ref = one(); got = torch.cat(split(), 0); torch.cuda.synchronize()
print("max |one - split| =", (ref - got).abs().max().item())
arms = {"one": one, "split": split}
fps = {k: [] for k in arms}
for fn in arms.values():
    for _ in range(3): fn()
torch.cuda.synchronize()
for r in range(a.rounds):
    for k, fn in arms.items():
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(a.steps): fn()
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        fps[k].append(a.batch * a.steps / dt)
        print(f"round {r} {k}: {fps[k][-1]:.1f} fps", flush=True)
for k in arms:
    print(f"{k:6s} median {statistics.median(fps[k]):.1f}  min {min(fps[k]):.1f}  max {max(fps[k]):.1f} frames/s")
