"""Determinism stress of the K-cut path (round 6: the LAST workgroup of a tile to arrive adds the slices - who that is
changes from run to run, the result must not): small shapes whose deep levels are cut, every precision, 300 forwards
each against the first one, bit for bit; shapes alternate so that the arrival counters are reused by launches of
different geometry back to back.  python tools/stress_small.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dev = torch.device("cuda:0")
shapes = [(1, 256, 256), (2, 64, 96), (1, 17, 31), (3, 135, 240), (1, 720, 1280), (16, 64, 64)]
for prec in ("bf16", "fp32", "bf16x2"):
    m = bench.make_bench_model(prec).to(dev).eval()
    g = torch.Generator(device=dev).manual_seed(5)
    data = [(torch.rand(b, 1, h, w, device=dev, generator=g) * 2 - 1, torch.rand(b, 1, h, w, device=dev, generator=g) * 2 - 1)
            for b, h, w in shapes]
    ref = [m(a, c).clone() for a, c in data]
    bad = 0
    for it in range(300):
        for k, (a, c) in enumerate(data):
            if not torch.equal(m(a, c), ref[k]):
                bad += 1
    torch.cuda.synchronize()
    print(f"{prec}: {300 * len(shapes)} forwards over {shapes}: {bad} differ from their first run")
    assert bad == 0
print("ok")
