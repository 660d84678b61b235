"""Where does ms_per_step - sum(stage ms) come from?  Times 20 back-to-back forwards (wall clock around a sync, as
bench.py does) with and without the per-stage event records, and the host time of the enqueue loop alone."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dev = torch.device("cuda:0")
m = bench.make_bench_model("bf16").to(dev).eval()
g = torch.Generator(device=dev).manual_seed(1)
f1 = torch.rand(8, 1, 1080, 1920, device=dev, generator=g) * 2 - 1
f2 = torch.rand(8, 1, 1080, 1920, device=dev, generator=g) * 2 - 1
for _ in range(3): m(f1, f2)
torch.cuda.synchronize()
for rep in range(4):
    for prof in (True, False):
        m._ctx.profile_enable(prof)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20): m(f1, f2)
        t_host = time.perf_counter() - t0
        torch.cuda.synchronize()
        t_all = time.perf_counter() - t0
        s = ""
        if prof:
            _, rows = m._ctx.profile_read()
            s = f" sum of stages {sum(r[1] for r in rows):.3f} ms"
        m._ctx.profile_enable(False)
        print(f"rep {rep} events={'on ' if prof else 'off'}: {t_all / 20 * 1e3:.3f} ms/step ({8 * 20 / t_all:.1f} frames/s), host enqueue {t_host / 20 * 1e3:.3f} ms/step{s}", flush=True)
