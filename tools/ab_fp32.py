"""A/B of two libraries on the fp32 legs (same box, separate processes, interleaved): python tools/ab_fp32.py libA.so libB.so [rounds]"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
libs = sys.argv[1:3]
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 2
code = r'''
import os, sys, json, torch
sys.path.insert(0, os.environ["FIUNET_ROOT"])
import bench
dev = torch.device("cuda:0")
m = bench.make_bench_model("fp32").to(dev).eval()
out = {}
for key, b, h, w, warm, steps in (("b4_1080p", 4, 1080, 1920, 1, 5), ("b1_1080p", 1, 1080, 1920, 2, 10), ("b16_256", 16, 256, 256, 10, 50), ("b2_720p", 2, 720, 1280, 2, 10)):
    g = torch.Generator(device=dev).manual_seed(1)
    f1 = torch.rand(b, 1, h, w, device=dev, generator=g) * 2 - 1; f2 = torch.rand(b, 1, h, w, device=dev, generator=g) * 2 - 1
    for _ in range(warm): m(f1, f2)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(steps): m(f1, f2)
    e1.record(); torch.cuda.synchronize()
    out[key] = round(b / (e0.elapsed_time(e1) / steps * 1e-3), 2)
print(json.dumps(out))
'''
for r in range(rounds):
    for lib in libs:
        env = dict(os.environ, FIUNET_ROOT=ROOT)
        if lib != "default": env["FIUNET_LIB"] = os.path.join(ROOT, lib)
        res = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
        line = [l for l in res.stdout.splitlines() if l.startswith("{")]
        print(f"round {r} {lib}: {line[-1] if line else 'FAILED ' + res.stderr[-300:]}", flush=True)
