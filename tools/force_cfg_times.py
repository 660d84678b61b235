"""Stage times of one shape with some convs forced to a tile family / K cut (diagnostic override `fiunet_debug_force_cfg`):
    python tools/force_cfg_times.py B H W precision [layer:tile:ksplit ...] [--cf 3] [--steps 10]
tile: 0 = choose, 1 = big, 2 = small (64 couts x 8x32 pixels, 64 x 64 wave tiles); ksplit: 0 = choose.  Prints the
default configuration first, then the forced one, stage by stage (HIP events between the stages)."""
import ctypes, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import bench
from ai_based_frame_interpolation_amd import _native
argv = sys.argv[1:]
args = [a for i, a in enumerate(argv) if not a.startswith("--") and not (i > 0 and argv[i - 1] in ("--cf", "--steps"))]
b, h, w, prec = int(args[0]), int(args[1]), int(args[2]), args[3]
forced = [tuple(int(v) for v in a.split(":")) for a in args[4:]]
cf = int(sys.argv[sys.argv.index("--cf") + 1]) if "--cf" in sys.argv else 1
steps = int(sys.argv[sys.argv.index("--steps") + 1]) if "--steps" in sys.argv else 10
dev = torch.device("cuda:0")
m = bench.make_bench_model(prec, frame_channels=cf).to(dev).eval()
g = torch.Generator(device=dev).manual_seed(1)
f1 = torch.rand(b, cf, h, w, device=dev, generator=g) * 2 - 1
f2 = torch.rand(b, cf, h, w, device=dev, generator=g) * 2 - 1
L = _native.lib()
L.fiunet_debug_force_cfg.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int]


def run(tag):
    for _ in range(3): m(f1, f2)
    m._ctx.profile_enable(True)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(steps): m(f1, f2)
    e1.record(); torch.cuda.synchronize()
    _, rows = m._ctx.profile_read(); m._ctx.profile_enable(False)
    ms = e0.elapsed_time(e1) / steps
    print(f"## {tag}: B={b} {w}x{h} {prec} cf={cf}: {ms:.3f} ms/step, {b / ms * 1e3:.1f} frames/s")
    for i, (n, t, fl) in enumerate(rows):
        print(f"  {i:2d} {t:8.3f} ms {fl / (t * 1e-3) / 1e12 if t > 0 else 0:7.1f} TFLOP/s  {n}")
    return m(f1, f2).clone()


ref = run("default")
if forced:
    for layer, tile, k in forced:
        L.fiunet_debug_force_cfg(m._ctx._h, layer, tile, k)
    out = run("forced " + " ".join(f"{l}:{t}:{k}" for l, t, k in forced))
    print(f"max |forced - default| = {float((out - ref).abs().max()):.3e}")
