"""Per-layer sweep of the conv launch configuration (tile family x K split) on one shape: every candidate of every conv is
forced through the diagnostic override `fiunet_debug_force_cfg` and the STAGE time (conv + finalize pass, HIP events around
the stage) is averaged over `iters` forwards.  The constants of `conv_cost_ns` (csrc/fiunet.hip) come from this table.
    python tools/cfg_sweep.py B H W precision [iters]"""
import ctypes, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import bench
from ai_based_frame_interpolation_amd import _native
b, h, w = (int(v) for v in sys.argv[1:4])
prec = sys.argv[4]
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 20
dev = torch.device("cuda:0")
m = bench.make_bench_model(prec).to(dev).eval()
g = torch.Generator(device=dev).manual_seed(1)
f1 = torch.rand(b, 1, h, w, device=dev, generator=g) * 2 - 1
f2 = torch.rand(b, 1, h, w, device=dev, generator=g) * 2 - 1
ref = m(f1, f2).clone()
L = _native.lib()
L.fiunet_debug_force_cfg.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int]
H = m._ctx._h


def stage_times():
    for _ in range(3): m(f1, f2)
    m._ctx.profile_enable(True)
    for _ in range(iters): m(f1, f2)
    torch.cuda.synchronize()
    _, rows = m._ctx.profile_read()
    m._ctx.profile_enable(False)
    return rows


L.fiunet_debug_force_cfg(H, -1, 0, 0)
base = stage_times()
print(f"# B={b} {w}x{h} {prec}: stage time (us) per forced configuration; '*' = what choose_conv_cfg picks; max |out - chosen| in brackets")
tot = 0.0
for i in range(1, 18):
    chosen = base[i][0]
    seen = {}
    for tile, ks in ((1, (1, 2, 4, 8, 16, 32)), (2, (1, 2, 4, 8, 16, 32)), (3, (1,))):   # 3: the in-workgroup K cut where it applies
        for k in ks:
            L.fiunet_debug_force_cfg(H, -1, 0, 0)
            L.fiunet_debug_force_cfg(H, i, tile, k)
            try:
                rows = stage_times()
            except Exception as e:   # a candidate the launch refuses
                continue
            name = rows[i][0]
            if name in seen: continue
            d = float((m(f1, f2) - ref).abs().max())
            seen[name] = (rows[i][1] * 1e3, d)
    L.fiunet_debug_force_cfg(H, -1, 0, 0)
    best = min(seen.values())[0]
    tot += seen.get(chosen, (base[i][1] * 1e3, 0))[0]
    print(f"stage {i:2d}  chosen {base[i][1] * 1e3:6.1f} us  best {best:6.1f} us")
    for name, (us, d) in sorted(seen.items(), key=lambda kv: kv[1][0]):
        print(f"      {us:7.1f} us {'*' if name == chosen else ' '} {name.replace('conv3x3_mfma_kernel', 'conv')}  [{d:.1e}]")
print(f"sum of chosen stages {tot:.1f} us")
