import sys, torch
sys.path.insert(0, '.')
import ai_based_frame_interpolation_amd as P
from ai_based_frame_interpolation_amd import _native as N
ctx = N.Context(0, 1, True)
for flags, name in ((0, "default"), (N.OPT_KEEP_ALL, "keep_all"), (N.OPT_UNFUSED, "unfused")):
    ctx.set_options(flags)
    for (b, h, w) in ((8, 1080, 1920), (1, 2160, 3840), (16, 256, 256), (1, 256, 256)):
        print(name, (b, h, w), "bf16 %.3f GB" % (ctx.workspace_bytes(b, h, w, N.BF16) / 1e9), "fp32 %.3f GB" % (ctx.workspace_bytes(b, h, w, N.FP32) / 1e9))
