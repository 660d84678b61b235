"""Probe (round 6): would the factor-2 video loop gain from two chunks in flight on two streams (two model objects = two
contexts / workspaces, same weights)?  Frame pairs are independent, so overlapping chunk i+1 with the kernel tails of chunk
i is only scheduling.  python tools/two_lane_video_probe.py [n_frames=800] [batch=8]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from ai_based_frame_interpolation_amd import synthetic as S
n = int(sys.argv[1]) if len(sys.argv) > 1 else 800
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 8
dev = torch.device("cuda:0")
m0 = bench.make_bench_model("bf16").to(dev).eval()
m1 = bench.make_bench_model("bf16").to(dev).eval()
m1.load_state_dict(m0.state_dict())
frames = S.moving_frames(0, n, 1080, 1920, device=dev, seed=11).unsqueeze(1)
out = torch.empty((2 * n - 1, 1, 1080, 1920), dtype=torch.uint8, device=dev)
chunks = [(s, min(batch, n - 1 - s)) for s in range(0, n - 1, batch)]


def one_lane():
    for s, c in chunks:
        m0.forward_u8(frames[s:s + c], frames[s + 1:s + c + 1], out=out[2 * s + 1:2 * (s + c):2])


def two_lanes():
    cur = torch.cuda.current_stream(dev)
    lanes = [(m0, torch.cuda.Stream(device=dev)), (m1, torch.cuda.Stream(device=dev))]
    for _, st in lanes:
        st.wait_stream(cur)
    for i, (s, c) in enumerate(chunks):
        m, st = lanes[i & 1]
        with torch.cuda.stream(st):
            m.forward_u8(frames[s:s + c], frames[s + 1:s + c + 1], out=out[2 * s + 1:2 * (s + c):2])
    for _, st in lanes:
        cur.wait_stream(st)


for name, fn in (("one lane", one_lane), ("two lanes", two_lanes), ("one lane", one_lane), ("two lanes", two_lanes)):
    fn(); torch.cuda.synchronize()
    ref = out[1::2][:8].clone() if name == "one lane" else ref
    t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    same = bool(torch.equal(out[1::2][:8], ref))
    print(f"{name}: {(n - 1) / dt:.1f} interpolated frames/s ({dt:.2f} s), first chunk equal to the one-lane result: {same}", flush=True)
