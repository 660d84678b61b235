"""CPU, world_size 2, gloo: the frame-pair sharding (scatter frames -> per-rank pairs -> gather
middles) is exact for ragged and tiny inputs.  pair_fn is a stand-in integer op: this tests
the host-side partition/communication logic, not the network."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from ai_based_frame_interpolation_amd import video


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _pair_fn(a, b):
    return ((a.to(torch.int32) + b.to(torch.int32) + 1) // 2).to(torch.uint8)


def _worker(rank, world, port, n_frames, shape, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = torch.Generator().manual_seed(5)
        frames = torch.randint(0, 256, (n_frames,) + shape, dtype=torch.uint8, generator=g)
        out = video.interpolate_video_sharded(_pair_fn, frames if rank == 0 else None, n_frames,
                                              shape, torch.device("cpu"), batch=3)
        if rank == 0:
            want = torch.empty((2 * n_frames - 1,) + shape, dtype=torch.uint8)
            want[0::2] = frames
            if n_frames > 1:
                want[1::2] = _pair_fn(frames[:-1], frames[1:])
            q.put(bool(torch.equal(out, want)))
        else:
            assert out is None
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_frames", [2, 3, 8, 11])
def test_sharded_video_world2(n_frames):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_frames, (6, 10), q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True
