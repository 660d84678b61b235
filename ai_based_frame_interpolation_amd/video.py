"""Frame-pair sharding of the factor-2 video loop across the GPUs of one node.

The reference has no multi-GPU code; frame pairs are independent forwards, so
the path shards with NO collective inside the forward.  What does move between ranks is
  (1) the uint8 frames from the ingest rank to the others   (point-to-point send/recv), and
  (2) the uint8 interpolated frames back                   (point-to-point send/recv).
On GPUs the process group is NCCL (= RCCL on ROCm), so each transfer is an ncclSend/ncclRecv
over the direct xGMI link between the ingest GPU and that peer: the root's 7 links are used
in parallel, there is no ring.  Weights are not broadcast: every rank loads the same
checkpoint from disk.  The same code runs on gloo/CPU tensors, which is how tests/ cover it.

One process per GPU; rank r owns the contiguous pair range partition_pairs(n, world)[r].
"""
from __future__ import annotations

from typing import Callable, List, Optional, Tuple

import torch
import torch.distributed as dist


def partition_pairs(n_frames: int, world: int) -> List[Tuple[int, int]]:
    """(first_pair, n_pairs) per rank: n_frames-1 pairs in contiguous chunks of
    ceil((n-1)/world); neighbouring ranks overlap by one frame.  Trailing ranks
    may get (start, 0) when there are fewer pairs than ranks."""
    n_pairs = max(n_frames - 1, 0)
    per = -(-n_pairs // world) if n_pairs else 0
    out = []
    for r in range(world):
        s = min(r * per, n_pairs)
        out.append((s, min(per, n_pairs - s)))
    return out


def _p2p(ops):
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()


def scatter_frames(frames: Optional[torch.Tensor], n_frames: int, frame_shape, device,
                   src: int = 0, group=None) -> torch.Tensor:
    """Rank `src` holds `frames` [n_frames, *frame_shape] uint8 on `device`; every rank returns
    its chunk [n_pairs+1, *frame_shape] (empty if it owns no pair)."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    parts = partition_pairs(n_frames, world)
    s, c = parts[rank]
    if rank == src:
        ops = []
        for r, (rs, rc) in enumerate(parts):
            if r != src and rc > 0:
                ops.append(dist.P2POp(dist.isend, frames[rs:rs + rc + 1].contiguous(), r, group))
        _p2p(ops)
        return frames[s:s + c + 1] if c > 0 else frames[:0]
    if c == 0:
        return torch.empty((0,) + tuple(frame_shape), dtype=torch.uint8, device=device)
    buf = torch.empty((c + 1,) + tuple(frame_shape), dtype=torch.uint8, device=device)
    _p2p([dist.P2POp(dist.irecv, buf, src, group)])
    return buf


def gather_middles(local_mid: torch.Tensor, n_frames: int, frame_shape, device, dst: int = 0,
                   group=None) -> Optional[torch.Tensor]:
    """Inverse of scatter_frames for the results: rank `dst` returns [n_frames-1, *frame_shape]."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    parts = partition_pairs(n_frames, world)
    if rank != dst:
        if parts[rank][1] > 0:
            _p2p([dist.P2POp(dist.isend, local_mid.contiguous(), dst, group)])
        return None
    out = torch.empty((max(n_frames - 1, 0),) + tuple(frame_shape), dtype=torch.uint8, device=device)
    ops = []
    for r, (rs, rc) in enumerate(parts):
        if rc == 0:
            continue
        if r == dst:
            out[rs:rs + rc] = local_mid
        else:
            ops.append(dist.P2POp(dist.irecv, out[rs:rs + rc], r, group))
    _p2p(ops)
    return out


def interpolate_video_sharded(pair_fn: Callable[[torch.Tensor, torch.Tensor], torch.Tensor],
                              frames: Optional[torch.Tensor], n_frames: int, frame_shape, device,
                              batch: int = 8, root: int = 0, group=None) -> Optional[torch.Tensor]:
    """factor-2 interpolation of a video held by `root`: scatter frame chunks, run
    `pair_fn(F[i:i+b], F[i+1:i+b+1]) -> M` (uint8 in, uint8 out; on the GPU this is
    FrameInterpolationUNet.forward_u8) on every rank's own pairs, gather the middles.
    Returns the interleaved [2n-1, ...] stack on `root`, None elsewhere."""
    local = scatter_frames(frames, n_frames, frame_shape, device, root, group)
    n_local = max(local.shape[0] - 1, 0)
    mids = torch.empty((n_local,) + tuple(frame_shape), dtype=torch.uint8, device=device)
    for s in range(0, n_local, batch):
        e = min(s + batch, n_local)
        mids[s:e] = pair_fn(local[s:e], local[s + 1:e + 1])
    gathered = gather_middles(mids, n_frames, frame_shape, device, root, group)
    if dist.get_rank(group) != root:
        return None
    out = torch.empty((2 * n_frames - 1,) + tuple(frame_shape), dtype=torch.uint8, device=device)
    out[0::2] = frames
    out[1::2] = gathered
    return out
