"""Frame-pair sharding of the factor-2 video loop across the GPUs of one node.

The reference has no multi-GPU code; frame pairs are independent forwards, so
the path shards with NO collective inside the forward.  What does move between ranks is
  (1) the uint8 frames from the ingest rank to the others   (point-to-point send/recv), and
  (2) the uint8 interpolated frames back                   (point-to-point send/recv).
On GPUs the process group is NCCL (= RCCL on ROCm), so each transfer is an ncclSend/ncclRecv
over the direct xGMI link between the ingest GPU and that peer: the root's 7 links are used
in parallel, there is no ring.  On any other backend (gloo: the tests) device tensors are staged through
pinned host memory with explicit stream synchronisation (`transport.py`); CPU tensors go as they are.
Weights: every rank normally loads the same checkpoint from disk (no collective at all); `broadcast_model_weights` is there for ranks that do not have the file - ONE
broadcast of the flattened state (69 MB fp32).  The same code runs on gloo/CPU
tensors, which is how tests/ cover it.

One process per GPU; rank r owns the contiguous pair range partition_pairs(n, world)[r].

Transfers are pipelined in sub-batches of `batch` pairs ("chunk into sub-batches of
8-16 frames to overlap with compute"): while a rank forwards sub-batch j, sub-batch j+1 is already
arriving and the middles of sub-batch j-1 are on their way back.  The transfers are issued from a
side stream so that RCCL's send/recv kernels do not queue behind the conv kernels of the compute
stream; events order them against the forwards that produce / consume the buffers.  One 9-frame
1080p message is 18.7 MB (~0.12 ms on one xGMI link) against 17-18 ms of compute per sub-batch,
so the wire time is hidden entirely; what is left exposed is the first scatter and the last gather.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Tuple

import torch
import torch.distributed as dist

from . import transport


def broadcast_model_weights(model, src: int = 0, group=None) -> None:
    """one broadcast of the checkpoint from rank `src` to every rank (ncclBroadcast on
    GPUs, 17.27 M parameters + BatchNorm statistics = 69 MB fp32), for ranks that cannot read the file.
    All floating-point tensors of the state dict travel as ONE flat fp32 buffer in state-dict order
    (the int64 `num_batches_tracked` counters are not used by the forward and stay local); every rank then
    loads the received values, which marks the module's device weights stale (re-uploaded on the next
    forward).  Every rank must have constructed the same architecture."""
    sd = model.state_dict()
    keys = [k for k, v in sd.items() if v.is_floating_point()]
    dev = next(model.parameters()).device
    flat = torch.cat([sd[k].detach().reshape(-1).to(device=dev, dtype=torch.float32) for k in keys])
    dist.broadcast(flat, src=src, group=group)
    if dist.get_rank(group) != src:
        out, o = {}, 0
        for k in keys:
            n = sd[k].numel()
            out[k] = flat[o:o + n].reshape(sd[k].shape).to(sd[k].dtype)
            o += n
        model.load_state_dict(out, strict=False)


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


def sub_batches(n_pairs: int, batch: int) -> List[Tuple[int, int]]:
    """(offset, count) of the sub-batches a rank's `n_pairs` pairs are moved and forwarded in."""
    return [(s, min(batch, n_pairs - s)) for s in range(0, n_pairs, batch)]


def _p2p(ops):
    transport.p2p(ops)


def scatter_frames(frames: Optional[torch.Tensor], n_frames: int, frame_shape, device,
                   src: int = 0, group=None) -> torch.Tensor:
    """Rank `src` holds `frames` [n_frames, *frame_shape] uint8 on `device`; every rank returns
    its chunk [n_pairs+1, *frame_shape] (empty if it owns no pair).  One message per peer."""
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


class _Streams:
    """Compute stream = the caller's current stream; transfers go on a side stream.  On CPU
    tensors (gloo, the tests) every method is a no-op and the calls simply run in program order."""

    def __init__(self, device):
        self.cuda = torch.device(device).type == "cuda"
        if self.cuda:
            self.compute = torch.cuda.current_stream(device)
            self.comm = torch.cuda.Stream(device=device)
            self.comm.wait_stream(self.compute)  # the frames on root were produced on `compute`

    def on_comm(self):
        return torch.cuda.stream(self.comm) if self.cuda else _Null()

    def comm_after_compute(self):
        """Transfers issued from now on wait for everything already queued on the compute stream."""
        if self.cuda:
            self.comm.wait_stream(self.compute)

    def compute_after(self, works):
        """The compute stream waits for these transfers (a wait() on an NCCL work makes the
        CURRENT stream wait for it; on gloo it blocks the host, which is what a CPU tensor needs, and
        for a staged device tensor the H2D copy is then queued on this, the consuming, stream)."""
        for w in works:
            w.wait()

    def finish(self, works):
        if self.cuda:
            with torch.cuda.stream(self.comm):
                for w in works:
                    w.wait()
            self.compute.wait_stream(self.comm)
        else:
            for w in works:
                w.wait()


class _Null:
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


def interpolate_video_sharded(pair_fn: Callable[[torch.Tensor, torch.Tensor], torch.Tensor],
                              frames: Optional[torch.Tensor], n_frames: int, frame_shape, device,
                              batch: int = 8, root: int = 0, group=None,
                              out: Optional[torch.Tensor] = None,
                              trace: Optional[list] = None) -> Optional[torch.Tensor]:
    """factor-2 interpolation of a video held by `root` (uint8 `[n_frames, *frame_shape]` on
    `device`): every rank forwards its own contiguous pair range with
    `pair_fn(F[i:i+b], F[i+1:i+b+1]) -> M` (uint8 in, uint8 out; on the GPU this is
    FrameInterpolationUNet.forward_u8), sub-batch by sub-batch, with the scatter of sub-batch j+1
    and the gather of sub-batch j-1 in flight while sub-batch j is being forwarded.
    Returns the interleaved `[2n-1, ...]` stack on `root` (written into `out` if given), None
    elsewhere.  Every rank issues its sends/recvs towards a given peer in the same order
    (scatter 0, scatter 1, gather 0, scatter 2, gather 1, ...), so the pipeline cannot deadlock;
    `trace`, if given, receives one `(op, peer, kind, j, frames)` tuple per issued transfer in issue
    order (tests/test_dist.py compares the two ends of every link).

    Buffers and streams (all of it a no-op on CPU tensors).  Two streams touch every transfer buffer:
    the comm stream (where the send/recv is issued and, on root, where received middles are
    interleaved into `out`) and the compute stream (where `pair_fn` reads the frames and writes the
    middles).  Nothing here relies on the caching allocator for ordering:
      * a non-root rank receives into a RING of three pre-allocated `[batch+1, ...]` buffers (the recv
        of sub-batch j+1 is issued while sub-batch j-1 may still be running, so three are live at
        most); before a slot is received into, the comm stream waits for the compute stream, i.e. for
        the forward that last read that slot;
      * root receives middles into per-peer staging buffers from a ring of two sets, allocated up
        front; they are only ever touched from the comm stream, in its order;
      * a rank's middles are allocated by `pair_fn` on the compute stream; the comm stream waits for
        the compute stream before the send and the tensor is `record_stream`-ed on the comm stream
        before its reference is dropped."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    shape = tuple(frame_shape)
    parts = partition_pairs(n_frames, world)
    subs = [sub_batches(c, batch) for _, c in parts]
    nsteps = max((len(s) for s in subs), default=0)
    st = _Streams(device)
    is_root = rank == root
    if is_root:
        if out is None:
            out = torch.empty((2 * n_frames - 1,) + shape, dtype=torch.uint8, device=device)
        out[0::2] = frames
    my_first, my_cnt = parts[rank]
    my_subs = subs[rank]
    mids = {}       # sub-batch j's middles (kept until their send has been issued)
    pending = []    # works not yet known to be complete
    RING = 3
    recv_ring = stage_ring = None
    # pre-allocated transfer buffers.  They come from the compute stream's pool, so a kernel queued there
    # earlier may still be using the same bytes: every transfer INTO them is issued after a
    # comm_after_compute() (scatter_step / gather_step), which orders it behind all of that
    if not is_root and my_subs:
        recv_ring = [torch.empty((batch + 1,) + shape, dtype=torch.uint8, device=device)
                     for _ in range(min(RING, len(my_subs)))]
    if is_root and world > 1:
        stage_ring = [{r: torch.empty((batch,) + shape, dtype=torch.uint8, device=device)
                       for r in range(world) if r != root and subs[r]} for _ in range(2)]

    def note(op, peer, kind, j, n):
        if trace is not None:
            trace.append((op, peer, kind, j, n))

    def scatter_step(j):
        """root -> every peer that has a j-th sub-batch (one grouped launch: 7 links in parallel)."""
        ops = []
        if is_root:
            for r in range(world):
                if r != root and j < len(subs[r]):
                    o, c = subs[r][j]
                    a = parts[r][0] + o
                    ops.append(dist.P2POp(dist.isend, frames[a:a + c + 1], r, group))
                    note("send", r, "scatter", j, c + 1)
        elif j < len(my_subs):
            o, c = my_subs[j]
            # slot j % RING was last read by the forward of sub-batch j - RING, queued on the compute
            # stream at least two loop iterations ago: order the receive after it
            st.comm_after_compute()
            ops.append(dist.P2POp(dist.irecv, recv_ring[j % RING][:c + 1], root, group))
            note("recv", root, "scatter", j, c + 1)
        if not ops:
            return []
        with st.on_comm():
            return transport.batch_isend_irecv(ops)

    def gather_step(j):
        """every peer's j-th middles -> root, then (comm stream) into the interleaved output."""
        ops = []
        if is_root:
            for r in range(world):
                if r != root and j < len(subs[r]):
                    o, c = subs[r][j]
                    a = parts[r][0] + o
                    # odd rows of `out` are strided views; receive into a contiguous staging
                    # tensor and let the comm stream interleave it afterwards.  Set j % 2 was last
                    # used by gather j - 2, whose interleave copy is earlier on this same stream.
                    stage = stage_ring[j % 2][r][:c]
                    ops.append((dist.P2POp(dist.irecv, stage, r, group), (a, c, stage)))
                    note("recv", r, "gather", j, c)
        elif j < len(my_subs):
            ops.append((dist.P2POp(dist.isend, mids[j], root, group), None))
            note("send", root, "gather", j, int(mids[j].shape[0]))
        if not ops:
            return []
        st.comm_after_compute()  # the middles of step j are queued on the compute stream
        with st.on_comm():
            works = transport.batch_isend_irecv([op for op, _ in ops])
            if is_root:
                for w in works:
                    w.wait()  # comm stream waits for the receives, then interleaves
                for _, (a, c, stage) in ops:
                    out[2 * a + 1:2 * (a + c):2] = stage
                return []  # already waited for (a second wait() on a gloo work never returns)
            if st.cuda:
                mids[j].record_stream(st.comm)  # allocated on the compute stream, read by the send
        return works

    inflight = {0: scatter_step(0)} if nsteps else {}
    for j in range(nsteps):
        if j + 1 < nsteps:
            inflight[j + 1] = scatter_step(j + 1)
        works = inflight.pop(j, [])
        if is_root:
            pending += works             # root's own forwards do not depend on its sends
        else:
            st.compute_after(works)      # sub-batch j has arrived
        if j < len(my_subs):
            o, c = my_subs[j]
            if is_root:
                src = frames[my_first + o:my_first + o + c + 1]
            else:
                src = recv_ring[j % RING][:c + 1]
            if is_root:
                a = my_first + o
                dst = out[2 * a + 1:2 * (a + c):2]
                if getattr(pair_fn, "accepts_out", False):   # (inference.sequence_pair_fn: written in place by the fused head)
                    pair_fn(src[:c], src[1:c + 1], out=dst)
                else:
                    dst.copy_(pair_fn(src[:c], src[1:c + 1]))
            else:
                mids[j] = pair_fn(src[:c], src[1:c + 1])
        pending += gather_step(j)
        mids.pop(j - 1, None)  # its send was issued one step ago (and record_stream-ed on the comm stream)
    st.finish(pending)
    if st.cuda:  # the rings go back to the allocator of the stream that is current here (compute)
        for t in (recv_ring or []):
            t.record_stream(st.comm)
        for d in (stage_ring or []):
            for t in d.values():
                t.record_stream(st.comm)
    return out if is_root else None
