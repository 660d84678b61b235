"""Point-to-point transfers for the two multi-rank flows (video.py: frame-pair sharding, tiling.py: row
bands).  The reference has no multi-GPU code; this is build-side plumbing.

Two transports, chosen by the process group's backend and the tensor's device:

  * **nccl (= RCCL) with device tensors** - the production path: `dist.batch_isend_irecv` as it is (one
    grouped ncclSend/ncclRecv launch; `work.wait()` makes the CURRENT stream wait).  Untouched here.
  * **any other backend (gloo) with device tensors** - the one-GPU rehearsal of the tests.  torch's gloo
    `send`/`recv` take `tensor.data_ptr()` and let a host thread read / write that address; for a device
    tensor that is a CPU access to VRAM through the PCIe BAR with NO ordering against any HIP stream
    (round 3's red `test_tiled_forward_two_ranks_on_one_gpu`: the host could read an output band before
    the head kernel had written it).  Such transfers are therefore STAGED through pinned host memory
    with explicit synchronisation on both ends:
        send:  D2H copy on the current stream -> stream.synchronize() -> isend(host tensor);
        recv:  irecv(host tensor); `wait()` blocks the host until it has arrived, then issues the H2D
               copy on the stream that is current at wait() time (the consumer's stream).
    What these rehearsals check is the control flow above the transport (partitioning, issue order,
    rings, which stream consumes what) - not RCCL and not xGMI.
  * CPU tensors (the gloo tests in tests/test_dist.py): passed straight through.
"""
from __future__ import annotations

from typing import List

import torch
import torch.distributed as dist


def is_device_native(group=None) -> bool:
    """True when the group's backend moves device memory itself (RCCL)."""
    return "nccl" in str(dist.get_backend(group)).lower()


class _SendWork:
    def __init__(self, work, host):
        self._work, self._host = work, host

    def wait(self):
        if self._work is not None:      # a second wait() on a gloo work never returns
            self._work.wait()
            self._work = self._host = None
        return True


class _RecvWork:
    def __init__(self, work, host, dst):
        self._work, self._host, self._dst = work, host, dst

    def wait(self):
        if self._work is not None:
            self._work.wait()                             # host: the bytes are in pinned memory
            self._dst.copy_(self._host, non_blocking=True)  # H2D on the consumer's (current) stream;
            self._work = self._host = self._dst = None      # the pinned block is kept alive by torch's
        return True                                         # host allocator until that copy has run


def _peer_kwargs(op: dist.P2POp) -> dict:
    return {("group_dst" if op.op is dist.isend else "group_src"): op.group_peer}


def batch_isend_irecv(ops: List[dist.P2POp]) -> list:
    """`dist.batch_isend_irecv` with the staging described above for device tensors on a backend that
    is not RCCL.  Returns work-like objects with an idempotent `wait()`."""
    if not ops:
        return []
    group = ops[0].group
    if not any(op.tensor.is_cuda for op in ops) or is_device_native(group):
        return dist.batch_isend_irecv(ops)
    works = []
    staged = []
    for op in ops:          # all D2H copies first, then ONE host synchronisation
        host = torch.empty(op.tensor.shape, dtype=op.tensor.dtype, pin_memory=True)
        if op.op is dist.isend:
            host.copy_(op.tensor, non_blocking=True)
        staged.append(host)
    if any(op.op is dist.isend for op in ops):
        for dev in {op.tensor.device for op in ops if op.op is dist.isend and op.tensor.is_cuda}:
            torch.cuda.current_stream(dev).synchronize()
    for op, host in zip(ops, staged):
        w = op.op(host, group=op.group, tag=op.tag, **_peer_kwargs(op))
        works.append(_SendWork(w, host) if op.op is dist.isend else _RecvWork(w, host, op.tensor))
    return works


def p2p(ops: List[dist.P2POp]) -> None:
    """Issue and complete a batch of transfers (on RCCL: the current stream waits for them)."""
    for w in batch_isend_irecv(ops):
        w.wait()
