"""Spatial tiling of one large frame pair into horizontal strips.

The reference has no tiling code; a frame pair is one forward (model/unet.py:84-95).  For frames
whose latency matters (2160x3840) the forward is cut into `n` row bands, one per GPU.  Facts the
plan rests on (measured on the reference): an output pixel depends on inputs within
+-109 pixels; the encoder floor-halves the grid four times; the decoder upsamples with
align_corners=True, i.e. with a scale that depends on the WHOLE image's size.  Hence:

  * bands start at multiples of 16 rows, so the band's pyramid is a window of the image's pyramid;
  * each band carries a halo of 112 input rows (>= 109, multiple of 16) on every cut edge and the
    halo's output rows are discarded ("one-shot input halo", SURVEY 8e option B: the halo is 1-channel
    input, 2 x 112 x W x 4 B per cut, instead of ~22 per-layer exchanges of 64-1024-channel rows);
  * the kernels evaluate the upsample mapping and the F.pad offsets in whole-image coordinates
    (`fiunet_forward_strip`), so the kept rows are the un-tiled result - bit for bit wherever the
    kernels take the same path (the split-K of very small problems can differ; see the tests).

Multi-GPU: one process per GPU, rank r computes band r; point-to-point send/recv only (RCCL over the
direct xGMI links on GPUs, gloo in the CPU tests), no collective and no exchange inside the forward.
Two ways to get a band's input halo to its rank:

  * `forward_tiled_distributed` - the rank that holds the frames sends every band WITH its halo (the
    halo rows cross the root's links twice);
  * `forward_tiled_halo_exchange` (round 5; BASELINE configs[4] "spatial-tile across 4 GPUs with xGMI
    halo exchange") - the root scatters the CORE rows only, or the ranks already hold them (a row-sharded
    decoder), and every rank fetches its 112 halo rows per cut edge from the ranks whose cores contain
    them: its two NEIGHBOURS for bands of at least 112 rows, in one grouped send/recv - 2 x 112 x W
    uint8 pixels per cut and frame over the neighbours' own links, in parallel on every cut.
Either way the halo is INPUT (SURVEY 8e option B: the halo's outputs are recomputed and discarded);
the per-layer activation-row exchange of SURVEY 8e option A (~22 latency-bound messages per forward)
is not built.  Neither path has run over RCCL on more than one GPU (no multi-GPU box in this build).

Wire format: the dtype of the frames.  Video frames are uint8 (`wire=torch.uint8`, SURVEY 8e "keep frames
uint8 on the wire"): bands travel as uint8 - a quarter of the fp32 bytes in both directions - and every
rank applies the reference's pre/post-processing (model/inference.py:31-35, :54-61) on its own device
around its band, so the assembled uint8 frame equals `model.forward_u8` of the whole pair bit for bit.
fp32 tensors (`wire=torch.float32`, the default) travel as they are: any narrower float on the wire would
change the input of the network.
"""
from __future__ import annotations

from typing import Callable, List, NamedTuple, Optional

import torch
import torch.distributed as dist

from . import transport

HALO = 112   # rows; >= the 109-pixel receptive-field radius, multiple of 16
ALIGN = 16   # four 2x2 max-pools


class Strip(NamedTuple):
    core0: int  # first output row this strip is responsible for
    core1: int  # one past its last output row
    ext0: int   # first input row it needs (core0 - halo, clipped to the image)
    ext1: int   # one past the last input row it needs


def strip_plan(height: int, n_strips: int, halo: int = HALO) -> List[Strip]:
    """Cut `height` rows into `n_strips` bands of near-equal size with origins at multiples of 16
    (2160 rows, 4 strips -> origins 0/544/1088/1632 as in SURVEY 8d config 5).  Trailing strips may
    be empty (core0 == core1) when the image has fewer 16-row groups than strips."""
    if height < ALIGN:
        raise ValueError(f"image height {height} < {ALIGN}")
    if halo % ALIGN or halo < 109:
        raise ValueError("halo must be a multiple of 16 and cover the 109-row receptive field")
    groups = -(-height // ALIGN)                 # 16-row groups, the last may be partial
    per = -(-groups // n_strips)
    plan = []
    for i in range(n_strips):
        c0 = min(i * per * ALIGN, height)
        c1 = min((i + 1) * per * ALIGN, height)
        e0 = max(c0 - halo, 0)
        e1 = min(c1 + halo, height)
        plan.append(Strip(c0, c1, e0, e1))
    return plan


StripFn = Callable[[torch.Tensor, torch.Tensor, int, int], torch.Tensor]


def u8_strip_fn(strip_fn: StripFn, pre=None, post=None) -> StripFn:
    """`strip_fn` on uint8 bands: device pre-processing -> fp32 strip forward -> device post-processing
    (the kernels behind `fiunet_preprocess_u8` / `fiunet_postprocess_u8`; `pre` / `post` replace them in the
    CPU tests, which have no device)."""
    if pre is None or post is None:
        from . import _native
        pre, post = pre or _native.preprocess_u8, post or _native.postprocess_u8

    def fn(a, b, y_origin, image_height):
        return post(strip_fn(pre(a), pre(b), y_origin, image_height))
    return fn


def forward_tiled(strip_fn: StripFn, frame1: torch.Tensor, frame2: torch.Tensor, n_strips: int,
                  halo: int = HALO) -> torch.Tensor:
    """All strips on this device, one after the other (bounds the workspace of a huge frame to one
    band's; also the single-GPU check of the multi-GPU path).  `strip_fn(f1_band, f2_band, y_origin,
    image_height)` is `FrameInterpolationUNet.forward_strip`."""
    h = frame1.shape[-2]
    out = torch.empty_like(frame1)
    for s in strip_plan(h, n_strips, halo):
        if s.core1 <= s.core0:
            continue
        band = strip_fn(frame1[..., s.ext0:s.ext1, :].contiguous(),
                        frame2[..., s.ext0:s.ext1, :].contiguous(), s.ext0, h)
        out[..., s.core0:s.core1, :] = band[..., s.core0 - s.ext0:s.core1 - s.ext0, :]
    return out


def _p2p(ops):
    transport.p2p(ops)


def exchange_halos(core1: torch.Tensor, core2: torch.Tensor, plan: List[Strip], rank: int, width: int, device,
                   group=None):
    """Neighbour halo exchange: every rank holds the CORE rows [core0, core1) of both frames (`[B, C, rows, W]`)
    and needs [ext0, ext1).  Rank q sends rank r the rows of q's core that fall inside r's halo; for bands of at
    least `halo` rows those are r's two neighbours only.  One grouped batch of sends and receives per rank, peers
    in rank order on every rank (a consistent issue order, as in video.py).  Returns the two bands
    `[B, C, ext1 - ext0, W]` (the core is copied in place between the received halos)."""
    mine = plan[rank]
    b, c = core1.shape[0], core1.shape[1]
    band1 = torch.empty((b, c, mine.ext1 - mine.ext0, width), dtype=core1.dtype, device=device)
    band2 = torch.empty_like(band1)
    band1[..., mine.core0 - mine.ext0:mine.core1 - mine.ext0, :] = core1
    band2[..., mine.core0 - mine.ext0:mine.core1 - mine.ext0, :] = core2
    ops, landing = [], []
    for q, other in enumerate(plan):
        if q == rank or other.core1 <= other.core0 or mine.core1 <= mine.core0:
            continue
        # what q needs from my core
        s0, s1 = max(other.ext0, mine.core0), min(other.ext1, mine.core1)
        if s1 > s0:
            for core in (core1, core2):
                ops.append(dist.P2POp(dist.isend, core[..., s0 - mine.core0:s1 - mine.core0, :].contiguous(), q, group))
        # what I need from q's core
        r0, r1 = max(mine.ext0, other.core0), min(mine.ext1, other.core1)
        if r1 > r0:
            for band in (band1, band2):
                buf = torch.empty((b, c, r1 - r0, width), dtype=core1.dtype, device=device)
                ops.append(dist.P2POp(dist.irecv, buf, q, group))
                landing.append((band, r0 - mine.ext0, r1 - mine.ext0, buf))
    _p2p(ops)
    for band, a0, a1, buf in landing:
        band[..., a0:a1, :] = buf
    return band1, band2


def _agree_or_raise(problem: Optional[str], device, group) -> None:
    """Argument check that must fail on EVERY rank or on none: a rank that raised alone would leave its peers inside
    receives that never complete (over RCCL a hang, not an error).  One all-reduce (MAX) of a flag - control plane, a few
    bytes, before any point-to-point operation of the call - then the same exception everywhere."""
    flag = torch.tensor([1 if problem else 0], dtype=torch.int32, device=device)
    dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=group)
    if int(flag.item()):
        raise ValueError(problem or "a peer rank rejected its arguments (see that rank's message)")


def forward_tiled_halo_exchange(strip_fn: StripFn, frame1: Optional[torch.Tensor], frame2: Optional[torch.Tensor],
                                shape, device, root: int = 0, halo: int = HALO, group=None,
                                wire: torch.dtype = torch.float32, pre=None, post=None,
                                cores=None) -> Optional[torch.Tensor]:
    """Spatial tiling with a neighbour-to-neighbour halo exchange.  Rank `root` holds the pair `[B, C, H, W]`
    (dtype `wire`) and scatters the CORE rows of every band - or, `cores=(core1, core2)`, every rank already
    holds its own core rows and `frame1` / `frame2` are ignored; the ranks exchange their halos among themselves
    (`exchange_halos`), compute their bands, and `root` gathers and returns the assembled output (the others None).
    Same result as `forward_tiled_distributed`, bit for bit."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    if wire not in (torch.float32, torch.uint8):
        raise ValueError("wire must be torch.float32 or torch.uint8")
    if wire == torch.uint8:
        strip_fn = u8_strip_fn(strip_fn, pre, post)
    b, c, h, w = shape
    plan = strip_plan(h, world, halo)
    mine = plan[rank]
    have = mine.core1 > mine.core0
    # every buffer that goes on a link must have the size its peer posted: checked on all ranks before the first transfer
    problem = None
    if cores is not None:
        want = (b, c, max(mine.core1 - mine.core0, 0), w)
        for t in cores:
            if tuple(t.shape) != want or t.dtype != wire:
                problem = f"rank {rank}: core rows are {tuple(t.shape)} {t.dtype}, the plan says {want} {wire}"
    elif rank == root:
        for t in (frame1, frame2):
            if t is None or tuple(t.shape) != (b, c, h, w) or t.dtype != wire:
                problem = (f"root's frames are {None if t is None else (tuple(t.shape), t.dtype)}, "
                           f"expected {(b, c, h, w)} in the wire dtype {wire}")
    _agree_or_raise(problem, device, group)
    # 1. core rows: root -> ranks (no halo on these links)
    if cores is not None:
        c1, c2 = cores
    elif rank == root:
        ops = []
        for r, s in enumerate(plan):
            if r != root and s.core1 > s.core0:
                ops.append(dist.P2POp(dist.isend, frame1[..., s.core0:s.core1, :].contiguous(), r, group))
                ops.append(dist.P2POp(dist.isend, frame2[..., s.core0:s.core1, :].contiguous(), r, group))
        _p2p(ops)
        c1 = frame1[..., mine.core0:mine.core1, :].contiguous()
        c2 = frame2[..., mine.core0:mine.core1, :].contiguous()
    else:
        c1 = torch.empty((b, c, max(mine.core1 - mine.core0, 0), w), dtype=wire, device=device)
        c2 = torch.empty_like(c1)
        if have:
            _p2p([dist.P2POp(dist.irecv, c1, root, group), dist.P2POp(dist.irecv, c2, root, group)])
    # 2. halos: rank <-> neighbours
    core = None
    if have:
        f1, f2 = exchange_halos(c1, c2, plan, rank, w, device, group)
        band = strip_fn(f1, f2, mine.ext0, h)
        core = band[..., mine.core0 - mine.ext0:mine.core1 - mine.ext0, :].contiguous()
    # 3. output cores: ranks -> root
    if rank != root:
        if core is not None:
            _p2p([dist.P2POp(dist.isend, core, root, group)])
        return None
    out = torch.empty((b, c, h, w), dtype=wire, device=device)
    ops, bufs = [], []
    for r, s in enumerate(plan):
        if s.core1 <= s.core0:
            continue
        if r == root:
            out[..., s.core0:s.core1, :] = core
        else:
            buf = torch.empty((b, c, s.core1 - s.core0, w), dtype=wire, device=device)
            bufs.append((s, buf))
            ops.append(dist.P2POp(dist.irecv, buf, r, group))
    _p2p(ops)
    for s, buf in bufs:
        out[..., s.core0:s.core1, :] = buf
    return out


def forward_tiled_distributed(strip_fn: StripFn, frame1: Optional[torch.Tensor],
                              frame2: Optional[torch.Tensor], shape, device, root: int = 0,
                              halo: int = HALO, group=None, wire: torch.dtype = torch.float32,
                              pre=None, post=None) -> Optional[torch.Tensor]:
    """Rank `root` holds the pair `[B, C, H, W]` (`shape`) in dtype `wire`; rank r computes strip r of
    world_size strips; `root` returns the assembled `[B, C, H, W]` output (dtype `wire`), the others None.
    wire = torch.uint8: uint8 frames in, uint8 frame out, uint8 on the wire (see the module docstring);
    `strip_fn` is still the fp32 `forward_strip`."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    if wire not in (torch.float32, torch.uint8):
        raise ValueError("wire must be torch.float32 or torch.uint8")
    b, c, h, w = shape
    problem = None
    if rank == root:
        for t in (frame1, frame2):
            if t is None or tuple(t.shape) != (b, c, h, w) or t.dtype != wire:
                problem = (f"root's frames are {None if t is None else (tuple(t.shape), t.dtype)}, "
                           f"expected {(b, c, h, w)} in the wire dtype {wire}")
    _agree_or_raise(problem, device, group)   # (every rank raises, none is left inside a receive)
    if wire == torch.uint8:
        strip_fn = u8_strip_fn(strip_fn, pre, post)
    plan = strip_plan(h, world, halo)
    mine = plan[rank]
    # 1. input bands (+halo): root -> ranks
    if rank == root:
        ops = []
        for r, s in enumerate(plan):
            if r != root and s.core1 > s.core0:
                ops.append(dist.P2POp(dist.isend, frame1[..., s.ext0:s.ext1, :].contiguous(), r, group))
                ops.append(dist.P2POp(dist.isend, frame2[..., s.ext0:s.ext1, :].contiguous(), r, group))
        _p2p(ops)
        f1 = frame1[..., mine.ext0:mine.ext1, :].contiguous()
        f2 = frame2[..., mine.ext0:mine.ext1, :].contiguous()
    elif mine.core1 > mine.core0:
        f1 = torch.empty((b, c, mine.ext1 - mine.ext0, w), dtype=wire, device=device)
        f2 = torch.empty_like(f1)
        _p2p([dist.P2POp(dist.irecv, f1, root, group), dist.P2POp(dist.irecv, f2, root, group)])
    # 2. this rank's band
    core = None
    if mine.core1 > mine.core0:
        band = strip_fn(f1, f2, mine.ext0, h)
        core = band[..., mine.core0 - mine.ext0:mine.core1 - mine.ext0, :].contiguous()
    # 3. output bands: ranks -> root
    if rank != root:
        if core is not None:
            _p2p([dist.P2POp(dist.isend, core, root, group)])
        return None
    out = torch.empty((b, c, h, w), dtype=wire, device=device)
    ops, bufs = [], []
    for r, s in enumerate(plan):
        if s.core1 <= s.core0:
            continue
        if r == root:
            out[..., s.core0:s.core1, :] = core
        else:
            buf = torch.empty((b, c, s.core1 - s.core0, w), dtype=wire, device=device)
            bufs.append((s, buf))
            ops.append(dist.P2POp(dist.irecv, buf, r, group))
    _p2p(ops)
    for s, buf in bufs:
        out[..., s.core0:s.core1, :] = buf
    return out
