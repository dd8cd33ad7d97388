"""Multi-GPU: one process per GPU (torch.distributed; backend "nccl" is RCCL
on ROCm, xGMI between the GPUs of a node).

Rays never interact and the surface table (a few KB) is replicated, so the
index space (field x wavelength x pupil row) is cut into contiguous row blocks,
one run of blocks per rank, and every rank traces its own blocks with no
data-path communication.  The single exchange step of a spot diagram is the
gather of the image-plane hits (x, y, status: 17 B per ray) to the rank that
plots them.  FULL ray packets are never exchanged: they stay resident on the
GPU that traced them.

xGMI is point-to-point: a root gathering from 7 peers receives on 7 links at
once, while a ring all-gather is bound by one link -- so the default is a
gather to rank 0 and ``all_ranks=True`` (all-gather) is opt-in.
"""
from dataclasses import dataclass

import numpy as np

from . import abi


@dataclass(frozen=True)
class Block:
    """rows [row_begin, row_begin+row_count) of the num x num pupil grid of
    (field fi, wavelength wi)"""
    fi: int
    wi: int
    row_begin: int
    row_count: int


def partition(n_fields, n_wvls, num, world):
    """contiguous split of the n_fields*n_wvls*num pupil rows over `world`
    ranks (row counts differ by at most one); returns blocks[rank] = [Block]."""
    total = n_fields * n_wvls * num
    bounds = [(total * k) // world for k in range(world + 1)]
    out = []
    for k in range(world):
        lo, hi = bounds[k], bounds[k + 1]
        blocks = []
        while lo < hi:
            g, row = divmod(lo, num)            # g = fi * n_wvls + wi
            take = min(hi - lo, num - row)
            blocks.append(Block(g // n_wvls, g % n_wvls, row, take))
            lo += take
        out.append(blocks)
    return out


def trace_spot_sharded(engine, fields, image_pts, n_wvls, num, foc, flags=None,
                       group=None, all_ranks=False, first_surf=1, last_surf=None):
    """spot diagrams for every (field, wavelength), sharded over the process
    group.  Each rank traces its row blocks in HITS mode on its own GPU, then
    the hits are gathered.  Returns on rank 0 (every rank if all_ranks) a dict
    {(fi, wi): (xy[num*num, 2], status[num*num])} in the reference's
    i-outer/j-inner order; None elsewhere."""
    import torch
    import torch.distributed as dist
    from .engine import make_opts, make_grid
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    N = engine.table.n_ifcs
    if flags is None:
        flags = abi.INTERSECT_OBJ | abi.CHECK_APERTURES | abi.APPLY_VIGNETTING
    last = N - 2 if last_surf is None else last_surf
    plan = partition(len(fields), n_wvls, num, world)
    sizes = [sum(b.row_count for b in blocks) * num for blocks in plan]
    cap = max(max(sizes), 1)

    # local trace: packed [3, cap] = x, y, status (as f64) so that one
    # collective moves everything
    packed = None
    off = 0
    for b in plan[rank]:
        opts = make_opts(flags=flags, out_mode=abi.OUT_HITS, first_surf=first_surf,
                         last_surf=last, foc=foc, image_pt=image_pts[b.fi])
        grid = make_grid((-1., -1.), (1., 1.), num, row_begin=b.row_begin,
                         row_count=b.row_count)
        res = engine.trace_pupil_grid(fields[b.fi], grid, b.wi, opts, want_pupil=False,
                                      nan_fill=True)
        seg, status = res.seg, res.status
        if packed is None:
            packed = torch.full((3, cap), float('nan'), dtype=torch.float64,
                                device=seg.device)
        n = b.row_count * num
        packed[0:2, off:off + n] = seg
        packed[2, off:off + n] = status.to(torch.float64)
        off += n
    if packed is None:
        dev = getattr(engine, 'device', 'cpu')
        packed = torch.full((3, cap), float('nan'), dtype=torch.float64, device=dev)

    # the exchange step
    if world == 1:
        parts = [packed]
    elif all_ranks:
        parts = [torch.empty_like(packed) for _ in range(world)]
        dist.all_gather(parts, packed, group=group)
    else:
        parts = [torch.empty_like(packed) for _ in range(world)] if rank == 0 else None
        dist.gather(packed, parts, dst=0, group=group)
        if rank != 0:
            return None

    # reassemble per (field, wavelength) in row order
    out = {}
    for k, blocks in enumerate(plan):
        buf = parts[k].cpu().numpy()
        off = 0
        for b in blocks:
            xy, st = out.setdefault((b.fi, b.wi), (np.full((num * num, 2), np.nan),
                                                   np.full(num * num, 255, dtype=np.uint8)))
            n = b.row_count * num
            r0 = b.row_begin * num
            xy[r0:r0 + n, 0] = buf[0, off:off + n]
            xy[r0:r0 + n, 1] = buf[1, off:off + n]
            st[r0:r0 + n] = buf[2, off:off + n].astype(np.uint8)
            off += n
    return out
