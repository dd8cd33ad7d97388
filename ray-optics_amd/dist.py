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


class _HitsWindow:
    """a DeviceResult-shaped window [off, off+n) of the exchange buffers, so
    the HITS kernel writes where the collective will read"""

    def __init__(self, xy, st, off, n, cap):
        import torch
        self.R, self.out_mode, self.ld = n, abi.OUT_HITS, cap
        self._xy, self._off = xy, off
        self.seg = xy[:, off:off + n]
        self.status = st[off:off + n]
        self.op = None
        self.fail_surf = None
        self.pupil = None
        self._torch = torch

    def out_struct(self):
        o = abi.Out()
        o.seg = self._xy.data_ptr() + 8 * self._off
        o.op = None
        o.status = self.status.data_ptr()
        o.fail_surf = None
        o.pupil = None
        o.ld = self.ld
        return o


def trace_blocks(engine, blocks, cap, fields, image_pts, num, foc, flags=None,
                 first_surf=1, last_surf=None):
    """HITS trace of this rank's row blocks straight into its exchange buffers
    xy [2, cap] f64 and status [cap] u8 (17 B per ray on the wire); no
    intermediate copies.  Returns (xy, status, rays traced)."""
    import torch
    from .engine import make_opts, make_grid
    N = engine.table.n_ifcs
    if flags is None:
        flags = abi.INTERSECT_OBJ | abi.CHECK_APERTURES | abi.APPLY_VIGNETTING
    last = N - 2 if last_surf is None else last_surf
    dev = getattr(engine, 'device', 'cpu')
    xy_loc = torch.full((2, cap), float('nan'), dtype=torch.float64, device=dev)
    st_loc = torch.full((cap,), 255, dtype=torch.uint8, device=dev)
    off = 0
    for b in blocks:
        opts = make_opts(flags=flags, out_mode=abi.OUT_HITS, first_surf=first_surf,
                         last_surf=last, foc=foc, image_pt=image_pts[b.fi])
        grid = make_grid((-1., -1.), (1., 1.), num, row_begin=b.row_begin,
                         row_count=b.row_count)
        n = b.row_count * num
        out = _HitsWindow(xy_loc, st_loc, off, n, cap)
        res = engine.trace_pupil_grid(fields[b.fi], grid, b.wi, opts, want_pupil=False, out=out)
        if res is not out:          # engines without out= support (test doubles)
            xy_loc[:, off:off + n] = res.seg
            st_loc[off:off + n] = res.status
        off += n
    return xy_loc, st_loc, off


def gather_hits(xy_loc, st_loc, group=None, all_ranks=False):
    """the path's one exchange step: (x, y, status) of every rank to rank 0
    (7 peers -> 7 xGMI links in parallel), or to every rank (all_ranks, a ring
    bound by one link).  Returns (xy_parts, st_parts) lists, None off the root."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    if world == 1:
        return [xy_loc], [st_loc]
    if all_ranks:
        xy_parts = [torch.empty_like(xy_loc) for _ in range(world)]
        st_parts = [torch.empty_like(st_loc) for _ in range(world)]
        dist.all_gather(xy_parts, xy_loc, group=group)
        dist.all_gather(st_parts, st_loc, group=group)
        return xy_parts, st_parts
    xy_parts = [torch.empty_like(xy_loc) for _ in range(world)] if rank == 0 else None
    st_parts = [torch.empty_like(st_loc) for _ in range(world)] if rank == 0 else None
    dist.gather(xy_loc, xy_parts, dst=0, group=group)
    dist.gather(st_loc, st_parts, dst=0, group=group)
    return (xy_parts, st_parts) if rank == 0 else (None, None)


def trace_spot_sharded(engine, fields, image_pts, n_wvls, num, foc, flags=None,
                       group=None, all_ranks=False, first_surf=1, last_surf=None):
    """spot diagrams for every (field, wavelength), sharded over the process
    group.  Each rank traces its row blocks in HITS mode on its own GPU, then
    the hits are gathered.  Returns on rank 0 (every rank if all_ranks) a dict
    {(fi, wi): (xy[num*num, 2], status[num*num])} in the reference's
    i-outer/j-inner order; None elsewhere."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    plan = partition(len(fields), n_wvls, num, world)
    sizes = [sum(b.row_count for b in blocks) * num for blocks in plan]
    cap = max(max(sizes), 1)
    xy_loc, st_loc, _n = trace_blocks(engine, plan[rank], cap, fields, image_pts, num, foc,
                                      flags, first_surf, last_surf)
    xy_parts, st_parts = gather_hits(xy_loc, st_loc, group, all_ranks)
    if xy_parts is None:
        return None

    def to_numpy(t):
        if t.device.type == 'cpu':
            return t.numpy()
        h = torch.empty(t.shape, dtype=t.dtype).pin_memory()
        h.copy_(t, non_blocking=True)
        torch.cuda.synchronize(t.device)
        return h.numpy()

    # reassemble per (field, wavelength) in row order
    out = {}
    for k, blocks in enumerate(plan):
        buf, stb = to_numpy(xy_parts[k]), to_numpy(st_parts[k])
        off = 0
        for b in blocks:
            xy, st = out.setdefault((b.fi, b.wi), (np.full((num * num, 2), np.nan),
                                                   np.full(num * num, 255, dtype=np.uint8)))
            n = b.row_count * num
            r0 = b.row_begin * num
            xy[r0:r0 + n, 0] = buf[0, off:off + n]
            xy[r0:r0 + n, 1] = buf[1, off:off + n]
            st[r0:r0 + n] = stb[off:off + n]
            off += n
    return out
