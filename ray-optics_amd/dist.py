"""Multi-GPU: one process per GPU (torch.distributed; backend "nccl" is RCCL
on ROCm, xGMI between the GPUs of a node).

Rays never interact and the surface table (a few KB) is replicated, so the
index space (field x wavelength x pupil row) is cut into contiguous row blocks,
one run of blocks per rank, and every rank traces its own blocks with no
data-path communication (``partition``; ``by='field'`` keeps whole fields
together -- BASELINE configs[3]'s "shard-by-field").  FULL ray packets are never
exchanged: they stay resident on the GPU that traced them.

What a spot diagram's consumer needs (``SequentialModel.trace_grid(spot, ...,
form='list', append_if_none=False)``, rayoptics/seq/sequential.py:1058-1085)
is, per (field, wavelength), the (R_ok, 2) array of the rays that got through,
in ray order.  Every rank therefore traces its blocks in ROX_OUT_HITS_COMPACT
mode with ROX_HITS_APPEND: survivors only, packed in ray order, block after
block, into one buffer -- 16 B per surviving ray, nothing for a blocked one --
with the per-block counts kept on the device.  Ranks own contiguous runs of
the global block order, so the concatenation of the ranks' buffers *is* the
global order and each (field, wavelength) is one contiguous slice of it.

Two ways to get the pairs into the consumer's host memory, both built:

``exchange='rccl'``   the path's one exchange step: the per-block counts go
    round in a small all-gather, then every rank sends its packed pairs to
    rank 0 (grouped send / recv: 7 peers -> 7 xGMI links in parallel; a ring
    all-gather would be bound by one link), which receives them at their final
    offsets in one device buffer and copies that to pinned host memory -- over
    rank 0's single PCIe link.
``exchange='host'``   no xGMI step: every rank's trace kernel writes its packed
    pairs straight into its own slice of one pinned, shared host segment (the
    consumer maps the same segment), so N PCIe links carry the data while the
    kernels run, and what remains is the count exchange.

C5 arithmetic (9 x 5 x 2048^2 = 188.7 M rays, ~73 % through -> 2.2 GB of pairs):
rccl = 0.28 GB per peer over xGMI (~153 GB/s per link: ~2 ms) + 2.2 GB over one
PCIe Gen5 x16 link (~55 GB/s: ~40 ms); host = 0.28 GB per rank over its own
link (~5 ms), overlapped with ~9 ms of kernels.  The (x, y, status) gather of
round 2 moved 3.2 GB incl. 27 % NaN padding through the same single link.
"""
import os
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


def partition(n_fields, n_wvls, num, world, by='rows'):
    """blocks[rank] = [Block]: a contiguous split of the global block order
    (field-major, then wavelength, then pupil row) over `world` ranks.

    by='rows'   row counts differ by at most one (a grid may be cut between two
                ranks)
    by='field'  whole fields per rank, contiguous, as even as the count allows
                (5 fields over 4 ranks: 2/1/1/1); ranks beyond the field count
                get nothing"""
    out = []
    if by == 'field':
        bounds = [(n_fields * k + world - 1) // world for k in range(world + 1)]
        # (ceil split: the earlier ranks take the extra fields)
        bounds = [min(b, n_fields) for b in bounds]
        for k in range(world):
            out.append([Block(fi, wi, 0, num) for fi in range(bounds[k], bounds[k + 1])
                        for wi in range(n_wvls)])
        return out
    if by != 'rows':
        raise ValueError(f'partition by {by!r}')
    total = n_fields * n_wvls * num
    bounds = [(total * k) // world for k in range(world + 1)]
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


def rays_of(blocks, num):
    return sum(b.row_count for b in blocks) * num


# ---------------------------------------------------------------- per-rank trace
def trace_blocks(engine, blocks, num, fields, image_pts, foc, flags=None, first_surf=1,
                 last_surf=None, cap=None, dest=None):
    """ROX_OUT_HITS_COMPACT | ROX_HITS_APPEND trace of this rank's row blocks into
    one packed buffer (``engine.hits_pack``).  Nothing is synchronised: the
    launches are enqueued back to back, the running count lives on the device.
    ``dest`` = (pointer, capacity in pairs) of device-visible memory to write
    into instead of a fresh HBM buffer (the shared host segment of
    ``exchange='host'``).  Returns the pack; ``pack.counts()`` synchronises and
    gives the survivors per block."""
    from .engine import make_opts, make_grid
    N = engine.table.n_ifcs
    if flags is None:
        flags = abi.INTERSECT_OBJ | abi.CHECK_APERTURES | abi.APPLY_VIGNETTING
    last = N - 2 if last_surf is None else last_surf
    n_rays = rays_of(blocks, num)
    pack = engine.hits_pack(max(n_rays if cap is None else cap, 1), max(len(blocks), 1), dest=dest)
    for b in blocks:
        opts = make_opts(flags=flags | abi.HITS_APPEND, out_mode=abi.OUT_HITS_COMPACT,
                         first_surf=first_surf, last_surf=last, foc=foc, image_pt=image_pts[b.fi])
        grid = make_grid((-1., -1.), (1., 1.), num, row_begin=b.row_begin, row_count=b.row_count)
        engine.trace_pupil_grid_hits_append(fields[b.fi], grid, b.wi, opts, pack)
    return pack


# ---------------------------------------------------------------- exchange
def _group_info(group):
    import torch.distributed as dist
    if not dist.is_initialized():
        return 1, 0, 'none'
    return dist.get_world_size(group), dist.get_rank(group), dist.get_backend(group)


def _wire(t, backend):
    """gloo carries host tensors (CPU tests, 1-GPU rehearsals); RCCL device tensors"""
    return t.cpu() if (backend != 'nccl' and t.is_cuda) else t


def exchange_counts(counts_local, plan, group=None, device='cpu'):
    """survivors per block of every rank -> every rank: a [world, max_blocks] int64
    all-gather (a few hundred bytes).  Returns counts[rank] = int64 array."""
    import torch
    import torch.distributed as dist
    world, _rank, backend = _group_info(group)
    if world == 1:
        return [np.asarray(counts_local, dtype=np.int64)]
    width = max(max(len(b) for b in plan), 1)
    mine = torch.zeros(width, dtype=torch.int64)
    mine[:len(counts_local)] = torch.as_tensor(np.asarray(counts_local, dtype=np.int64))
    if backend == 'nccl':
        mine = mine.to(device)
    allc = torch.empty(world * width, dtype=torch.int64, device=mine.device)
    dist.all_gather_into_tensor(allc, mine, group=group)
    allc = allc.cpu().numpy().reshape(world, width)
    return [allc[k, :len(plan[k])].copy() for k in range(world)]


def gather_packed(xy_local, n_local, totals, group=None, dst=0):
    """the path's one data exchange: every rank's packed pairs xy_local[:n_local]
    ([*, 2] f64) to rank `dst`, received at their final offsets (prefix sums of
    `totals`) in one [sum(totals), 2] tensor -- grouped point-to-point, so that the
    root's peers use their own xGMI links concurrently.  Returns that tensor on the
    root, None elsewhere."""
    import torch
    import torch.distributed as dist
    world, rank, backend = _group_info(group)
    totals = [int(t) for t in totals]
    if world == 1:
        return xy_local[:n_local]
    mine = _wire(xy_local[:n_local], backend)
    if rank != dst:
        if n_local:
            for w in dist.batch_isend_irecv([dist.P2POp(dist.isend, mine.contiguous(), dst, group)]):
                w.wait()
        return None
    out = torch.empty((sum(totals), 2), dtype=torch.float64, device=mine.device)
    offs = np.concatenate([[0], np.cumsum(totals)])
    ops = []
    for k in range(world):
        if k == dst:
            out[offs[k]:offs[k + 1]].copy_(mine)
        elif totals[k]:
            ops.append(dist.P2POp(dist.irecv, out[offs[k]:offs[k + 1]], k, group))
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    return out


def to_pinned_host(t):
    """one D2H copy of a device tensor into pooled pinned memory; the NumPy array
    returned views that memory (CPU tensors are viewed as they are)"""
    import torch
    if t.device.type == 'cpu':
        return t.numpy()
    from .engine import _pool, _np_dtypes, _NP_DTYPES
    _np_dtypes(torch)
    lease = _pool.take(torch, max(t.numel() * t.element_size(), 1))
    dst = lease.tensor[:t.numel() * t.element_size()].view(t.dtype).view(t.shape)
    dst.copy_(t, non_blocking=True)
    torch.cuda.current_stream(t.device).synchronize()
    return lease.array(t.shape, _NP_DTYPES[t.dtype])


def spot_views(host_xy, plan, counts, rank_offsets=None):
    """{(fi, wi): (R_ok, 2) array} over one host buffer holding every rank's packed
    pairs.  ``rank_offsets`` None: the buffer is the plain concatenation (rccl
    exchange) and every array is a zero-copy slice; otherwise rank k's pairs start
    at rank_offsets[k] (the slices of the shared host segment) and only a grid cut
    between two ranks is concatenated."""
    out = {}
    pieces = {}
    pos = 0
    for k, blocks in enumerate(plan):
        if rank_offsets is not None:
            pos = int(rank_offsets[k])
        for b, n in zip(blocks, counts[k]):
            pieces.setdefault((b.fi, b.wi), []).append((pos, int(n)))
            pos += int(n)
    for key, segs in pieces.items():
        merged = [list(segs[0])]
        for p, n in segs[1:]:
            if p == merged[-1][0] + merged[-1][1]:
                merged[-1][1] += n
            else:
                merged.append([p, n])
        if len(merged) == 1:
            out[key] = host_xy[merged[0][0]:merged[0][0] + merged[0][1]]
        else:
            out[key] = np.concatenate([host_xy[p:p + n] for p, n in merged])
    return out


# ---------------------------------------------------------------- shared host segment
class HostSegment:
    """One pinned host segment shared by the ranks of a node: rank k's kernels write
    their packed pairs into slice k (capacity = its ray count), the consumer reads
    the same pages.  Backed by a file in /dev/shm (or ``dir``) that every rank
    maps MAP_SHARED and registers with the HIP runtime (``rox_pin_host_memory``)."""

    def __init__(self, engine, name, caps, rank, create, dir=None):
        import torch
        self.caps = [max(int(c), 0) for c in caps]
        self.offsets = np.concatenate([[0], np.cumsum(self.caps)]).astype(np.int64)
        self.nbytes = max(int(self.offsets[-1]) * 16, 16)
        d = dir or os.environ.get('ROX_SHM_DIR') or ('/dev/shm' if os.path.isdir('/dev/shm') else '/tmp')
        self.path = os.path.join(d, name)
        if create:
            st = os.statvfs(d)
            if st.f_bavail * st.f_frsize < self.nbytes + (64 << 20):
                raise OSError(f'{d} has {st.f_bavail * st.f_frsize >> 20} MiB free, '
                              f'the segment needs {self.nbytes >> 20} MiB')
            with open(self.path, 'wb') as f:
                f.truncate(self.nbytes)
        self.rank = rank
        self._t = torch.from_file(self.path, shared=True, size=self.nbytes, dtype=torch.uint8)
        self._engine = engine
        self.dev_ptr = engine.pin_host_memory(self._t.data_ptr(), self.nbytes)
        self.array = np.frombuffer(memoryview(self._t.numpy()), dtype=np.float64).reshape(-1, 2)

    def dest(self, rank=None):
        """(device-visible pointer, capacity in pairs) of a rank's slice"""
        k = self.rank if rank is None else rank
        return self.dev_ptr + 16 * int(self.offsets[k]), self.caps[k]

    def close(self, unlink=False):
        if self._t is not None:
            try:
                self._engine.unpin_host_memory(self._t.data_ptr())
            except Exception:
                pass
            self.array = None
            self._t = None
        if unlink:
            try:
                os.unlink(self.path)
            except OSError:
                pass


# ---------------------------------------------------------------- the sharded spot diagram
def trace_spot_sharded(engine, fields, image_pts, n_wvls, num, foc, flags=None, group=None,
                       first_surf=1, last_surf=None, by='rows', exchange='rccl', segment=None,
                       timings=None):
    """Spot diagrams for every (field, wavelength), sharded over the process
    group: each rank traces its row blocks on its own GPU (packed hits), the
    pairs reach rank 0's host memory by the chosen exchange.  Returns on rank 0
    {(fi, wi): (R_ok, 2) float64 array in the reference's i-outer/j-inner ray
    order}; None on the other ranks.  ``timings`` (a dict) receives the phases in
    milliseconds: trace (launch -> counts on the host), counts, gather, d2h,
    reassembly."""
    import time
    import torch
    world, rank, backend = _group_info(group)
    plan = partition(len(fields), n_wvls, num, world, by)
    t = [time.perf_counter()]

    def lap():
        t.append(time.perf_counter())
        return (t[-1] - t[-2]) * 1e3
    dest = None
    if exchange == 'host':
        if segment is None:
            raise ValueError("exchange='host' needs the shared HostSegment")
        dest = segment.dest(rank)
    elif exchange != 'rccl':
        raise ValueError(f'exchange {exchange!r}')
    pack = trace_blocks(engine, plan[rank], num, fields, image_pts, foc, flags, first_surf,
                        last_surf, dest=dest)
    counts_local = pack.counts()                    # synchronises this rank's launches
    tm = {'trace_ms': lap()}
    counts = exchange_counts(counts_local, plan, group, getattr(engine, 'device', 'cpu'))
    tm['counts_ms'] = lap()
    totals = [int(c.sum()) for c in counts]
    result = None
    if exchange == 'rccl':
        got = gather_packed(pack.xy, int(totals[rank]), totals, group)
        if backend == 'nccl' and got is not None:
            torch.cuda.current_stream(got.device).synchronize()
        tm['gather_ms'] = lap()
        if rank == 0:
            host = to_pinned_host(got)
            tm['d2h_ms'] = lap()
            result = spot_views(host, plan, counts)
            tm['reassembly_ms'] = lap()
    else:
        # the pairs are already in the shared segment; the count exchange above is also
        # the point after which every rank's stores are complete and visible
        tm['gather_ms'] = 0.0
        if rank == 0:
            tm['d2h_ms'] = 0.0
            result = spot_views(segment.array, plan, counts, segment.offsets)
            tm['reassembly_ms'] = lap()
    if timings is not None:
        timings.update(tm)
        timings['pairs_total'] = int(sum(totals))
        timings['pairs_per_rank'] = totals
    return result
