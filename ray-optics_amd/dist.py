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

Two ways to get the pairs into the consumer's host memory, both built, and both PIPELINED
(``trace_spot_sharded(..., pipeline=True)``, the default): a rank's row blocks are cut into
pieces of at most 4 Mi rays, each piece is one packed-hits launch into its own region of the
rank's HBM buffer (count on the device), and as soon as a piece's count is known -- while the
next pieces are being traced -- its pairs move on: over xGMI to rank 0 and from there by the
copy engine into host memory (``rccl``), or by the copy engine of the rank's own PCIe link
straight into the shared host segment (``host``).  The host buffer holds one region per
(field, wavelength) grid, so a piece's place depends only on the counts of the earlier pieces
of ITS grid; a rank traces the head of the grid it shares with the next rank first and the
tail of the grid it shares with the previous rank last, so those counts are known long before
they are needed.  End to end -> max(kernels, transfers) instead of their sum.  The
un-pipelined forms of round 3 (``pipeline=False``) are described next:

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


def _peer(group, r):
    """the GLOBAL rank of group-local rank r: P2POp names its peer by global rank, and a
    sub-group's members need not be global ranks 0..world-1"""
    if group is None:
        return r
    import torch.distributed as dist
    return dist.get_global_rank(group, r)


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
            for w in dist.batch_isend_irecv([dist.P2POp(dist.isend, mine.contiguous(), _peer(group, dst), group)]):
                w.wait()
        return None
    out = torch.empty((sum(totals), 2), dtype=torch.float64, device=mine.device)
    offs = np.concatenate([[0], np.cumsum(totals)])
    ops = []
    for k in range(world):
        if k == dst:
            out[offs[k]:offs[k + 1]].copy_(mine)
        elif totals[k]:
            ops.append(dist.P2POp(dist.irecv, out[offs[k]:offs[k + 1]], _peer(group, k), group))
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
                try:
                    # reserve the pages now: a full tmpfs then fails here with ENOSPC instead
                    # of with SIGBUS when a copy engine or a reader first touches them
                    os.posix_fallocate(f.fileno(), 0, self.nbytes)
                except OSError:
                    try:
                        os.unlink(self.path)
                    except OSError:
                        pass
                    raise
        self.rank = rank
        self._t = torch.from_file(self.path, shared=True, size=self.nbytes, dtype=torch.uint8)
        self._engine = engine
        self.dev_ptr = engine.pin_host_memory(self._t.data_ptr(), self.nbytes)
        self.array = np.frombuffer(memoryview(self._t.numpy()), dtype=np.float64).reshape(-1, 2)

    @classmethod
    def for_grids(cls, engine, name, n_grids, num, rank, create, dir=None):
        """the segment of the pipelined exchange: one region of num * num pairs per (field,
        wavelength) grid -- every rank copies its pieces to their final place in it"""
        return cls(engine, name, [num * num] * n_grids, rank, create, dir)

    @property
    def host_ptr(self):
        """host address of the mapping (destination of copy-engine transfers)"""
        return self._t.data_ptr()

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


# ---------------------------------------------------------------- the pipelined exchange
PIECE_RAYS = 1 << 22        # rays per packed-hits launch of the pipeline (a 2048 x 2048 grid)
TAPER_MIN_RAYS = 1 << 20    # the last piece of a rank is cut further when it has at least this many


@dataclass(frozen=True)
class Piece:
    """rows [row_begin, row_begin + row_count) of grid g = (fi, wi), traced by `rank` into
    its HBM buffer at ray offset `roff` (room for every ray of the piece)"""
    rank: int
    fi: int
    wi: int
    row_begin: int
    row_count: int
    roff: int
    g: int


def schedule(plan, num, n_wvls, max_rays=None, taper=True):
    """(pieces, order): pieces[rank] = the rank's blocks cut into Pieces of at most
    `max_rays` rays, in ray order; order[rank] = the indices of pieces[rank] in TRACE order --
    first the head of the grid the rank shares with the next rank (its count is what the next
    rank's tail waits for), last the tail of the grid it shares with the previous rank."""
    rows_max = max(1, (max_rays or PIECE_RAYS) // num)
    pieces = []
    for rank, blocks in enumerate(plan):
        lst, roff = [], 0
        for b in blocks:
            r, end = b.row_begin, b.row_begin + b.row_count
            while r < end:
                take = min(rows_max, end - r)
                lst.append(Piece(rank, b.fi, b.wi, r, take, roff, b.fi * n_wvls + b.wi))
                roff += take * num
                r += take
        pieces.append(lst)

    def trace_order(lst):
        idx = list(range(len(lst)))
        if not lst:
            return idx
        g_first, g_last = lst[0].g, lst[-1].g
        shared_prev = lst[0].row_begin > 0
        shared_next = lst[-1].row_begin + lst[-1].row_count < num
        head = [i for i in idx if lst[i].g == g_last] \
            if shared_next and not (shared_prev and g_first == g_last) else []
        tail = [i for i in idx if lst[i].g == g_first and i not in head] if shared_prev else []
        mid = [i for i in idx if i not in head and i not in tail]
        return head + mid + tail

    # the piece a rank traces LAST is the one nothing overlaps with: its pack pass, its copy and
    # (rccl) its send all sit behind the last kernel.  Cut it into a half and two quarters so
    # that only a quarter's worth of that work is exposed (two more stages: two more count
    # exchanges of ~0.06 ms against ~0.7 ms of a 48 MB copy)
    if taper:
        for rank, lst in enumerate(pieces):
            if not lst:
                continue
            k = trace_order(lst)[-1]
            p = lst[k]
            if p.row_count < 4 or p.row_count * num < TAPER_MIN_RAYS:
                continue
            half, quarter = p.row_count // 2, p.row_count // 4
            cuts = [half, quarter, p.row_count - half - quarter]
            sub, r, roff = [], p.row_begin, p.roff
            for c in cuts:
                sub.append(Piece(rank, p.fi, p.wi, r, c, roff, p.g))
                r += c
                roff += c * num
            pieces[rank] = lst[:k] + sub + lst[k + 1:]
    order = [trace_order(lst) for lst in pieces]
    return pieces, order


class _Placer:
    """where a piece's pairs go inside its grid's host region: behind the survivors of the
    earlier pieces of the same grid -- known once all their counts are"""

    def __init__(self, pieces):
        self.seq = {}
        for lst in pieces:                      # rank-major = row order within a grid
            for i, p in enumerate(lst):
                self.seq.setdefault(p.g, []).append((p.rank, i))
        self.count = {}

    def set(self, rank, idx, n):
        self.count[(rank, idx)] = int(n)

    def offset(self, g, rank, idx):
        off = 0
        for key in self.seq[g]:
            if key == (rank, idx):
                return off
            n = self.count.get(key)
            if n is None:
                return None
            off += n
        raise KeyError((g, rank, idx))

    def total(self, g):
        return sum(self.count[k] for k in self.seq[g])


class _NoEvent:
    def synchronize(self):
        pass


def _trace_spot_pipelined(engine, fields, image_pts, n_wvls, num, foc, flags, group, first_surf,
                          last_surf, by, exchange, segment, timings, lookahead=2, max_rays=None,
                          result_on='host'):
    import contextlib
    import ctypes as C
    import time
    import torch
    import torch.distributed as dist
    from .engine import make_opts, make_grid
    world, rank, backend = _group_info(group)
    plan = partition(len(fields), n_wvls, num, world, by)
    pieces, order = schedule(plan, num, n_wvls, max_rays)
    n_grids = len(fields) * n_wvls
    mine, my_order = pieces[rank], order[rank]
    n_mine = len(mine)
    S = max(len(o) for o in order)
    dev = getattr(engine, 'device', 'cpu')
    cuda = str(dev).startswith('cuda')
    nccl = backend == 'nccl'
    N = engine.table.n_ifcs
    if flags is None:
        flags = abi.INTERSECT_OBJ | abi.CHECK_APERTURES | abi.APPLY_VIGNETTING
    last = N - 2 if last_surf is None else last_surf
    t0 = time.perf_counter()

    if cuda:
        main = torch.cuda.current_stream(dev)
        side, copy = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
        on_side = lambda: torch.cuda.stream(side)       # noqa: E731

        def event(stream):
            e = torch.cuda.Event()
            e.record(stream)
            return e
    else:
        main = side = copy = None
        on_side = contextlib.nullcontext

        def event(stream):
            return _NoEvent()

    # this rank's pieces: one region each in xy, one count word per STAGE
    rays_mine = sum(p.row_count for p in mine) * num
    xy = torch.empty((max(rays_mine, 1), 2), dtype=torch.float64, device=dev if cuda else 'cpu')
    cnt = torch.zeros(max(S, 1), dtype=torch.int64, device=dev if cuda else 'cpu')
    allc = torch.zeros((max(S, 1), world), dtype=torch.int64)
    if cuda:
        allc = allc.pin_memory()
    allc_dev = torch.zeros((max(S, 1), world), dtype=torch.int64, device=dev) if (cuda and nccl) else None

    # where the pairs end up: one region per (field, wavelength) grid
    root = rank == 0
    lease = None
    if exchange == 'host':
        if segment is None or len(segment.caps) != n_grids or min(segment.caps) < num * num:
            raise ValueError("exchange='host' (pipelined) needs a HostSegment with one region of "
                             "num*num pairs per (field, wavelength) grid: HostSegment.for_grids(...)")
        host_ptr, grid_off, host_arr = segment.host_ptr, [int(o) for o in segment.offsets], segment.array
    elif exchange == 'rccl':
        grid_off = [g * num * num for g in range(n_grids + 1)]
        host_ptr = host_arr = None
        if root and result_on == 'device' and cuda:
            # the gathered spot diagram stays in rank 0's HBM (a consumer on the device): the
            # per-grid regions are device memory and the placement copies are device-to-device
            host_arr = torch.empty((max(grid_off[-1], 1), 2), dtype=torch.float64, device=dev)
            host_ptr = host_arr.data_ptr()
        elif root:
            if cuda:
                from .engine import _pool
                lease = _pool.take(torch, max(16 * grid_off[-1], 16))
                host_ptr = lease.ptr
                host_arr = lease.array((grid_off[-1], 2), np.float64)
            else:
                host_arr = np.empty((grid_off[-1], 2))
                host_ptr = host_arr.ctypes.data
    else:
        raise ValueError(f'exchange {exchange!r}')
    # rccl: rank 0 receives every other rank's pieces at the offsets they have over there
    stage = {}
    if exchange == 'rccl' and root and world > 1:
        for r in range(1, world):
            n_r = sum(p.row_count for p in pieces[r]) * num
            stage[r] = torch.empty((max(n_r, 1), 2), dtype=torch.float64,
                                   device=dev if (cuda and nccl) else 'cpu')

    placer = _Placer(pieces)
    evs = [None] * S
    launched = 0

    def launch_upto(k):
        nonlocal launched
        while launched < min(k, S):
            s = launched
            if s < n_mine:
                p = mine[my_order[s]]
                opts = make_opts(flags=flags, out_mode=abi.OUT_HITS_COMPACT, first_surf=first_surf,
                                 last_surf=last, foc=foc, image_pt=image_pts[p.fi])
                grid = make_grid((-1., -1.), (1., 1.), num, row_begin=p.row_begin, row_count=p.row_count)
                engine.trace_pupil_grid_hits_at(fields[p.fi], grid, p.wi, opts,
                                                xy.data_ptr() + 16 * p.roff, p.row_count * num,
                                                cnt.data_ptr() + 8 * s)
            evs[s] = event(main)
            launched += 1

    tm = {'stage_sync_ms': 0.0, 'gather_ms': 0.0}
    t_kernels_done = None
    pending = []
    overflow = None
    try:
        for s in range(S):
            launch_upto(s + 1 + lookahead)
            # ---- the counts of stage s, of every rank, on the host
            t1 = time.perf_counter()
            if cuda:
                side.wait_event(evs[s])
            with on_side():
                if world > 1:
                    if nccl:
                        dist.all_gather_into_tensor(allc_dev[s], cnt[s:s + 1], group=group)
                        allc[s].copy_(allc_dev[s], non_blocking=True)
                    else:
                        dist.all_gather_into_tensor(allc[s], cnt[s:s + 1].cpu(), group=group)
                else:
                    allc[s, 0:1].copy_(cnt[s:s + 1], non_blocking=True)
                ev_c = event(side)
            ev_c.synchronize()
            tm['stage_sync_ms'] += (time.perf_counter() - t1) * 1e3
            counts_s = [int(v) for v in allc[s].tolist()]
            for r in range(world):
                if s < len(order[r]):
                    if counts_s[r] < 0:
                        # every rank sees the same counts: all of them finish this stage's
                        # exchange (nothing moves for the overflowed piece) and raise together
                        # below, so no peer is left waiting in a collective
                        overflow = overflow or f'rank {r}: packed-hits overflow in piece {order[r][s]}'
                        counts_s[r] = 0
                    placer.set(r, order[r][s], counts_s[r])
            if s == n_mine - 1 or (n_mine == 0 and s == 0):
                t_kernels_done = time.perf_counter()
            # ---- rccl: the pairs of stage s go to rank 0 (grouped point-to-point)
            ev_x = None
            arrived = []
            if exchange == 'rccl' and world > 1:
                t1 = time.perf_counter()
                ops, keep = [], []
                if not root:
                    if s < n_mine and counts_s[rank] > 0:
                        p = mine[my_order[s]]
                        with on_side():
                            buf = xy[p.roff:p.roff + counts_s[rank]]
                            buf = buf if nccl else buf.cpu()
                        keep.append(buf)
                        ops.append(dist.P2POp(dist.isend, buf, _peer(group, 0), group))
                else:
                    for r in range(1, world):
                        if s < len(order[r]):
                            q = pieces[r][order[r][s]]
                            arrived.append((r, order[r][s]))
                            if counts_s[r] > 0:
                                ops.append(dist.P2POp(dist.irecv, stage[r][q.roff:q.roff + counts_s[r]], _peer(group, r), group))
                if ops:
                    with on_side():
                        for w in dist.batch_isend_irecv(ops):
                            w.wait()
                        ev_x = event(side)
                tm['gather_ms'] += (time.perf_counter() - t1) * 1e3
            if overflow:
                raise RuntimeError(overflow)
            # ---- host placement: copy engine, behind the events of this stage
            if exchange == 'host' or root:
                cand = pending + ([(rank, my_order[s])] if s < n_mine else []) + arrived
                pending = []
                if cuda:
                    copy.wait_event(evs[s])
                    if ev_x is not None:
                        copy.wait_event(ev_x)
                for r, i in cand:
                    q = pieces[r][i]
                    off = placer.offset(q.g, r, i)
                    if off is None:
                        pending.append((r, i))
                        continue
                    n = placer.count[(r, i)]
                    if n:
                        src = (xy.data_ptr() if r == rank else stage[r].data_ptr()) + 16 * q.roff
                        engine.copy_async(host_ptr + 16 * (grid_off[q.g] + off), src, 16 * n, copy)
    except BaseException:
        # an exception out of the stage loop (overflow, a launch or collective error) must not
        # hand xy / stage[] / the pinned lease back while copies or sends still use them
        if cuda:
            try:
                copy.synchronize()
                side.synchronize()
            except Exception:       # noqa: BLE001  (the original error is the one to report)
                pass
        raise
    if t_kernels_done is None:
        t_kernels_done = time.perf_counter()
    tm['trace_ms'] = (t_kernels_done - t0) * 1e3
    tm['counts_ms'] = tm['stage_sync_ms']
    # pieces whose place depended on a later stage of another rank (a grid held by three ranks)
    for r, i in pending:
        q = pieces[r][i]
        off = placer.offset(q.g, r, i)
        assert off is not None, 'every count is known after the last stage'
        n = placer.count[(r, i)]
        if n:
            src = (xy.data_ptr() if r == rank else stage[r].data_ptr()) + 16 * q.roff
            engine.copy_async(host_ptr + 16 * (grid_off[q.g] + off), src, 16 * n, copy)
    if cuda:
        copy.synchronize()
        side.synchronize()
    if exchange == 'host' and world > 1:
        # every rank's copies have landed in the shared segment before rank 0 reads it
        tok = torch.zeros(1, device=dev if nccl else 'cpu')
        dist.all_reduce(tok, group=group)
        if nccl:
            torch.cuda.current_stream(dev).synchronize()
    t2 = time.perf_counter()
    tm['d2h_ms'] = (t2 - t_kernels_done) * 1e3
    result = None
    totals = [sum(placer.count[(r, i)] for i in range(len(pieces[r]))) for r in range(world)]
    if root:
        result = {}
        for g in range(n_grids):
            n = placer.total(g)
            result[(g // n_wvls, g % n_wvls)] = host_arr[grid_off[g]:grid_off[g] + n]
        tm['reassembly_ms'] = (time.perf_counter() - t2) * 1e3
    if timings is not None:
        timings.update(tm)
        timings['pairs_total'] = int(sum(totals))
        timings['pairs_per_rank'] = totals
        timings['pieces'] = [len(p) for p in pieces]
        timings['stages'] = S
        timings['pipelined'] = True
    del lease
    return result


# ---------------------------------------------------------------- the sharded spot diagram
def trace_spot_sharded(engine, fields, image_pts, n_wvls, num, foc, flags=None, group=None,
                       first_surf=1, last_surf=None, by='rows', exchange='rccl', segment=None,
                       timings=None, pipeline=True, max_piece_rays=None, result_on='host'):
    """Spot diagrams for every (field, wavelength), sharded over the process
    group: each rank traces its row blocks on its own GPU (packed hits), the
    pairs reach rank 0's host memory by the chosen exchange.  Returns on rank 0
    {(fi, wi): (R_ok, 2) float64 array in the reference's i-outer/j-inner ray
    order}; None on the other ranks.  ``timings`` (a dict) receives the phases in
    milliseconds: trace (launch -> counts on the host), counts, gather, d2h,
    reassembly.  ``pipeline=True`` (default): pieces of at most ``max_piece_rays`` rays move
    on while later pieces are traced (module docstring); ``exchange='host'`` then wants a
    ``HostSegment.for_grids`` segment (one region per grid).  ``pipeline=False``: round 3's
    trace-everything-then-exchange form (per-rank slices of the segment).
    ``result_on='device'`` (pipelined ``rccl`` only): the arrays are torch tensors in rank 0's
    HBM -- the RCCL gather alone, for a consumer on the device."""
    import time
    import torch
    if pipeline:
        return _trace_spot_pipelined(engine, fields, image_pts, n_wvls, num, foc, flags, group,
                                     first_surf, last_surf, by, exchange, segment, timings,
                                     max_rays=max_piece_rays, result_on=result_on)
    if result_on != 'host':
        raise ValueError("result_on='device' needs the pipelined exchange")
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


# ---------------------------------------------------------------- FULL packets stay where they are
class ShardedPackets:
    """The FULL packets of one (field, wavelength) pupil grid cut by pupil rows over the ranks
    (SURVEY section 8e): rank k keeps ``local`` -- seg [n_seg, 10, R_k], op, status, fail_surf
    of its rows [row_begin, row_begin + row_count) -- in its own HBM.  Packets are never
    exchanged wholesale (BASELINE configs[1] is 1.1 GB per grid, configs[4] would be 636 GB);
    a consumer that needs some of them asks for those rays with :meth:`fetch`."""

    def __init__(self, local, plan, num, n_seg, group):
        self.local, self.plan, self.num, self.n_seg, self.group = local, plan, int(num), int(n_seg), group
        self.world, self.rank, self.backend = _group_info(group)
        mine = plan[self.rank]
        self.row_begin = mine[0].row_begin if mine else 0
        self.row_count = sum(b.row_count for b in mine)

    def owner_of(self, rays):
        """rank that holds each global ray index r = i * num + j (i: pupil row)"""
        rows = np.asarray(rays, dtype=np.int64) // self.num
        ends = np.array([(self.num * (k + 1)) // self.world for k in range(self.world)], dtype=np.int64)
        return np.searchsorted(ends, rows, side='right').astype(np.int64)     # partition()'s bounds

    def fetch(self, rays, dst=0):
        """COLLECTIVE (every rank calls it with the same ``rays``): the packets of the global ray
        indices ``rays`` travel from the ranks that hold them to rank ``dst`` -- grouped
        point-to-point, one message per owner; who owns what follows from the plan, so no
        sizes are exchanged.  Returns on ``dst`` a dict seg [n_seg, 10, n] / op [n] / status [n] /
        fail_surf [n] (NumPy, in the order of ``rays``), None elsewhere."""
        import torch
        import torch.distributed as dist
        rays = np.asarray(rays, dtype=np.int64).ravel()
        if rays.size and (rays.min() < 0 or rays.max() >= self.num * self.num):
            raise IndexError('ray index outside the grid')
        owner = self.owner_of(rays)
        width = self.n_seg * abi.SEG_DOUBLES + 3        # + op, status, fail_surf (as doubles)

        def rows_of(k):
            return np.nonzero(owner == k)[0]

        def pack_mine():
            sel = rows_of(self.rank)
            if not len(sel):
                return sel, torch.empty((0, width), dtype=torch.float64)
            loc = torch.as_tensor(rays[sel] - self.row_begin * self.num, device=self.local.seg.device)
            seg = self.local.seg[:self.n_seg].index_select(2, loc)          # [n_seg, 10, n]
            out = torch.empty((len(sel), width), dtype=torch.float64, device=seg.device)
            out[:, :width - 3] = seg.permute(2, 0, 1).reshape(len(sel), width - 3)
            out[:, width - 3] = self.local.op.index_select(0, loc)
            out[:, width - 2] = self.local.status.index_select(0, loc).to(torch.float64)
            out[:, width - 1] = self.local.fail_surf.index_select(0, loc).to(torch.float64)
            return sel, out

        sel_mine, mine = pack_mine()
        if self.world > 1:
            if self.rank != dst:
                if len(sel_mine):
                    buf = _wire(mine, self.backend).contiguous()
                    for w in dist.batch_isend_irecv([dist.P2POp(dist.isend, buf, _peer(self.group, dst), self.group)]):
                        w.wait()
                return None
            wire_dev = mine.device if self.backend == 'nccl' else 'cpu'
            got = {self.rank: _wire(mine, self.backend)}
            ops = []
            for k in range(self.world):
                n_k = len(rows_of(k))
                if k != dst and n_k:
                    got[k] = torch.empty((n_k, width), dtype=torch.float64, device=wire_dev)
                    ops.append(dist.P2POp(dist.irecv, got[k], _peer(self.group, k), self.group))
            if ops:
                for w in dist.batch_isend_irecv(ops):
                    w.wait()
        else:
            got = {0: mine}
        table = np.empty((len(rays), width))
        for k, t in got.items():
            table[rows_of(k)] = t.cpu().numpy()
        return {'seg': np.ascontiguousarray(table[:, :width - 3].reshape(len(rays), self.n_seg, abi.SEG_DOUBLES)
                                            .transpose(1, 2, 0)),
                'op': table[:, width - 3].copy(),
                'status': table[:, width - 2].astype(np.uint8),
                'fail_surf': table[:, width - 1].astype(np.int16)}


def trace_packets_sharded(engine, fld, wvl_idx, num, opts, group=None):
    """One (field, wavelength) pupil grid with FULL packets, the pupil rows cut over the ranks;
    every rank traces its rows into its own HBM and nothing is exchanged.  ``opts`` as for
    ``engine.trace_pupil_grid`` with ``out_mode`` ROX_OUT_FULL.  Returns :class:`ShardedPackets`."""
    from .engine import make_grid
    if opts.out_mode != abi.OUT_FULL:
        raise ValueError('trace_packets_sharded keeps FULL packets: opts.out_mode must be OUT_FULL')
    world, rank, _backend = _group_info(group)
    plan = partition(1, 1, num, world, 'rows')
    mine = plan[rank]
    row_begin = mine[0].row_begin if mine else 0
    row_count = sum(b.row_count for b in mine)
    local = None
    if row_count:
        grid = make_grid((-1., -1.), (1., 1.), num, row_begin=row_begin, row_count=row_count)
        local = engine.trace_pupil_grid(fld, grid, wvl_idx, opts)
    return ShardedPackets(local, plan, num, engine.num_segments(opts.flags), group)
