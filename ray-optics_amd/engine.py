"""Launch wrappers over the C ABI of libroxtrace.so (include/roxtrace.h).

PyTorch is used only as plumbing: device buffers and the current HIP stream.
There is no CPU fallback -- a missing library, or a missing GPU, raises.
"""
import collections
import ctypes as C
import functools
import os
import threading

import numpy as np

from . import abi
from .table import SurfaceTable

_HERE = os.path.dirname(os.path.abspath(__file__))
# ROX_LIB selects an experiment build of the same HIP source (tools/ab_bench.py)
LIB_PATH = os.environ.get('ROX_LIB') or os.path.join(_HERE, 'libroxtrace.so')
_lib = None


class EngineError(RuntimeError):
    pass


def load_library():
    """dlopen libroxtrace.so; never falls back to anything else."""
    global _lib
    if _lib is None:
        # torch ships its own libamdhip64; it must be the HIP runtime of the
        # process *before* libroxtrace.so (linked against the same SONAME)
        # is dlopen'ed, or the two sides would not share device pointers
        import torch  # noqa: F401
        if not os.environ.get('ROX_LIB'):
            # missing (fresh checkout) or stale (a source, header or flag changed
            # since it was built -- build.py compares a digest of its inputs):
            # compile the HIP sources in-tree now.  A stale library is never loaded.
            try:
                import importlib.util
                spec = importlib.util.spec_from_file_location(
                    'rox_build', os.path.join(_HERE, 'build.py'))
                b = importlib.util.module_from_spec(spec)
                spec.loader.exec_module(b)
                b.build()
            except Exception as e:
                raise EngineError(
                    f'{LIB_PATH} is missing or stale and could not be built ({e!r}): run '
                    '`python ray-optics_amd/build.py` (hipcc --offload-arch=gfx950).  '
                    'There is no CPU fallback.')
        if not os.path.exists(LIB_PATH):
            raise EngineError(f'{LIB_PATH} is missing.  There is no CPU fallback.')
        lib = abi.declare(C.CDLL(LIB_PATH))
        if lib.rox_abi_version() != abi.ABI_VERSION:
            raise EngineError('libroxtrace.so ABI version mismatch')
        _lib = lib
    return _lib


def _check(rc, what):
    if rc != 0:
        msg = load_library().rox_last_error().decode(errors='replace')
        raise EngineError(f'{what} failed ({rc}): {msg}')


def make_opts(flags=abi.INTERSECT_OBJ, out_mode=abi.OUT_FULL, first_surf=0,
              last_surf=-1, eps=1.0e-12, fuzz=1e-5, foc=0.0, image_pt=(0., 0.), wf=None):
    o = abi.Opts()
    if wf is not None:
        o.wf = wf
    o.flags, o.out_mode = int(flags), int(out_mode)
    o.first_surf, o.last_surf = int(first_surf), int(last_surf)
    o.eps, o.fuzz, o.foc = float(eps), float(fuzz), float(foc)
    o.image_pt[0], o.image_pt[1] = float(image_pt[0]), float(image_pt[1])
    return o


def make_grid(start, stop, num, kind=abi.GRID_PRODUCT, row_begin=0, row_count=0):
    g = abi.Grid()
    g.start[0], g.start[1] = float(start[0]), float(start[1])
    g.stop[0], g.stop[1] = float(stop[0]), float(stop[1])
    g.num, g.kind = int(num), int(kind)
    g.row_begin, g.row_count = int(row_begin), int(row_count)
    return g


def grid_rays(grid):
    """number of rays a rox_grid describes"""
    if grid.kind == abi.GRID_FAN:
        return grid.num
    return (grid.row_count or grid.num) * grid.num


def calc_psf(opd, ndim, maxdim):
    """analyses.calc_psf (rayoptics/raytr/analyses.py:848-875) on the device:
    ``opd`` is the [ndim, ndim] OPD grid in waves (NumPy, NaN = no data, or a
    float64 torch tensor already in HBM); returns the normalised [maxdim, maxdim]
    PSF as the same kind of array.  A pruned DFT as two complex fp64 GEMMs on
    the matrix cores (csrc/psf.hip)."""
    import torch
    if not torch.cuda.is_available():
        raise EngineError('no GPU visible: the PSF kernels have no CPU fallback')
    lib = load_library()
    ndim, maxdim = int(ndim), int(maxdim)
    if isinstance(opd, torch.Tensor):
        w = opd.to(dtype=torch.float64).contiguous()
        if tuple(w.shape) != (ndim, ndim):
            raise ValueError(f'opd is {tuple(w.shape)}, expected ({ndim}, {ndim})')
        out = torch.empty((maxdim, maxdim), dtype=torch.float64, device=w.device)
        with torch.cuda.device(w.device):
            _check(lib.rox_calc_psf(w.data_ptr(), ndim, maxdim, out.data_ptr(), 0,
                                    C.c_void_p(torch.cuda.current_stream(w.device).cuda_stream)),
                   'rox_calc_psf')
        out._keep = w
        return out
    w = np.ascontiguousarray(opd, dtype=np.float64)
    if w.shape != (ndim, ndim):
        raise ValueError(f'opd is {w.shape}, expected ({ndim}, {ndim})')
    out = np.empty((maxdim, maxdim))
    _check(lib.rox_calc_psf(w.ctypes.data, ndim, maxdim, out.ctypes.data, abi.HOST_POINTERS, None),
           'rox_calc_psf')
    return out


_NP_DTYPES = {}


def _np_dtypes(torch):
    if not _NP_DTYPES:
        _NP_DTYPES.update({torch.float64: np.float64, torch.uint8: np.uint8,
                           torch.int16: np.int16, torch.int32: np.int32,
                           torch.int64: np.int64})


def padded_ld(R):
    """row pitch (in doubles) of the SoA packet buffer.  Rows exactly 2^k bytes
    apart put one ray's 130 packet components on the same HBM channel; a pitch
    of 2 KiB mod 4 KiB spreads them (tools/store_ld.hip: 5.9 vs 5.0 TB/s for the
    same store stream)."""
    if R <= 0:
        return 1
    return (R + 511) // 512 * 512 + 256


def _default_pool_budget():
    """ROX_PINNED_POOL_MB, else an eighth of the host's RAM, at most 4 GiB: page-locked memory
    kept by the pool is taken from what HostSegment registration, RCCL and other pinned
    allocations can get"""
    env = os.environ.get('ROX_PINNED_POOL_MB')
    if env:
        return int(env) << 20
    try:
        ram = os.sysconf('SC_PAGE_SIZE') * os.sysconf('SC_PHYS_PAGES')
    except (ValueError, OSError, AttributeError):
        ram = 32 << 30
    return int(min(4 << 30, max(256 << 20, ram // 8)))


class PinnedPool:
    """page-locked host buffers that outlive one call: handing the caller a
    NumPy array that *is* the pinned buffer saves a second pass over the data
    (a 13 MB spot diagram is ~1.5 ms of memcpy, five times the trace).  A block
    goes back to the pool when the last array viewing it is garbage-collected.

    Thread safety: ``take`` and ``trim`` hold the pool's lock.  A block comes back from
    ``_Lease.__del__`` -- any thread, any time, possibly inside a garbage collection that a
    thread holding the lock triggered -- so ``give_back`` takes no lock: it appends to a deque
    (atomic under the GIL) that the next ``take`` / ``trim`` shelves under the lock."""

    def __init__(self):
        self._lock = threading.Lock()
        self._returned = collections.deque()    # (nbytes, tensor) not yet shelved
        self._free = {}         # nbytes (rounded) -> [uint8 pinned tensors]
        self._shelved = 0       # bytes sitting in _free
        # Blocks are kept up to a byte budget, not a count: a SpotDiagramFigure holds nine arrays
        # of one size at a time, and releasing / re-pinning the surplus cost 3 ms per block
        # (hipHostFree + hipHostMalloc) -- 95 ms per refresh at num_rays = 256 with a bound of
        # four blocks per size (tools/spot_figure_profile.py).
        self.budget = _default_pool_budget()

    @property
    def _held(self):
        """bytes the pool keeps pinned right now (shelved + returned and not yet shelved)"""
        with self._lock:
            self._drain()
            return self._shelved

    @staticmethod
    def _round(nbytes):
        if nbytes > (64 << 20):                 # big blocks: 16 MiB granules
            return (nbytes + (16 << 20) - 1) // (16 << 20) * (16 << 20)
        n = 4096
        while n < nbytes:
            n *= 2
        return n

    def take(self, torch, nbytes):
        n = self._round(max(int(nbytes), 1))
        with self._lock:
            self._drain()
            lst = self._free.get(n)
            t = None
            if lst:
                t = lst.pop()
                self._shelved -= n
        if t is None:
            t = torch.empty(n, dtype=torch.uint8).pin_memory()
        return _Lease(self, n, t)

    def give_back(self, n, t):
        # (from _Lease.__del__: no lock, see the class docstring)
        self._returned.append((n, t))

    def _drain(self):
        while True:
            try:
                n, t = self._returned.popleft()
            except IndexError:
                return
            self._shelve(n, t)

    def _shelve(self, n, t):
        # bounded by max(budget, one block): what does not fit is unpinned.  Room is made by
        # dropping blocks of OTHER sizes first (largest first) -- a pool that has swept many sizes
        # keeps what is in use now -- and a single block larger than the whole budget is still
        # kept while nothing else is (re-pinning gigabytes per call costs a second)
        if self._shelved + n > self.budget:
            for m in sorted((m for m, lst in self._free.items() if m != n and lst), reverse=True):
                while self._free[m] and self._shelved + n > self.budget:
                    self._free[m].pop()
                    self._shelved -= m
                if self._shelved + n <= self.budget:
                    break
        if self._shelved + n <= self.budget or self._shelved == 0:
            self._free.setdefault(n, []).append(t)
            self._shelved += n

    def trim(self):
        """un-pin everything the pool keeps (blocks on loan are not touched); returns the
        bytes released.  ``session.clear()`` calls it."""
        with self._lock:
            self._drain()
            freed = self._shelved
            self._free.clear()
            self._shelved = 0
        return freed


class _Lease:
    """one pinned block on loan; NumPy arrays made by :meth:`array` keep it
    alive (they hold it as their base object)"""

    def __init__(self, pool, n, tensor):
        self._pool, self._n, self.tensor = pool, n, tensor
        self.ptr = tensor.data_ptr()

    def array(self, shape, dtype, offset=0):
        a = np.asarray(_View(self, shape, np.dtype(dtype).str, self.ptr + offset))
        return a

    def __del__(self):
        try:
            self._pool.give_back(self._n, self.tensor)
        except Exception:
            pass


class _View:
    def __init__(self, lease, shape, typestr, ptr):
        self._lease = lease
        self.__array_interface__ = {'shape': tuple(shape), 'typestr': typestr,
                                    'data': (ptr, False), 'version': 3}


_pool = PinnedPool()

# pupil launches whose outputs are at most this many bytes go through ROX_HOST_POINTERS
# (TraceEngine.trace_pupil_np); the library bounces up to 4 MiB through its mapped block
HOST_DIRECT_BYTES = 1 << 20


class DeviceResult:
    """SoA trace results resident in HBM (torch tensors on the engine's device).

    seg: FULL [n_seg, 10, R] | LAST [10, R] | HITS [2, R];  op [R] f64;
    status [R] u8;  fail_surf [R] i16;  pupil [2, R] f64 or None.
    Slots the trace does not write (segments past a failure, LAST/HITS rows of
    failed rays) keep the buffer's prior contents: NaN when ``nan_fill``.
    """

    def __init__(self, torch, device, n_seg, R, out_mode, want_pupil, nan_fill, ld=None):
        self.ld = padded_ld(R) if ld is None else max(int(ld), 1)
        if out_mode == abi.OUT_FULL:
            shape = (n_seg, abi.SEG_DOUBLES, self.ld)
        elif out_mode == abi.OUT_LAST:
            shape = (abi.SEG_DOUBLES, self.ld)
        elif out_mode == abi.OUT_OPD:
            shape = (1, self.ld)
        elif out_mode == abi.OUT_FAN:
            shape = (3, self.ld)
        else:
            shape = (2, self.ld)
        new = (lambda s: torch.full(s, float('nan'), dtype=torch.float64, device=device)) \
            if nan_fill else (lambda s: torch.empty(s, dtype=torch.float64, device=device))
        self.R = R
        self.out_mode = out_mode
        self._torch = torch
        self._seg = new(shape)                  # pitched storage
        self.seg = self._seg[..., :R]           # [.., R] view the callers index
        self.op = new((R,))
        self.status = torch.empty((R,), dtype=torch.uint8, device=device)
        self.fail_surf = torch.empty((R,), dtype=torch.int16, device=device)
        self._pupil = new((2, self.ld)) if want_pupil else None
        self.pupil = self._pupil[:, :R] if want_pupil else None

    def out_struct(self):
        o = abi.Out()
        o.seg = self._seg.data_ptr()
        o.op = self.op.data_ptr()
        o.status = self.status.data_ptr()
        o.fail_surf = self.fail_surf.data_ptr()
        o.pupil = self._pupil.data_ptr() if self._pupil is not None else None
        o.ld = self.ld
        return o

    def to_host(self, want=('seg', 'op', 'status', 'fail_surf', 'pupil')):
        """NumPy copies of the arrays named in ``want`` (others None), through
        pooled pinned memory, one synchronisation"""
        class _H:
            pass
        h = _H()
        h.R, h.out_mode = self.R, self.out_mode
        torch = self._torch
        pend = []
        for name in ('seg', 'op', 'status', 'fail_surf', 'pupil'):
            t = getattr(self, name) if name in want else None
            if t is None:
                setattr(h, name, None)
                continue
            lease = _pool.take(torch, t.numel() * t.element_size())
            dst = lease.tensor[:t.numel() * t.element_size()].view(t.dtype).view(t.shape)
            dst.copy_(t, non_blocking=True)
            pend.append((name, lease, t))
        torch.cuda.current_stream(self.seg.device).synchronize()
        for name, lease, t in pend:
            setattr(h, name, lease.array(t.shape, _NP_DTYPES[t.dtype]))
        return h


class HitsPack:
    """A buffer that ROX_OUT_HITS_COMPACT | ROX_HITS_APPEND launches pack their
    surviving (x, y) pairs into, one launch behind the other, with the running
    count on the device: ``xy`` [cap, 2] f64 in HBM -- or ``dest`` = (pointer,
    capacity) of other device-visible memory, e.g. a slice of a pinned shared host
    segment -- ``count`` [1] i64, and the cumulative count after each launch."""

    def __init__(self, torch, device, cap, max_launches, dest=None):
        self._torch, self.device = torch, device
        if dest is None:
            self.xy = torch.empty((int(cap), 2), dtype=torch.float64, device=device)
            self.seg_ptr, self.cap = self.xy.data_ptr(), int(cap)
        else:
            self.xy = None
            self.seg_ptr, self.cap = int(dest[0]), int(dest[1])
        self.count = torch.zeros(1, dtype=torch.int64, device=device)
        self.cum = torch.zeros(max(int(max_launches), 1), dtype=torch.int64, device=device)
        self.n_launches = 0
        self.rays = 0

    def counts(self):
        """survivors per launch (synchronises the launch stream).  The kernels never store
        at or beyond the capacity handed to them (rox_out.ld); a launch that needed more
        leaves a negative running count, reported here"""
        c = self.cum[:self.n_launches].cpu().numpy()
        if (c < 0).any():
            k = int(np.argmax(c < 0))
            raise EngineError(f'HitsPack overflow: launch {k} needed {-int(c[k])} pairs of room, '
                              f'the destination holds {self.cap}')
        return np.diff(np.concatenate([[0], c])).astype(np.int64)


def _in_flight(fn):
    """a TraceEngine entry that uses the device handle (see TraceEngine.__init__)"""
    @functools.wraps(fn)
    def guarded(self, *args, **kwargs):
        self._enter()
        try:
            return fn(self, *args, **kwargs)
        finally:
            self._leave()
    return guarded


class ModelMemo:
    """What the drop-in layer remembers per MODEL STATE -- i.e. per engine: a model edit makes a
    new engine (session.engine_for), so nothing here outlives the state it was computed from.

    ``chief_rays``  {(bytes(rox_field), wavelength index): (seg [N, 10], op) | None}: the
                    one-launch chief-ray batch of trace.trace_chief_ray
    ``obj_coords``  the memo of ``osp.obj_coords(fld)`` (table._obj_coords: the reverse
                    chief-ray iteration of real-image-height fields, once per launch
                    instead of once per ray)
    ``scratch``     device-resident launch outputs that product calls reduce on the device
                    (trace.trace_grid_spot_stats)

    session.engine_for hands one engine to every thread that traces the same model: writers
    hold ``lock``."""
    __slots__ = ('lock', 'chief_rays', 'obj_coords', 'scratch')

    def __init__(self):
        self.lock = threading.Lock()
        self.chief_rays = {}
        self.obj_coords = {}
        self.scratch = {}       # device buffers the product calls reuse ({(thread, rays): DeviceResult})

    def clear(self):
        with self.lock:
            self.chief_rays.clear()
            self.obj_coords.clear()
            self.scratch.clear()


class TraceEngine:
    """one immutable surface table on one GPU.

    A model edit means a new engine, as ``path_sequence.cache_clear()`` does in
    the reference (rayoptics/seq/sequential.py:666-668)."""

    def __init__(self, table: SurfaceTable, device=None):
        import torch
        if not torch.cuda.is_available():
            raise EngineError('no GPU visible: the trace engine has no CPU fallback')
        self.torch = torch
        _np_dtypes(torch)
        self.lib = load_library()
        self.device = torch.device('cuda', torch.cuda.current_device() if device is None
                                   else torch.device(device).index or 0)
        self.table = table
        self.memo = ModelMemo()         # cleared by close()
        self._nseg = {}
        self._handle = C.c_void_p()
        # lifetime of the device handle under threads (session.engine_for hands one engine to
        # every thread that traces the same model, and evicts by LRU): calls in flight are
        # counted; close() with calls in flight is carried out by the last of them; a call on
        # a closed engine re-creates the handle from the table it was built from.  The handle
        # is therefore never destroyed under a running rox_* call (ctypes drops the GIL).
        self._life = threading.Lock()
        self._calls = 0
        self._close_pending = False
        with self._life:
            self._open()

    def _open(self):
        with self.torch.cuda.device(self.device):
            _check(self.lib.rox_set_device(self.device.index), 'rox_set_device')
            _check(self.lib.rox_system_create(self.table.rows, self.table.n_ifcs,
                                              self.table.n_table.ctypes.data,
                                              self.table.wvls_arr.ctypes.data,
                                              len(self.table.wvls), C.byref(self._handle)),
                   'rox_system_create')

    def _destroy(self):
        if self._handle:
            self.lib.rox_system_destroy(self._handle)
            self._handle = C.c_void_p()
        self._close_pending = False

    def _enter(self):
        with self._life:
            if not self._handle:
                self._open()
            self._calls += 1

    def _leave(self):
        with self._life:
            self._calls -= 1
            if self._calls == 0 and self._close_pending:
                self._destroy()

    @property
    def is_open(self):
        return bool(self._handle)

    def close(self):
        """destroy the device handle -- at once, or, with calls in flight on other threads,
        when the last of them returns.  A later call re-creates it."""
        self.memo.clear()
        with self._life:
            if self._calls > 0:
                self._close_pending = True
            else:
                self._destroy()

    def __del__(self):
        try:
            self._destroy()
        except Exception:
            pass

    # -- helpers ------------------------------------------------------------
    @_in_flight
    def num_segments(self, flags=0):
        n = C.c_int32()
        _check(self.lib.rox_system_num_segments(self._handle, int(flags), C.byref(n)),
               'rox_system_num_segments')
        return n.value

    def _stream(self):
        return C.c_void_p(self.torch.cuda.current_stream(self.device).cuda_stream)

    def _dev(self, x, dtype):
        t = self.torch
        if isinstance(x, t.Tensor):
            return x.to(device=self.device, dtype=dtype).contiguous()
        return t.from_numpy(np.ascontiguousarray(x)).to(device=self.device, dtype=dtype)

    def _result(self, R, opts, want_pupil, nan_fill, out):
        if out is not None:
            return out
        return DeviceResult(self.torch, self.device, self.num_segments(opts.flags), R,
                            opts.out_mode, want_pupil, nan_fill)

    # -- entries --------------------------------------------------------------
    @_in_flight
    def trace_rays(self, pt0, dir0, wvl_idx=0, opts=None, nan_fill=False, out=None):
        """pt0, dir0: [3, R] (numpy or torch); wvl_idx: int or int32[R]"""
        t = self.torch
        opts = opts or make_opts()
        pt0 = self._dev(pt0, t.float64)
        dir0 = self._dev(dir0, t.float64)
        R = pt0.shape[1]
        if np.ndim(wvl_idx) == 0 and not isinstance(wvl_idx, t.Tensor):
            wi, wi_ptr, wi_all = None, None, int(wvl_idx)
        else:
            wi = self._dev(wvl_idx, t.int32)
            wi_ptr, wi_all = wi.data_ptr(), 0
        res = self._result(R, opts, False, nan_fill, out)
        o = res.out_struct()
        with t.cuda.device(self.device):
            _check(self.lib.rox_trace_rays(self._handle, R, pt0.data_ptr(), dir0.data_ptr(),
                                           wi_ptr, wi_all, C.byref(opts), C.byref(o),
                                           self._stream()), 'rox_trace_rays')
        res._keep = (pt0, dir0, wi)     # inputs must outlive the async launch
        return res

    @_in_flight
    def trace_pupil_grid(self, fld, grid, wvl_idx=0, opts=None, want_pupil=True,
                         nan_fill=False, out=None):
        opts = opts or make_opts()
        R = grid_rays(grid)
        res = self._result(R, opts, want_pupil, nan_fill, out)
        o = res.out_struct()
        with self.torch.cuda.device(self.device):
            _check(self.lib.rox_trace_pupil_grid(self._handle, C.byref(fld), C.byref(grid),
                                                 int(wvl_idx), C.byref(opts), C.byref(o),
                                                 self._stream()), 'rox_trace_pupil_grid')
        return res

    def _batch_args(self, flds, wvl_idxs, opts_list, outs):
        n = len(flds)
        if not (len(wvl_idxs) == len(opts_list) == len(outs) == n):
            raise EngineError('trace_pupil_grids: flds, wvl_idxs, opts and outs differ in length')
        f_arr = (abi.Field * n)(*flds)
        w_arr = (C.c_int32 * n)(*[int(w) for w in wvl_idxs])
        o_arr = (abi.Opts * n)(*opts_list)
        out_arr = (abi.Out * n)(*outs)
        return n, f_arr, w_arr, o_arr, out_arr

    @_in_flight
    def trace_pupil_grids(self, flds, wvl_idxs, grid, opts_list, want_pupil=True,
                          nan_fill=False, outs=None):
        """``grid`` traced for every (flds[i], wvl_idxs[i], opts_list[i]) in ONE launch
        (rox_trace_pupil_grids): the (field x wavelength) loops of a spot diagram, a set of
        ray fans or a wavefront map.  Returns one DeviceResult per item; each is what
        trace_pupil_grid would have returned for it."""
        R = grid_rays(grid)
        n = len(flds)
        res = [self._result(R, opts_list[i], want_pupil, nan_fill, outs[i] if outs else None)
               for i in range(n)]
        n, f_arr, w_arr, o_arr, out_arr = self._batch_args(flds, wvl_idxs, opts_list,
                                                           [r.out_struct() for r in res])
        with self.torch.cuda.device(self.device):
            _check(self.lib.rox_trace_pupil_grids(self._handle, n, f_arr, w_arr, C.byref(grid),
                                                  o_arr, out_arr, self._stream()),
                   'rox_trace_pupil_grids')
        return res

    @_in_flight
    def trace_pupil_grids_host(self, flds, wvl_idxs, grid, opts_list):
        """FULL / LAST packets (or the FAN / OPD / HITS rows) of several small pupil grids by ONE
        launch whose stores go straight into one pinned host block -- no copy-engine transfer,
        one synchronise -- for figure-sized batches (the chief rays of every field and
        wavelength of a model, trace.trace_chief_ray; the fans of a RayFanFigure).  Returns one host result per item (``seg`` [n_seg, 10, R],
        ``op``, ``status``, ``fail_surf``, ``pupil``: NumPy views of the block; segments past
        a failure are NaN)."""
        R = grid_rays(grid)
        n = len(flds)
        mode = opts_list[0].out_mode
        # every item's slice of the pinned block is sized from the first item's output mode and
        # segment count: the library takes per-item options, so a list that mixes them would
        # have the kernel write past a slice (the library checks the same for a batched launch,
        # but falls back to per-item launches for a single item)
        ph = opts_list[0].flags & abi.FILTER_PHANTOMS
        for o in opts_list:
            if o.out_mode != mode or (o.flags & abi.FILTER_PHANTOMS) != ph:
                raise EngineError('trace_pupil_grids_host: out_mode and FILTER_PHANTOMS must be the '
                                  'same for every item')
        if len(opts_list) != n or len(wvl_idxs) != n:
            raise EngineError('trace_pupil_grids_host: one wavelength index and one rox_opts per field')
        if mode == abi.OUT_FULL:
            rows = self.num_segments(opts_list[0].flags) * abi.SEG_DOUBLES
        else:
            rows = {abi.OUT_LAST: abi.SEG_DOUBLES, abi.OUT_OPD: 1, abi.OUT_FAN: 3, abi.OUT_HITS: 2}[mode]
        b_seg, b_op, b_pu = 8 * rows * R, 8 * R, 16 * R
        b_st, b_fs = (R + 15) // 16 * 16, (2 * R + 15) // 16 * 16
        per = b_seg + b_op + b_pu + b_st + b_fs
        lease = _pool.take(self.torch, per * n)
        whole = lease.array((per * n // 8,), np.float64)
        whole.fill(np.nan)
        outs, views = [], []
        for i in range(n):
            base = lease.ptr + i * per
            o = abi.Out()
            o.seg, o.op, o.pupil = base, base + b_seg, base + b_seg + b_op
            o.status = base + b_seg + b_op + b_pu
            o.fail_surf = o.status + b_st
            o.ld = R
            outs.append(o)
            off = i * per

            class _H:
                pass
            h = _H()
            h.R, h.out_mode = R, mode
            h.seg = lease.array((rows // abi.SEG_DOUBLES, abi.SEG_DOUBLES, R) if mode == abi.OUT_FULL
                                else (rows, R), np.float64, off)
            h.op = lease.array((R,), np.float64, off + b_seg)
            h.pupil = lease.array((2, R), np.float64, off + b_seg + b_op)
            h.status = lease.array((R,), np.uint8, off + b_seg + b_op + b_pu)
            h.fail_surf = lease.array((R,), np.int16, off + b_seg + b_op + b_pu + b_st)
            views.append(h)
        n, f_arr, w_arr, o_arr, out_arr = self._batch_args(flds, wvl_idxs, opts_list, outs)
        with self.torch.cuda.device(self.device):
            _check(self.lib.rox_trace_pupil_grids(self._handle, n, f_arr, w_arr, C.byref(grid),
                                                  o_arr, out_arr, self._stream()),
                   'rox_trace_pupil_grids')
        self.torch.cuda.current_stream(self.device).synchronize()
        return views

    @_in_flight
    def trace_pupil_grids_hits(self, flds, wvl_idxs, grid, opts_list):
        """ROX_OUT_HITS_COMPACT for several (field, wavelength) pairs in one launch: a list of
        (R_ok, 2) arrays, each a view of the pinned block the kernel packed that item's
        survivors into (one synchronise for all of them)."""
        R = grid_rays(grid)
        leases, outs = [], []
        for _ in flds:
            lease, o, _st = self._hits_out(R)
            leases.append(lease)
            outs.append(o)
        n, f_arr, w_arr, o_arr, out_arr = self._batch_args(flds, wvl_idxs, opts_list, outs)
        with self.torch.cuda.device(self.device):
            _check(self.lib.rox_trace_pupil_grids(self._handle, n, f_arr, w_arr, C.byref(grid),
                                                  o_arr, out_arr, self._stream()),
                   'rox_trace_pupil_grids')
        self.torch.cuda.current_stream(self.device).synchronize()
        return [lease.array((C.c_int64.from_address(lease.ptr + 16 * R).value, 2), np.float64)
                for lease in leases]

    @_in_flight
    def trace_pupil_list(self, fld, px, py, wvl_idx=0, opts=None, want_pupil=True,
                         nan_fill=False, out=None):
        t = self.torch
        opts = opts or make_opts()
        px = self._dev(px, t.float64)
        py = self._dev(py, t.float64)
        R = px.shape[0]
        res = self._result(R, opts, want_pupil, nan_fill, out)
        o = res.out_struct()
        with t.cuda.device(self.device):
            _check(self.lib.rox_trace_pupil_list(self._handle, C.byref(fld), R, px.data_ptr(),
                                                 py.data_ptr(), int(wvl_idx), C.byref(opts),
                                                 C.byref(o), self._stream()),
                   'rox_trace_pupil_list')
        res._keep = (px, py)
        return res

    @_in_flight
    def trace_pupil_np(self, fld, wvl_idx, opts, grid=None, px=None, py=None):
        """A SMALL pupil launch (a fan, a handful of rays, a figure-sized grid) with plain NumPy
        buffers through the library's ROX_HOST_POINTERS path: inputs and results cross in one
        device-mapped pinned block that the kernel reads and writes directly -- one launch, one
        synchronise, no torch tensors, no copy-engine transfers (a DeviceResult + to_host costs
        five device-to-host copies: ~1.2 ms of a RayFanFigure refresh went there).  Returns the
        host result (``seg`` [n_seg, 10, R] or [rows, R], ``op``, ``status``, ``fail_surf``,
        ``pupil``; slots the trace does not produce are NaN).  Callers keep it to batches of
        at most HOST_DIRECT_BYTES."""
        mode = opts.out_mode
        if grid is not None:
            R = grid_rays(grid)
        else:
            px = np.ascontiguousarray(px, dtype=np.float64)
            py = np.ascontiguousarray(py, dtype=np.float64)
            R = px.shape[0]
        if mode == abi.OUT_FULL:
            key = opts.flags & abi.FILTER_PHANTOMS
            nseg = self._nseg.get(key)
            if nseg is None:
                nseg = self._nseg[key] = self.num_segments(opts.flags)
            shape = (nseg, abi.SEG_DOUBLES, R)
        else:
            shape = ({abi.OUT_LAST: abi.SEG_DOUBLES, abi.OUT_OPD: 1, abi.OUT_FAN: 3, abi.OUT_HITS: 2}[mode], R)

        class _H:
            pass
        h = _H()
        h.R, h.out_mode = R, mode
        h.seg = np.empty(shape)
        h.op = np.empty(R)
        h.status = np.empty(R, dtype=np.uint8)
        h.fail_surf = np.empty(R, dtype=np.int16)
        h.pupil = np.empty((2, R))
        o = abi.Out()
        o.seg, o.op, o.status = h.seg.ctypes.data, h.op.ctypes.data, h.status.ctypes.data
        o.fail_surf, o.pupil, o.ld = h.fail_surf.ctypes.data, h.pupil.ctypes.data, R
        saved = opts.flags
        opts.flags = saved | abi.HOST_POINTERS
        try:
            with self.torch.cuda.device(self.device):
                if grid is not None:
                    rc = self.lib.rox_trace_pupil_grid(self._handle, C.byref(fld), C.byref(grid), int(wvl_idx),
                                                       C.byref(opts), C.byref(o), self._stream())
                else:
                    rc = self.lib.rox_trace_pupil_list(self._handle, C.byref(fld), R, px.ctypes.data,
                                                       py.ctypes.data, int(wvl_idx), C.byref(opts),
                                                       C.byref(o), self._stream())
        finally:
            opts.flags = saved
        _check(rc, 'rox_trace_pupil_grid' if grid is not None else 'rox_trace_pupil_list')
        return h

    @_in_flight
    def trace_rays_np(self, pt0, dir0, wvl_idx, opts):
        """explicit rays ([3, R] arrays, one wavelength index or R of them) with plain NumPy
        buffers through ROX_HOST_POINTERS -- the small-batch form of :meth:`trace_rays`, see
        :meth:`trace_pupil_np`"""
        pt0 = np.ascontiguousarray(pt0, dtype=np.float64)
        dir0 = np.ascontiguousarray(dir0, dtype=np.float64)
        R = pt0.shape[1]
        mode = opts.out_mode
        if mode == abi.OUT_FULL:
            shape = (self.num_segments(opts.flags), abi.SEG_DOUBLES, R)
        else:
            shape = ({abi.OUT_LAST: abi.SEG_DOUBLES, abi.OUT_HITS: 2}[mode], R)
        if np.ndim(wvl_idx) == 0:
            wi, wi_ptr, wi_all = None, None, int(wvl_idx)
        else:
            wi = np.ascontiguousarray(wvl_idx, dtype=np.int32)
            wi_ptr, wi_all = wi.ctypes.data, 0

        class _H:
            pass
        h = _H()
        h.R, h.out_mode, h.pupil = R, mode, None
        h.seg = np.empty(shape)
        h.op = np.empty(R)
        h.status = np.empty(R, dtype=np.uint8)
        h.fail_surf = np.empty(R, dtype=np.int16)
        o = abi.Out()
        o.seg, o.op, o.status = h.seg.ctypes.data, h.op.ctypes.data, h.status.ctypes.data
        o.fail_surf, o.ld = h.fail_surf.ctypes.data, R
        saved = opts.flags
        opts.flags = saved | abi.HOST_POINTERS
        try:
            with self.torch.cuda.device(self.device):
                rc = self.lib.rox_trace_rays(self._handle, R, pt0.ctypes.data, dir0.ctypes.data, wi_ptr,
                                             wi_all, C.byref(opts), C.byref(o), self._stream())
        finally:
            opts.flags = saved
        _check(rc, 'rox_trace_rays')
        return h

    # -- spot diagram: hits compacted on the device, written to pinned memory ----
    def _hits_out(self, R, want_status=False):
        """a pinned block for up to R (x, y) pairs + the count, and its rox_out"""
        lease = _pool.take(self.torch, 16 * R + 64)
        o = abi.Out()
        o.seg = lease.ptr
        o.n_hits = lease.ptr + 16 * R           # 8-byte count behind the pairs
        o.ld = R
        st = None
        if want_status:
            st = self.torch.empty((R,), dtype=self.torch.uint8, device=self.device)
            o.status = st.data_ptr()
        return lease, o, st

    def _hits_finish(self, lease, R):
        self.torch.cuda.current_stream(self.device).synchronize()
        n = C.c_int64.from_address(lease.ptr + 16 * R).value
        return lease.array((n, 2), np.float64)

    @_in_flight
    def trace_pupil_grid_hits(self, fld, grid, wvl_idx, opts):
        """ROX_OUT_HITS_COMPACT over a pupil grid: the (R_ok, 2) array of
        transverse aberrations, in ray order, as a NumPy array that views the
        pinned buffer the kernel wrote (no device->host copy, no host copy)."""
        R = grid_rays(grid)
        lease, o, _st = self._hits_out(R)
        with self.torch.cuda.device(self.device):
            _check(self.lib.rox_trace_pupil_grid(self._handle, C.byref(fld), C.byref(grid),
                                                 int(wvl_idx), C.byref(opts), C.byref(o),
                                                 self._stream()), 'rox_trace_pupil_grid')
        return self._hits_finish(lease, R)

    def hits_pack(self, cap, max_launches, dest=None):
        return HitsPack(self.torch, self.device, cap, max_launches, dest)

    @_in_flight
    def spot_stats(self, res, x_edges=None, y_edges=None):
        """rox_spot_stats over the device-resident output of a ROX_OUT_HITS launch
        (``res``: a DeviceResult of that mode -- rays that did not reach the image are skipped):
        ``(summary, hist)`` with ``summary`` a dict (n, centroid, rms_radius about the centroid,
        min / max per axis, the raw sums) and ``hist`` = numpy.histogram2d(x, y, bins=[x_edges,
        y_edges])[0] as uint32 (None without edges).  Nothing but 72 bytes and the histogram
        crosses PCIe (RayGeoPSF's 2-D histogram, rayoptics/mpl/analysisfigure.py:237-290)."""
        if res.out_mode != abi.OUT_HITS:
            raise EngineError('spot_stats reads the rows of a ROX_OUT_HITS launch')
        return self._spot_stats(res.seg.data_ptr(), res.ld, res.status.data_ptr(), None, res.R,
                                abi.SPOT_ROWS, x_edges, y_edges)

    def _spot_stats(self, seg_ptr, ld, status_ptr, n_hits_ptr, n, layout, x_edges, y_edges):
        summ = abi.SpotSummary()
        hist = None
        xe = ye = None
        nxe = nye = 0
        hp = None
        if x_edges is not None:
            xe = np.ascontiguousarray(x_edges, dtype=np.float64)
            ye = np.ascontiguousarray(y_edges, dtype=np.float64)
            nxe, nye = len(xe), len(ye)
            hist = np.empty((nxe - 1, nye - 1), dtype=np.uint32)
            hp = hist.ctypes.data
        with self.torch.cuda.device(self.device):
            _check(self.lib.rox_spot_stats(seg_ptr, int(ld), status_ptr, n_hits_ptr, int(n), int(layout),
                                           None if xe is None else xe.ctypes.data, nxe,
                                           None if ye is None else ye.ctypes.data, nye,
                                           C.byref(summ), hp, self._stream()), 'rox_spot_stats')
        n_ok = int(summ.n)
        out = {'n': n_ok, 'sum': (summ.sum[0], summ.sum[1]), 'sum_sq': (summ.sum_sq[0], summ.sum_sq[1]),
               'min': (summ.min[0], summ.min[1]), 'max': (summ.max[0], summ.max[1])}
        if n_ok:
            cx, cy = summ.sum[0] / n_ok, summ.sum[1] / n_ok
            var = (summ.sum_sq[0] + summ.sum_sq[1]) / n_ok - (cx * cx + cy * cy)
            out['centroid'] = (cx, cy)
            out['rms_radius'] = float(np.sqrt(max(var, 0.0)))
        else:
            out['centroid'], out['rms_radius'] = (float('nan'), float('nan')), float('nan')
        return out, hist

    @_in_flight
    def trace_pupil_grid_hits_append(self, fld, grid, wvl_idx, opts, pack):
        """enqueue one ROX_OUT_HITS_COMPACT | ROX_HITS_APPEND launch behind the pairs
        ``pack`` already holds; nothing is synchronised"""
        R = grid_rays(grid)
        if pack.rays + R > pack.cap or pack.n_launches >= pack.cum.numel():
            raise EngineError(f'HitsPack full: {pack.rays} + {R} rays into a capacity of {pack.cap}')
        if not (opts.flags & abi.HITS_APPEND) or opts.out_mode != abi.OUT_HITS_COMPACT:
            raise EngineError('trace_pupil_grid_hits_append needs OUT_HITS_COMPACT | HITS_APPEND')
        o = abi.Out()
        o.seg = pack.seg_ptr
        o.n_hits = pack.count.data_ptr()
        o.ld = pack.cap
        with self.torch.cuda.device(self.device):
            _check(self.lib.rox_trace_pupil_grid(self._handle, C.byref(fld), C.byref(grid),
                                                 int(wvl_idx), C.byref(opts), C.byref(o),
                                                 self._stream()), 'rox_trace_pupil_grid')
            k = pack.n_launches
            pack.cum[k:k + 1].copy_(pack.count, non_blocking=True)      # same stream: ordered
        pack.n_launches += 1
        pack.rays += R
        return pack

    @_in_flight
    def trace_pupil_grid_hits_at(self, fld, grid, wvl_idx, opts, seg_ptr, cap, n_hits_ptr):
        """enqueue one ROX_OUT_HITS_COMPACT launch (not appending) whose packed pairs go to
        ``seg_ptr`` (room for ``cap`` pairs, device or device-visible memory) and whose count
        goes to ``n_hits_ptr`` (int64, device-visible); nothing is synchronised.  The unit of
        the pipelined sharded spot diagram (dist.trace_spot_sharded)"""
        if opts.out_mode != abi.OUT_HITS_COMPACT or (opts.flags & abi.HITS_APPEND):
            raise EngineError('trace_pupil_grid_hits_at needs OUT_HITS_COMPACT without HITS_APPEND')
        o = abi.Out()
        o.seg, o.n_hits, o.ld = int(seg_ptr), int(n_hits_ptr), int(cap)
        with self.torch.cuda.device(self.device):
            _check(self.lib.rox_trace_pupil_grid(self._handle, C.byref(fld), C.byref(grid),
                                                 int(wvl_idx), C.byref(opts), C.byref(o),
                                                 self._stream()), 'rox_trace_pupil_grid')

    def copy_async(self, dst_ptr, src_ptr, nbytes, stream=None):
        """rox_copy_async in the order of ``stream`` (a torch stream; None = the current one)"""
        st = self._stream() if stream is None else C.c_void_p(stream.cuda_stream)
        with self.torch.cuda.device(self.device):
            _check(self.lib.rox_copy_async(C.c_void_p(int(dst_ptr)), C.c_void_p(int(src_ptr)),
                                           C.c_size_t(int(nbytes)), st), 'rox_copy_async')

    # -- host memory other processes share (dist.HostSegment) ------------------------
    def pin_host_memory(self, ptr, nbytes):
        """register [ptr, ptr+nbytes) with the HIP runtime; returns the pointer kernels
        on this device write it through"""
        dev = C.c_void_p()
        with self.torch.cuda.device(self.device):
            _check(self.lib.rox_pin_host_memory(C.c_void_p(ptr), C.c_size_t(nbytes), C.byref(dev)),
                   'rox_pin_host_memory')
        return dev.value

    def unpin_host_memory(self, ptr):
        _check(self.lib.rox_unpin_host_memory(C.c_void_p(ptr)), 'rox_unpin_host_memory')

    @_in_flight
    def trace_pupil_list_hits(self, fld, px, py, wvl_idx, opts):
        t = self.torch
        px = self._dev(px, t.float64)
        py = self._dev(py, t.float64)
        R = px.shape[0]
        lease, o, _st = self._hits_out(R)
        with t.cuda.device(self.device):
            _check(self.lib.rox_trace_pupil_list(self._handle, C.byref(fld), R, px.data_ptr(),
                                                 py.data_ptr(), int(wvl_idx), C.byref(opts),
                                                 C.byref(o), self._stream()),
                   'rox_trace_pupil_list')
        return self._hits_finish(lease, R)

    @_in_flight
    def trace_rays_hits(self, pt0, dir0, wvl_idx, opts):
        t = self.torch
        pt0 = self._dev(pt0, t.float64)
        dir0 = self._dev(dir0, t.float64)
        R = pt0.shape[1]
        if np.ndim(wvl_idx) == 0 and not isinstance(wvl_idx, t.Tensor):
            wi, wi_ptr, wi_all = None, None, int(wvl_idx)
        else:
            wi = self._dev(wvl_idx, t.int32)
            wi_ptr, wi_all = wi.data_ptr(), 0
        lease, o, _st = self._hits_out(R)
        with t.cuda.device(self.device):
            _check(self.lib.rox_trace_rays(self._handle, R, pt0.data_ptr(), dir0.data_ptr(),
                                           wi_ptr, wi_all, C.byref(opts), C.byref(o),
                                           self._stream()), 'rox_trace_rays')
        return self._hits_finish(lease, R)

    # -- one ray (raytrace.trace) -----------------------------------------------
    @_in_flight
    def trace_one(self, pt0, dir0, wvl_idx, opts):
        """one explicit ray, FULL packets, through the library's ROX_HOST_POINTERS
        path: the ray and its packet live in one NumPy block; the library copies
        the 48 input bytes into a device-mapped pinned block, the kernel reads and
        writes that block directly, and the packet is copied back (one launch, one
        synchronise, no copy-engine transfer).  Slots past a failure are NaN."""
        key = opts.flags & abi.FILTER_PHANTOMS
        nseg = self._nseg.get(key)
        if nseg is None:
            nseg = self._nseg[key] = self.num_segments(opts.flags)
        buf = np.empty(8 + abi.SEG_DOUBLES * nseg)
        buf[0:3] = pt0
        buf[3:6] = dir0
        base = buf.ctypes.data
        o = abi.Out()
        o.seg, o.op, o.status, o.fail_surf, o.ld = base + 64, base + 48, base + 56, base + 58, 1
        saved = opts.flags
        opts.flags = saved | abi.HOST_POINTERS
        try:
            if self.torch.cuda.current_device() == self.device.index:
                rc = self.lib.rox_trace_rays(self._handle, 1, base, base + 24, None, int(wvl_idx),
                                             C.byref(opts), C.byref(o), None)
            else:
                with self.torch.cuda.device(self.device):
                    rc = self.lib.rox_trace_rays(self._handle, 1, base, base + 24, None,
                                                 int(wvl_idx), C.byref(opts), C.byref(o), None)
        finally:
            opts.flags = saved
        _check(rc, 'rox_trace_rays')

        class _H:
            pass
        h = _H()
        h.R, h.out_mode, h.pupil = 1, abi.OUT_FULL, None
        h.seg = buf[8:].reshape(nseg, abi.SEG_DOUBLES, 1)
        h.op = buf[6:7]
        h.status = buf[7:8].view(np.uint8)[0:1]
        h.fail_surf = buf[7:8].view(np.int16)[1:2]
        return h

    # -- chief-ray aiming ---------------------------------------------------------
    @_in_flight
    def aim_chief_rays(self, probs, eps=1.0e-12):
        """probs: sequence of abi.Aim -> (aim float64[n, 2] = (x1, y1), result int32[n])"""
        n = len(probs)
        arr = (abi.Aim * n)(*probs)
        aim = np.zeros((n, 2))
        result = np.zeros(n, dtype=np.int32)
        with self.torch.cuda.device(self.device):
            _check(self.lib.rox_aim_chief_rays(self._handle, n, arr, float(eps),
                                               aim.ctypes.data, result.ctypes.data,
                                               self._stream()), 'rox_aim_chief_rays')
        return aim, result

    @_in_flight
    def iterate_pupil_rays(self, probs, eps=1.0e-12):
        """probs: sequence of abi.PupilIter -> start_r float64[n] (vigcalc.iterate_pupil_ray)"""
        n = len(probs)
        arr = (abi.PupilIter * n)(*probs)
        out = np.zeros(n)
        with self.torch.cuda.device(self.device):
            _check(self.lib.rox_iterate_pupil_rays(self._handle, n, arr, float(eps), out.ctypes.data,
                                                   self._stream()), 'rox_iterate_pupil_rays')
        return out

    @_in_flight
    def iterate_ray_raw(self, probs, eps=1.0e-12):
        """trace.iterate_ray_raw over the path this engine's table describes: (aim [n, 2],
        result [n], last_xy [n, 2] = pupil-plane coordinates of the last trial ray the
        iteration evaluated, last_status [n] = its trace status)"""
        n = len(probs)
        arr = (abi.Aim * n)(*probs)
        aim = np.zeros((n, 2))
        result = np.zeros(n, dtype=np.int32)
        last_xy = np.zeros((n, 2))
        last_st = np.zeros(n, dtype=np.int32)
        with self.torch.cuda.device(self.device):
            _check(self.lib.rox_iterate_ray_raw(self._handle, n, arr, float(eps), aim.ctypes.data,
                                                result.ctypes.data, last_xy.ctypes.data,
                                                last_st.ctypes.data, self._stream()),
                   'rox_iterate_ray_raw')
        return aim, result, last_xy, last_st

    @_in_flight
    def find_real_enp(self, probs, eps=1.0e-12):
        """probs: sequence of abi.Enp -> (z float64[n, 2] = (z_enp, z of the last trial ray),
        result int32[n] = abi.ENP_*)"""
        n = len(probs)
        arr = (abi.Enp * n)(*probs)
        z = np.zeros((n, 2))
        result = np.zeros(n, dtype=np.int32)
        with self.torch.cuda.device(self.device):
            _check(self.lib.rox_find_real_enp(self._handle, n, arr, float(eps),
                                              z.ctypes.data, result.ctypes.data,
                                              self._stream()), 'rox_find_real_enp')
        return z, result

    @_in_flight
    def calc_vignetting(self, probs, eps=1.0e-12):
        """probs: sequence of abi.Vig -> (vig float64[n], clip_surf int32[n])"""
        n = len(probs)
        arr = (abi.Vig * n)(*probs)
        vig = np.zeros(n)
        clip = np.zeros(n, dtype=np.int32)
        with self.torch.cuda.device(self.device):
            _check(self.lib.rox_calc_vignetting(self._handle, n, arr, float(eps),
                                                vig.ctypes.data, clip.ctypes.data,
                                                self._stream()), 'rox_calc_vignetting')
        return vig, clip

    def time_pupil_grid_sustained(self, fld, grid, wvl_idx, opts, out, launches=20, batches=7,
                                  warm_ms=120.0):
        """median over `batches` of :meth:`time_pupil_grid` after `warm_ms` of
        back-to-back launches: the GPU's clocks take tens of milliseconds of
        continuous work to settle (a cold 20-launch batch reads 10-30 % slow,
        tools/sustained_probe.py), so steady-state kernel times are quoted"""
        import time as _time
        t0 = _time.perf_counter()
        while (_time.perf_counter() - t0) * 1e3 < warm_ms:
            self.time_pupil_grid(fld, grid, wvl_idx, opts, out, launches)
        ts = sorted(self.time_pupil_grid(fld, grid, wvl_idx, opts, out, launches)
                    for _ in range(batches))
        return ts[len(ts) // 2]

    @_in_flight
    def time_pupil_grid(self, fld, grid, wvl_idx, opts, out, launches):
        """mean duration (ms) of the trace kernel over `launches` launches,
        from HIP events recorded on the launch stream"""
        ms = C.c_double()
        o = out.out_struct()
        with self.torch.cuda.device(self.device):
            _check(self.lib.rox_time_pupil_grid(self._handle, C.byref(fld), C.byref(grid),
                                                int(wvl_idx), C.byref(opts), C.byref(o),
                                                self._stream(), int(launches), C.byref(ms)),
                   'rox_time_pupil_grid')
        return ms.value
