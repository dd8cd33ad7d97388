"""TEST DOUBLE: an object with TraceEngine's interface whose launches are
served by the CPU oracle.  It exists so that the *host logic* of the drop-in
layer (result views, filters, grid ordering, install/uninstall) can be
exercised against the live reference in the build container, which has no GPU.
It is injected through rayoptics_amd.session._set_engine_factory() by tests only; the
product default is the HIP engine, which raises without a GPU."""
import numpy as np

from oracle import oracle
from rayoptics_amd import abi


class _Res:
    """DeviceResult look-alike: numpy on to_host(), torch views for dist.py"""

    def __init__(self, host):
        self._h = host

    def to_host(self, want=None):
        return self._h

    @property
    def seg(self):
        import torch
        return torch.from_numpy(self._h.seg)

    @property
    def status(self):
        import torch
        return torch.from_numpy(self._h.status)

    @property
    def op(self):
        import torch
        return torch.from_numpy(self._h.op)

    @property
    def fail_surf(self):
        import torch
        return torch.from_numpy(self._h.fail_surf)


class _Pack:
    """engine.HitsPack look-alike on host memory"""

    def __init__(self, cap, max_launches, dest=None):
        import ctypes as C
        import torch
        if dest is None:
            self.cap = int(cap)
            self.xy = torch.empty((self.cap, 2), dtype=torch.float64)
            self._np = self.xy.numpy()
        else:
            self.cap = int(dest[1])
            self.xy = None
            self._np = np.ctypeslib.as_array(
                (C.c_double * (2 * max(self.cap, 1))).from_address(int(dest[0]))).reshape(-1, 2)
        self._cum, self.n, self.rays, self.max_launches = [], 0, 0, max_launches

    def counts(self):
        return np.diff(np.concatenate([[0], self._cum])).astype(np.int64)


class OracleEngine:
    def __init__(self, table, device=None):
        from rayoptics_amd.engine import ModelMemo
        self.table = table
        self.device = 'cpu'
        self.memo = ModelMemo()

    def hits_pack(self, cap, max_launches, dest=None):
        return _Pack(cap, max_launches, dest)

    def trace_pupil_grid_hits_append(self, fld, grid, wvl_idx, opts, pack):
        assert opts.flags & abi.HITS_APPEND and opts.out_mode == abi.OUT_HITS_COMPACT
        assert len(pack._cum) < pack.max_launches
        o = oracle.make_opts(flags=opts.flags & ~abi.HITS_APPEND, out_mode=opts.out_mode,
                             first_surf=opts.first_surf, last_surf=opts.last_surf, eps=opts.eps,
                             fuzz=opts.fuzz, foc=opts.foc, image_pt=tuple(opts.image_pt))
        hits = oracle.trace_pupil_grid(self.table, fld, grid, wvl_idx, o).hits
        pack.rays += (grid.row_count or grid.num) * grid.num
        assert pack.rays <= pack.cap
        pack._np[pack.n:pack.n + len(hits)] = hits
        pack.n += len(hits)
        pack._cum.append(pack.n)
        return pack

    def trace_pupil_grid_hits_at(self, fld, grid, wvl_idx, opts, seg_ptr, cap, n_hits_ptr):
        import ctypes as C
        assert opts.out_mode == abi.OUT_HITS_COMPACT and not (opts.flags & abi.HITS_APPEND)
        hits = oracle.trace_pupil_grid(self.table, fld, grid, wvl_idx, opts).hits
        assert len(hits) <= cap
        if len(hits):
            dst = np.ctypeslib.as_array((C.c_double * (2 * len(hits))).from_address(int(seg_ptr)))
            dst[:] = hits.ravel()
        C.c_int64.from_address(int(n_hits_ptr)).value = len(hits)

    def copy_async(self, dst_ptr, src_ptr, nbytes, stream=None):
        import ctypes as C
        C.memmove(int(dst_ptr), int(src_ptr), int(nbytes))

    def pin_host_memory(self, ptr, nbytes):
        return ptr

    def unpin_host_memory(self, ptr):
        pass

    def close(self):
        pass

    def num_segments(self, flags=0):
        n = self.table.n_ifcs
        if flags & abi.FILTER_PHANTOMS:
            n -= sum(1 for i, r in enumerate(self.table.rows)
                     if r.mode == abi.PHANTOM and 0 < i < self.table.n_ifcs - 1)
        return n

    def _trim(self, res, opts):
        if opts.out_mode == abi.OUT_FULL:
            res.seg = res.seg[:self.num_segments(opts.flags)]
        return _Res(res)

    def trace_rays(self, pt0, dir0, wvl_idx=0, opts=None, **kw):
        return self._trim(oracle.trace_rays(self.table, np.asarray(pt0), np.asarray(dir0),
                                            wvl_idx, opts), opts)

    def trace_pupil_grid(self, fld, grid, wvl_idx=0, opts=None, **kw):      # (out= ignored)
        return self._trim(oracle.trace_pupil_grid(self.table, fld, grid, wvl_idx, opts), opts)

    def trace_pupil_list(self, fld, px, py, wvl_idx=0, opts=None, **kw):
        return self._trim(oracle.trace_pupil_list(self.table, fld, px, py, wvl_idx, opts), opts)

    def trace_one(self, pt0, dir0, wvl_idx, opts):
        h = oracle.trace_rays(self.table, np.asarray(pt0, dtype=float).reshape(3, 1),
                              np.asarray(dir0, dtype=float).reshape(3, 1), wvl_idx, opts)
        h.seg = h.seg[:self.num_segments(opts.flags)]
        return h

    # ROX_OUT_HITS_COMPACT entries: the (R_ok, 2) array
    def trace_pupil_grid_hits(self, fld, grid, wvl_idx, opts):
        return oracle.trace_pupil_grid(self.table, fld, grid, wvl_idx, opts).hits.copy()

    def trace_pupil_grids_hits(self, flds, wvl_idxs, grid, opts_list):
        return [self.trace_pupil_grid_hits(f, grid, w, o) for f, w, o in zip(flds, wvl_idxs, opts_list)]

    def trace_pupil_grids(self, flds, wvl_idxs, grid, opts_list, **kw):
        return [self.trace_pupil_grid(f, grid, w, o) for f, w, o in zip(flds, wvl_idxs, opts_list)]

    def trace_pupil_np(self, fld, wvl_idx, opts, grid=None, px=None, py=None):
        if grid is not None:
            return self.trace_pupil_grid(fld, grid, wvl_idx, opts).to_host()
        return self.trace_pupil_list(fld, px, py, wvl_idx, opts).to_host()

    def trace_pupil_grids_host(self, flds, wvl_idxs, grid, opts_list):
        return [self.trace_pupil_grid(f, grid, w, o).to_host() for f, w, o in zip(flds, wvl_idxs, opts_list)]

    def trace_pupil_list_hits(self, fld, px, py, wvl_idx, opts):
        return oracle.trace_pupil_list(self.table, fld, px, py, wvl_idx, opts).hits.copy()

    def trace_rays_hits(self, pt0, dir0, wvl_idx, opts):
        return oracle.trace_rays(self.table, np.asarray(pt0), np.asarray(dir0), wvl_idx,
                                 opts).hits.copy()

    def aim_chief_rays(self, probs, eps=1.0e-12):
        return oracle.aim_chief_rays(self.table, probs, eps)

    def iterate_pupil_rays(self, probs, eps=1.0e-12):
        return oracle.iterate_pupil_rays(self.table, probs, eps)

    def iterate_ray_raw(self, probs, eps=1.0e-12):
        return oracle.iterate_ray_raw(self.table, probs, eps)

    def find_real_enp(self, probs, eps=1.0e-12):
        return oracle.find_real_enp(self.table, probs, eps)

    def calc_vignetting(self, probs, eps=1.0e-12):
        return oracle.calc_vignetting(self.table, probs, eps)
