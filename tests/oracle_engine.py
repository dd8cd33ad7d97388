"""TEST DOUBLE: an object with TraceEngine's interface whose launches are
served by the CPU oracle.  It exists so that the *host logic* of the drop-in
layer (result views, filters, grid ordering, install/uninstall) can be
exercised against the live reference in the build container, which has no GPU.
It is injected through rayoptics_amd.session.ENGINE_FACTORY by tests only; the
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


class OracleEngine:
    def __init__(self, table, device=None):
        self.table = table

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

    def trace_pupil_list_hits(self, fld, px, py, wvl_idx, opts):
        return oracle.trace_pupil_list(self.table, fld, px, py, wvl_idx, opts).hits.copy()

    def trace_rays_hits(self, pt0, dir0, wvl_idx, opts):
        return oracle.trace_rays(self.table, np.asarray(pt0), np.asarray(dir0), wvl_idx,
                                 opts).hits.copy()

    def aim_chief_rays(self, probs, eps=1.0e-12):
        return oracle.aim_chief_rays(self.table, probs, eps)

    def calc_vignetting(self, probs, eps=1.0e-12):
        return oracle.calc_vignetting(self.table, probs, eps)
