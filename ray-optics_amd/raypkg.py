"""Lazy RayPkg / RaySeg views over SoA trace results.

The reference returns, per ray, a Python list of N lists
``[p(ndarray3), d(ndarray3), dst(float), nrml(ndarray3)]`` plus ``op_delta`` and
``wvl`` (rayoptics/raytr/raytrace.py:191, 260-264;
rayoptics/raytr/__init__.py:24-40).  Materialising that for a million rays is
tens of millions of Python objects, so the host side keeps the SoA arrays and
hands out sequence views that build a segment only when it is indexed.  Every
access pattern reference consumers use works on them: ``pkg[mc.ray]``,
``ray[-1][mc.p]``, ``ray[k][mc.d][2]``, ``len(ray)``, iteration, tuple
unpacking ``ray, op, wvl = pkg`` and ``RaySeg(*rs)``.
"""
from collections import namedtuple

import numpy as np

from . import abi

try:
    from rayoptics.raytr import RayPkg, RaySeg, RayResult
except Exception:
    RayResult = namedtuple('RayResult', ['pkg', 'err'])
    RayPkg = namedtuple('RayPkg', ['ray', 'op', 'wvl'])
    RaySeg = namedtuple('RaySeg', ['p', 'd', 'dst', 'nrml'])


class LazyRay:
    """sequence of ray segments of one ray; segment k is built on access.  The first segment
    asked for is gathered alone (a spot diagram reads ray[-1] of a million rays and nothing
    else); a second request gathers the ray's whole [n, 10] block once and every segment after
    that is three views of it (vigcalc.max_aperture_at_surf walks every segment of every
    boundary ray: 370 segments per model update)."""
    __slots__ = ('_seg', '_r', '_n', '_named', '_blk', '_asked')

    def __init__(self, seg, r, nseg, named=False):
        self._seg = seg         # numpy [K, 10, R] (FULL) or [1, 10, R]
        self._r = r
        self._n = nseg
        self._named = named
        self._blk = None
        self._asked = False

    def __len__(self):
        return self._n

    def _make(self, k):
        if self._blk is not None:
            s = self._blk[k]
            item = [s[0:3], s[3:6], float(s[6]), s[7:10]]
        elif self._asked and self._n > 1:
            self._blk = np.ascontiguousarray(self._seg[:self._n, :, self._r])
            s = self._blk[k]
            item = [s[0:3], s[3:6], float(s[6]), s[7:10]]
        else:
            self._asked = True
            s = self._seg[k, :, self._r]        # (fancy-free gather: a fresh 10-vector)
            item = [s[0:3].copy(), s[3:6].copy(), float(s[6]), s[7:10].copy()]
        return RaySeg(*item) if self._named else item

    def __getitem__(self, k):
        if isinstance(k, slice):
            return [self._make(i) for i in range(*k.indices(self._n))]
        if k < 0:
            k += self._n
        if not 0 <= k < self._n:
            raise IndexError('ray segment index out of range')
        return self._make(k)

    def to_list(self):
        self._asked = True
        return [self._make(k) for k in range(self._n)]

    def __iter__(self):
        self._asked = True
        for k in range(self._n):
            yield self._make(k)

    def __repr__(self):
        return f'LazyRay({self._n} segments, ray {self._r})'


class HostPackets:
    """host-side (numpy) copy of one trace call's outputs + the bookkeeping
    needed to serve reference-shaped results for ray ``r``"""

    def __init__(self, host, table, flags, out_mode, wvl_of_ray):
        # [K, 10, R] (FULL), or one pseudo-segment [1, rows, R] for LAST/HITS/OPD
        self.seg = host.seg if host.seg.ndim == 3 else host.seg[None]
        self.op = host.op
        self.status = host.status
        self.fail_surf = host.fail_surf
        self.pupil = host.pupil
        self.table = table
        self.out_mode = out_mode
        self.flags = flags
        self._wvl = wvl_of_ray          # float or array of floats
        # per-ray accessors run once per ray of a list-shaped result: plain Python numbers,
        # not NumPy scalar indexing (100 000 rays: 2.4 us -> 0.1 us per ray for the wavelength alone)
        self._wvl_scalar = float(wvl_of_ray) if np.ndim(wvl_of_ray) == 0 else None
        self._st_l = self._fs_l = self._op_l = None
        N = table.n_ifcs
        filt = bool(flags & abi.FILTER_PHANTOMS)
        nb, nxt = [], 0
        for i, row in enumerate(table.rows):
            nb.append(nxt)
            if not (filt and row.mode == abi.PHANTOM and 0 < i < N - 1):
                nxt += 1
        self._nslots_before = nb
        self._n_full = nxt

    def wvl(self, r):
        return self._wvl_scalar if self._wvl_scalar is not None else float(self._wvl[r])

    def _lists(self):
        if self._st_l is None:
            self._st_l = self.status.tolist()
            self._fs_l = self.fail_surf.tolist() if self.fail_surf is not None else None
            self._op_l = self.op.tolist() if self.op is not None else None
        return self._st_l

    def status_of(self, r):
        return (self._st_l or self._lists())[r]

    def nseg(self, r):
        st = (self._st_l or self._lists())[r]
        if self.out_mode != abi.OUT_FULL:
            return 1 if st == abi.OK else 0
        if st == abi.OK:
            return self._n_full
        s = self._fs_l[r]
        if s <= 0:
            return 0
        if st == abi.MISSED_SURFACE:        # raytrace.py:231-237
            return self._nslots_before[s - 1] + 1
        return self._nslots_before[s] + 1   # raytrace.py:239-257

    def pkg(self, r, named=False):
        ray = LazyRay(self.seg, r, self.nseg(r), named)
        op, wvl = self._op_l[r], self.wvl(r)
        return RayPkg(ray, op, wvl) if named else (ray, op, wvl)

    def error(self, r, ifcs=None, with_pkg=True, named=True):
        """the exception object trace_safe would report for a failed ray"""
        from .traceerror import make_error
        st, s = int(self.status[r]), int(self.fail_surf[r])
        ifc = ifcs[s] if ifcs is not None and 0 <= s < len(ifcs) else None
        pkg = self.pkg(r, named) if with_pkg and self.out_mode == abi.OUT_FULL else None
        int_pt = inc_dir = normal = n_in = n_out = None
        if pkg is not None and st != abi.MISSED_SURFACE and len(pkg[0]):
            last = pkg[0][-1]               # [inc_pt, before_dir, 0.0, normal], raytrace.py:239-257
            int_pt = last[0]
            if st in (abi.TIR, abi.EVANESCENT) and s >= 1:
                # bend()/phase() were given b4_dir = rt.dot(before_dir) (raytrace.py:172),
                # the unit normal and the indices either side of the interface
                row = self.table.rows[s - 1]
                rt = np.array(list(row.rt)).reshape(3, 3)
                if row.rt_order == abi.RT_F_ORDER:
                    rt = np.asfortranarray(rt)      # the dgemv chain of the transpose view
                inc_dir = rt.dot(last[1])
                normal = last[3]
                wi = self.table.wvl_index(self.wvl(r))
                n_in, n_out = float(self.table.n_table[wi, s - 1]), float(self.table.n_table[wi, s])
        return make_error(st, s, ifc, pkg, int_pt, inc_dir, normal, n_in, n_out)
