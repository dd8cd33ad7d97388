"""TEST INFRASTRUCTURE ONLY -- ctypes binding of librox_oracle.so (the plain-C
CPU restatement in rox_oracle.c).  Imported by tests/, by
__graft_entry__.smoke() and by bench.py's cpu_baseline leg, never by the
product package.  Struct definitions are shared with the product because they
are the data format under test."""
import ctypes as C
import os
import subprocess

import numpy as np

from rayoptics_amd import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build():
    subprocess.check_call(['make', '-s', '-C', _HERE, 'librox_oracle.so'])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, 'librox_oracle.so')
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        vp, i32, i64 = C.c_void_p, C.c_int32, C.c_int64
        P = C.POINTER
        L.rox_oracle_trace_rays.restype = C.c_int
        L.rox_oracle_trace_rays.argtypes = [P(abi.Surface), i32, vp, vp, i32, i64,
                                            vp, vp, vp, i32, P(abi.Opts), P(abi.Out)]
        L.rox_oracle_trace_pupil_grid.restype = C.c_int
        L.rox_oracle_trace_pupil_grid.argtypes = [P(abi.Surface), i32, vp, vp, i32,
                                                  P(abi.Field), P(abi.Grid), i32,
                                                  P(abi.Opts), P(abi.Out)]
        L.rox_oracle_trace_pupil_list.restype = C.c_int
        L.rox_oracle_trace_pupil_list.argtypes = [P(abi.Surface), i32, vp, vp, i32,
                                                  P(abi.Field), i64, vp, vp, i32,
                                                  P(abi.Opts), P(abi.Out)]
        L.rox_oracle_intersect.restype = C.c_int
        L.rox_oracle_intersect.argtypes = [P(abi.Surface), vp, vp, C.c_double,
                                           C.c_double, P(C.c_double), vp, vp]
        L.rox_oracle_aim_chief_rays.restype = C.c_int
        L.rox_oracle_aim_chief_rays.argtypes = [P(abi.Surface), i32, vp, vp, i32, i32,
                                                P(abi.Aim), C.c_double, vp, vp]
        L.rox_oracle_iterate_ray_raw.restype = C.c_int
        L.rox_oracle_iterate_ray_raw.argtypes = [P(abi.Surface), i32, vp, vp, i32, i32,
                                                 P(abi.Aim), C.c_double, vp, vp, vp, vp]
        L.rox_oracle_iterate_pupil_rays.restype = C.c_int
        L.rox_oracle_iterate_pupil_rays.argtypes = [P(abi.Surface), i32, vp, vp, i32, i32,
                                                    P(abi.PupilIter), C.c_double, vp]
        L.rox_oracle_find_real_enp.restype = C.c_int
        L.rox_oracle_find_real_enp.argtypes = [P(abi.Surface), i32, vp, vp, i32, i32,
                                               P(abi.Enp), C.c_double, vp, vp]
        L.rox_oracle_calc_vignetting.restype = C.c_int
        L.rox_oracle_calc_vignetting.argtypes = [P(abi.Surface), i32, vp, vp, i32, i32,
                                                 P(abi.Vig), C.c_double, vp, vp]
        _LIB = L
    return _LIB


def _wvls(table):
    """wavelengths (nm) as a float64 array kept alive on the table"""
    w = getattr(table, '_wvls_arr', None)
    if w is None:
        w = table._wvls_arr = np.ascontiguousarray(table.wvls, dtype=np.float64)
    return w


class HostResult:
    """numpy-backed SoA result buffers in the product's output layout,
    NaN / 0xff pre-filled so untouched slots compare equal."""

    def __init__(self, n_seg_rows, R, out_mode, want_pupil=False):
        self.R = R
        self.out_mode = out_mode
        if out_mode == abi.OUT_FULL:
            shape = (n_seg_rows, abi.SEG_DOUBLES, R)
        elif out_mode == abi.OUT_LAST:
            shape = (abi.SEG_DOUBLES, R)
        elif out_mode == abi.OUT_OPD:
            shape = (1, R)
        elif out_mode == abi.OUT_FAN:
            shape = (3, R)
        elif out_mode == abi.OUT_HITS_COMPACT:
            shape = (R, 2)              # packed (x, y) pairs; n_hits of them are valid
        else:
            shape = (2, R)
        self.n_hits = np.zeros(1, dtype=np.int64)
        self.seg = np.full(shape, np.nan)
        self.op = np.full(R, np.nan)
        self.status = np.full(R, 255, dtype=np.uint8)
        self.fail_surf = np.full(R, -2, dtype=np.int16)
        self.pupil = np.full((2, R), np.nan) if want_pupil else None

    def out_struct(self):
        o = abi.Out()
        o.seg = self.seg.ctypes.data
        o.op = self.op.ctypes.data
        o.status = self.status.ctypes.data
        o.fail_surf = self.fail_surf.ctypes.data
        o.pupil = self.pupil.ctypes.data if self.pupil is not None else None
        o.ld = self.R
        o.n_hits = self.n_hits.ctypes.data
        return o

    @property
    def hits(self):
        """HITS_COMPACT: the (n_hits, 2) array"""
        return self.seg[:int(self.n_hits[0])]


def make_opts(flags=abi.INTERSECT_OBJ, out_mode=abi.OUT_FULL, first_surf=0,
              last_surf=-1, eps=1.0e-12, fuzz=1e-5, foc=0.0, image_pt=(0., 0.), wf=None):
    o = abi.Opts()
    o.flags, o.out_mode = flags, out_mode
    o.first_surf, o.last_surf = first_surf, last_surf
    o.eps, o.fuzz, o.foc = eps, fuzz, foc
    o.image_pt[0], o.image_pt[1] = image_pt
    if wf is not None:
        o.wf = wf
    return o


def trace_rays(table, pt0, dir0, wvl_idx, opts):
    """pt0, dir0: float64 [3][R]; wvl_idx: int or int32[R]"""
    pt0 = np.ascontiguousarray(pt0, dtype=np.float64)
    dir0 = np.ascontiguousarray(dir0, dtype=np.float64)
    R = pt0.shape[1]
    res = HostResult(table.n_ifcs, R, opts.out_mode)
    out = res.out_struct()
    if np.ndim(wvl_idx) == 0:
        wi_ptr, wi_all = None, int(wvl_idx)
    else:
        wi = np.ascontiguousarray(wvl_idx, dtype=np.int32)
        wi_ptr, wi_all = wi.ctypes.data, 0
    rc = lib().rox_oracle_trace_rays(table.rows, table.n_ifcs,
                                     table.n_table.ctypes.data, _wvls(table).ctypes.data,
                                     len(table.wvls),
                                     R, pt0.ctypes.data, dir0.ctypes.data,
                                     wi_ptr, wi_all, C.byref(opts), C.byref(out))
    if rc:
        raise RuntimeError(f'oracle error {rc}')
    return res


def trace_pupil_grid(table, fld, grid, wvl_idx, opts):
    R = grid.num if grid.kind == abi.GRID_FAN else (grid.row_count or grid.num) * grid.num
    res = HostResult(table.n_ifcs, R, opts.out_mode, want_pupil=True)
    out = res.out_struct()
    rc = lib().rox_oracle_trace_pupil_grid(table.rows, table.n_ifcs,
                                           table.n_table.ctypes.data,
                                           _wvls(table).ctypes.data,
                                           len(table.wvls), C.byref(fld),
                                           C.byref(grid), wvl_idx,
                                           C.byref(opts), C.byref(out))
    if rc:
        raise RuntimeError(f'oracle error {rc}')
    return res


def trace_pupil_list(table, fld, px, py, wvl_idx, opts, res=None):
    """res: reuse a HostResult (bench.py times the C call, not the 1 GB fill)"""
    px = np.ascontiguousarray(px, dtype=np.float64)
    py = np.ascontiguousarray(py, dtype=np.float64)
    R = px.shape[0]
    if res is None:
        res = HostResult(table.n_ifcs, R, opts.out_mode, want_pupil=True)
    out = res.out_struct()
    rc = lib().rox_oracle_trace_pupil_list(table.rows, table.n_ifcs,
                                           table.n_table.ctypes.data,
                                           _wvls(table).ctypes.data,
                                           len(table.wvls), C.byref(fld), R,
                                           px.ctypes.data, py.ctypes.data,
                                           wvl_idx, C.byref(opts), C.byref(out))
    if rc:
        raise RuntimeError(f'oracle error {rc}')
    return res


def make_grid(start, stop, num, kind=abi.GRID_PRODUCT, row_begin=0, row_count=0):
    g = abi.Grid()
    g.start[0], g.start[1] = start
    g.stop[0], g.stop[1] = stop
    g.num, g.kind = num, kind
    g.row_begin, g.row_count = row_begin, row_count
    return g


def aim_chief_rays(table, probs, eps=1.0e-12):
    """probs: sequence of abi.Aim -> (aim float64[n, 2] = (x1, y1), result int32[n])"""
    n = len(probs)
    arr = (abi.Aim * n)(*probs)
    aim = np.zeros((n, 2))
    result = np.zeros(n, dtype=np.int32)
    rc = lib().rox_oracle_aim_chief_rays(table.rows, table.n_ifcs, table.n_table.ctypes.data,
                                         _wvls(table).ctypes.data, len(table.wvls), n, arr,
                                         eps, aim.ctypes.data, result.ctypes.data)
    if rc:
        raise RuntimeError(f'oracle error {rc}')
    return aim, result


def iterate_pupil_rays(table, probs, eps=1.0e-12):
    """vigcalc.iterate_pupil_ray per problem (abi.PupilIter) -> start_r float64[n]"""
    n = len(probs)
    arr = (abi.PupilIter * n)(*probs)
    out = np.zeros(n)
    rc = lib().rox_oracle_iterate_pupil_rays(table.rows, table.n_ifcs, table.n_table.ctypes.data,
                                             _wvls(table).ctypes.data, len(table.wvls), n, arr,
                                             eps, out.ctypes.data)
    if rc:
        raise RuntimeError(f'oracle error {rc}')
    return out


def iterate_ray_raw(table, probs, eps=1.0e-12):
    """trace.iterate_ray_raw over the path `table` describes: (aim [n, 2], result [n],
    last_xy [n, 2] = pupil-plane coordinates of the last trial ray, last_status [n])"""
    n = len(probs)
    arr = (abi.Aim * n)(*probs)
    aim = np.zeros((n, 2))
    result = np.zeros(n, dtype=np.int32)
    last_xy = np.zeros((n, 2))
    last_st = np.zeros(n, dtype=np.int32)
    rc = lib().rox_oracle_iterate_ray_raw(table.rows, table.n_ifcs, table.n_table.ctypes.data,
                                          _wvls(table).ctypes.data, len(table.wvls), n, arr,
                                          eps, aim.ctypes.data, result.ctypes.data,
                                          last_xy.ctypes.data, last_st.ctypes.data)
    if rc:
        raise RuntimeError(f'oracle error {rc}')
    return aim, result, last_xy, last_st


def find_real_enp(table, probs, eps=1.0e-12):
    """probs: sequence of abi.Enp -> (z float64[n, 2] = (z_enp, z of the last trial ray),
    result int32[n] = abi.ENP_*): wideangle.find_real_enp restated (rox_oracle.c)"""
    n = len(probs)
    arr = (abi.Enp * n)(*probs)
    z = np.zeros((n, 2))
    result = np.zeros(n, dtype=np.int32)
    rc = lib().rox_oracle_find_real_enp(table.rows, table.n_ifcs, table.n_table.ctypes.data,
                                        _wvls(table).ctypes.data, len(table.wvls), n, arr,
                                        C.c_double(eps), z.ctypes.data, result.ctypes.data)
    if rc:
        raise RuntimeError(f'oracle error {rc}')
    return z, result


SCALAR_FN = C.CFUNCTYPE(C.c_double, C.c_double, C.c_void_p)


def brentq(f, a, b, xtol=2e-12, rtol=8.881784197001252e-16, maxiter=100):
    """scipy.optimize.brentq's core as restated in rox_oracle.c, around a Python callable:
    (root, funcalls, iterations, err) -- the pin against scipy itself"""
    L = lib()
    L.rox_oracle_brentq.restype = C.c_double
    L.rox_oracle_brentq.argtypes = [SCALAR_FN, C.c_void_p, C.c_double, C.c_double, C.c_double,
                                    C.c_double, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int),
                                    C.POINTER(C.c_int)]
    fc, it, err = C.c_int(), C.c_int(), C.c_int()
    cb = SCALAR_FN(lambda x, _ctx: float(f(x)))
    root = L.rox_oracle_brentq(cb, None, a, b, xtol, rtol, maxiter, C.byref(fc), C.byref(it),
                               C.byref(err))
    return root, fc.value, it.value, err.value


def secant(f, x0, tol=1.48e-8, rtol=0.0, maxiter=50):
    """scipy.optimize.newton's secant branch (disp=False) as restated in rox_oracle.c:
    (root, converged, funcalls)"""
    L = lib()
    L.rox_oracle_secant.restype = C.c_double
    L.rox_oracle_secant.argtypes = [SCALAR_FN, C.c_void_p, C.c_double, C.c_double, C.c_double,
                                    C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    conv, fc = C.c_int(), C.c_int()
    cb = SCALAR_FN(lambda x, _ctx: float(f(x)))
    root = L.rox_oracle_secant(cb, None, x0, tol, rtol, maxiter, C.byref(conv), C.byref(fc))
    return root, bool(conv.value), fc.value


def calc_vignetting(table, probs, eps=1.0e-12):
    """probs: sequence of abi.Vig -> (vig float64[n], clip_surf int32[n])"""
    n = len(probs)
    arr = (abi.Vig * n)(*probs)
    vig = np.zeros(n)
    clip = np.zeros(n, dtype=np.int32)
    rc = lib().rox_oracle_calc_vignetting(table.rows, table.n_ifcs, table.n_table.ctypes.data,
                                          _wvls(table).ctypes.data, len(table.wvls), n, arr,
                                          eps, vig.ctypes.data, clip.ctypes.data)
    if rc:
        raise RuntimeError(f'oracle error {rc}')
    return vig, clip


def calc_psf(opd, ndim, maxdim):
    """rox_oracle_calc_psf: analyses.calc_psf restated (analyses.py:848-875)"""
    w = np.ascontiguousarray(opd, dtype=np.float64)
    assert w.shape == (ndim, ndim)
    out = np.empty((maxdim, maxdim))
    f = lib().rox_oracle_calc_psf
    f.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    f.restype = C.c_int
    rc = f(w.ctypes.data, int(ndim), int(maxdim), out.ctypes.data)
    if rc:
        raise ValueError(f'oracle calc_psf: shapes rejected ({rc})')
    return out


HYBRD_FCN = C.CFUNCTYPE(C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_void_p)


def hybrd(func, x0, xtol=1.49012e-8, maxfev=0, epsfcn=None, factor=100.0):
    """oracle/minpack_hybrd.c through a Python callback: what scipy.optimize.fsolve(func, x0,
    full_output=True) runs (minus fsolve's own extra func(x0) call).  func may raise
    StopIteration to abort (MINPACK's iflag < 0).  Returns (x, infodict, info)."""
    x = np.array(x0, dtype=np.float64).ravel().copy()
    n = len(x)
    if maxfev == 0:
        maxfev = 200 * (n + 1)
    if epsfcn is None:
        epsfcn = np.finfo(np.float64).eps

    def cb(n_, xp, fp, _ctx):
        try:
            f = np.asarray(func(np.array([xp[i] for i in range(n_)])), dtype=np.float64).ravel()
        except StopIteration:
            return -1
        for i in range(n_):
            fp[i] = f[i]
        return 0
    fvec = np.zeros(n)
    fjac = np.zeros(n * n)
    r = np.zeros(n * (n + 1) // 2)
    qtf = np.zeros(n)
    nfev = C.c_int(0)
    f = lib().rox_oracle_hybrd
    f.restype = C.c_int
    f.argtypes = [HYBRD_FCN, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_double, C.c_int,
                  C.c_double, C.c_double, C.POINTER(C.c_int), C.c_void_p, C.c_void_p, C.c_void_p]
    info = f(HYBRD_FCN(cb), None, n, x.ctypes.data, fvec.ctypes.data, float(xtol), int(maxfev),
             float(epsfcn), float(factor), C.byref(nfev), fjac.ctypes.data, r.ctypes.data,
             qtf.ctypes.data)
    # scipy hands fjac back as the C-ordered view of MINPACK's column-major array
    return x, dict(nfev=nfev.value, fvec=fvec, fjac=fjac.reshape(n, n), r=r, qtf=qtf), info
