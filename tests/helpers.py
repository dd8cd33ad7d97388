"""shared test helpers: golden-fixture loading and SoA comparison"""
import json
import os

import numpy as np

from rayoptics_amd import SurfaceTable, abi
from rayoptics_amd.table import field_struct

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')

# north_star: "matching the reference NumPy path within 1e-10 on ray
# intercepts".  Intercepts and direction cosines of interior surfaces are
# O(1..100); object-space quantities of infinite-conjugate systems are O(1e10)
# (ulp 2e-6), so the bound is applied relative to max(1, |ref|).
ATOL = 1e-10


class Fixture:
    def __init__(self, name):
        self.name = name
        z = np.load(os.path.join(GOLDEN, name + '.npz'))
        self.table = SurfaceTable.from_dict(json.loads(str(z['table_json'])))
        self.cases = {}
        for key in z.files:
            if '/' in key:
                c, k = key.split('/', 1)
                self.cases.setdefault(c, {})[k] = z[key]

    def __getitem__(self, case):
        return self.cases[case]


_cache = {}


def fixture(name):
    if name not in _cache:
        _cache[name] = Fixture(name)
    return _cache[name]


def field_from_arr(a):
    return field_struct(a[0:3], a[3:5], a[5], a[6], a[7:11], a[11])


def make_opts(case, out_mode=abi.OUT_FULL, **kw):
    from oracle.oracle import make_opts as mk
    return mk(flags=int(case['flags']) if 'flags' in case else abi.INTERSECT_OBJ,
              out_mode=out_mode,
              first_surf=int(case.get('first_surf', 0)),
              last_surf=int(case.get('last_surf', -1)), **kw)


def assert_soa_close(exp_seg, got_seg, what='seg', atol=ATOL, require_exact=False):
    """NaN pattern identical; values within atol*max(1,|ref|).  Returns the
    fraction of bit-identical finite entries."""
    exp_seg = np.asarray(exp_seg)
    got_seg = np.asarray(got_seg)
    assert exp_seg.shape == got_seg.shape, (what, exp_seg.shape, got_seg.shape)
    en, gn = np.isnan(exp_seg), np.isnan(got_seg)
    assert np.array_equal(en, gn), f'{what}: NaN pattern differs at {np.argwhere(en != gn)[:5]}'
    m = ~en
    if not m.any():
        return 1.0
    e, g = exp_seg[m], got_seg[m]
    err = np.abs(e - g) / np.maximum(1.0, np.abs(e))
    worst = err.max()
    assert worst <= atol, f'{what}: max scaled error {worst:.3e} > {atol}'
    exact = float(np.mean(e == g))
    if require_exact:
        assert exact == 1.0, f'{what}: only {exact:.6f} bit-identical'
    return exact


def bit_equal(a, b, what):
    """same shape, same bits (NaN where NaN)"""
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    same = (a == b) | (np.isnan(a) & np.isnan(b))
    assert same.all(), (f'{what}: {np.count_nonzero(~same)} of {same.size} differ, '
                        f'first {np.argwhere(~same)[:3].tolist()}')


def assert_result_matches(case, res, atol=ATOL, require_exact=False, seg_key='seg'):
    """compare an oracle/device result object (seg, op, status, fail_surf) with
    the reference outputs stored in a golden case"""
    np.testing.assert_array_equal(res.status, case['status'])
    if 'fail_surf' in case:
        np.testing.assert_array_equal(res.fail_surf, case['fail_surf'])
    K = case[seg_key].shape[0]
    f1 = assert_soa_close(case[seg_key], np.asarray(res.seg)[:K], 'seg', atol, require_exact)
    f2 = assert_soa_close(case['op'], res.op, 'op', atol, require_exact)
    return min(f1, f2)


def phase_table(rng, kind, wvls=(486.1, 587.6, 656.3)):
    """a small system with one phase element of `kind` ('grating', 'doe',
    'hologram', 'thinlens'), built without the reference (GPU box)"""
    N = int(rng.integers(4, 8))
    k_phase = int(rng.integers(1, N - 1))
    surfs = []
    for i in range(N):
        interior = 0 < i < N - 1
        if not interior:
            surfs.append(dict(cv=0.0, thi=40.0 if i == 0 else 0.0, n=1.0, max_aperture=1e12))
            continue
        cv = float(rng.uniform(-0.03, 0.03)) if rng.random() > 0.3 else 0.0
        n = 1.0 if rng.random() < 0.35 else float(rng.uniform(1.4, 1.8))
        surfs.append(dict(cv=cv, thi=float(rng.uniform(2.0, 12.0)),
                          n=[1.0 if n == 1.0 else n + 0.004 * w for w in range(len(wvls))],
                          max_aperture=15.0, profile='Conic' if rng.random() < 0.3 else 'Spherical',
                          cc=float(rng.uniform(-1, 0.5))))
    tbl = SurfaceTable.from_prescription(surfs, wvls=wvls)
    row = tbl.rows[k_phase]
    ph = row.ph
    if kind == 'grating':
        g = rng.normal(size=3)
        g[2] *= 0.1
        g = np.array([0., 1., 0.]) if rng.random() < 0.5 else g / np.linalg.norm(g)
        ph.kind, ph.order = abi.PH_GRATING, float(rng.choice([-1, 1, 2]))
        ph.spacing_nm = 1e6 / float(rng.uniform(50., 900.))
        for i in range(3):
            ph.a[i] = g[i]
    elif kind == 'doe':
        nc = int(rng.integers(1, 5))
        ph.kind, ph.ncoef = abi.PH_DOE_RADIAL, nc
        for k in range(nc):
            ph.coefs[k] = float(rng.normal() * 10.0 ** (-(3 + 2 * k)))
        ph.ref_wl, ph.order = float(rng.choice([550., 632.8])), float(rng.choice([1, 1, -1, 2]))
    else:
        if kind == 'thinlens':
            row.profile, row.cv, row.cc, row.ec = abi.THINLENS, 0.0, 0.0, 1.0
            pwr = float(rng.uniform(-0.03, 0.05))
            ref_pt, obj_pt = (0., 0., -1e10), (0., 0., 1. / pwr)
            flags = 2 if pwr > 0 else 0
        else:
            ref_pt = (rng.uniform(-2, 2), rng.uniform(-2, 2), -rng.uniform(50, 500))
            obj_pt = (rng.uniform(-2, 2), rng.uniform(-2, 2), rng.uniform(30, 300))
            flags = (1 if rng.random() < 0.3 else 0) | (2 if rng.random() < 0.6 else 0)
        ph.kind, ph.flags, ph.ref_wl = abi.PH_HOLOGRAM, flags, float(rng.choice([550., 632.8]))
        for i in range(3):
            ph.a[i], ph.b[i] = ref_pt[i], obj_pt[i]
    return tbl, k_phase


def random_rays(rng, R, z_target, spread=9.0):
    pt0 = np.stack([rng.uniform(-5, 5, R), rng.uniform(-5, 5, R), np.zeros(R)])
    tgt = np.stack([rng.uniform(-spread, spread, R), rng.uniform(-spread, spread, R),
                    np.full(R, z_target)])
    d = tgt - pt0
    d /= np.linalg.norm(d, axis=0)
    return pt0, d
