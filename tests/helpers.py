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


# ---- phase elements at their limits: constructions that REACH ROX_EVANESCENT and a TIR inside
# DiffractiveElement.phase (raytrace.py:41-48, 253-257; doe.py:296-297, 322-323), and the
# grating's np.sqrt NaN (doe.py:150: no exception -- the ray goes on as NaN with status OK).
# tests/test_oracle_phase_reference.py pins the same constructions against the live reference.
PHASE_LIMIT_WVLS = (486.1, 587.6, 656.3)
PHASE_LIMIT_CASES = {
    # kind: (object-space index per wavelength, cone half angle in degrees, seed)
    'grating_nan': ([1.0, 1.0, 1.0], 40., 1),
    'doe_tir': ([1.70, 1.71, 1.72], 50., 2),
    'doe_evanescent': ([1.0, 1.0, 1.0], 30., 3),
    'hologram_evanescent': ([1.0, 1.0, 1.0], 60., 4),
}


def phase_limit_params(kind):
    """the phase element of a limit case as plain numbers (shared with the reference-side builder)"""
    return {
        'grating_nan': dict(order=1, lpmm=1500., normal=(0., 1., 0.)),
        'doe_tir': dict(coefs=[-1e-3, 1e-6], ref_wl=587.6, order=1),
        'doe_evanescent': dict(coefs=[0.08], ref_wl=587.6, order=1),
        'hologram_evanescent': dict(ref_pt=(0., -40., -30.), obj_pt=(0., 40., 30.), ref_wl=450.0),
    }[kind]


def phase_limit_case(kind):
    """(table, pt0, dir0, wvl_idx): object surface, one flat phase-element surface 5 mm behind
    it, image surface; a cone of 4096 rays from points near the axis"""
    n_obj, half_angle, seed = PHASE_LIMIT_CASES[kind]
    surfs = [dict(cv=0., thi=5., n=list(n_obj), max_aperture=1e12),
             dict(cv=0., thi=5., n=1.0, max_aperture=1e3),
             dict(cv=0., thi=0., n=1.0, max_aperture=1e12)]
    tbl = SurfaceTable.from_prescription(surfs, wvls=PHASE_LIMIT_WVLS)
    ph, p = tbl.rows[1].ph, phase_limit_params(kind)
    if kind == 'grating_nan':
        ph.kind, ph.order, ph.spacing_nm = abi.PH_GRATING, float(p['order']), 1e6 / p['lpmm']
        for i in range(3):
            ph.a[i] = p['normal'][i]
    elif kind.startswith('doe'):
        ph.kind, ph.ncoef = abi.PH_DOE_RADIAL, len(p['coefs'])
        for k, c in enumerate(p['coefs']):
            ph.coefs[k] = c
        ph.ref_wl, ph.order = p['ref_wl'], float(p['order'])
    else:
        ph.kind, ph.flags, ph.ref_wl = abi.PH_HOLOGRAM, 0, p['ref_wl']
        for i in range(3):
            ph.a[i], ph.b[i] = p['ref_pt'][i], p['obj_pt'][i]
    rng = np.random.default_rng(seed)
    R = 4096
    pt0 = np.stack([rng.uniform(-1, 1, R), rng.uniform(-1, 1, R), np.zeros(R)])
    th, az = np.deg2rad(rng.uniform(0, half_angle, R)), rng.uniform(0, 2 * np.pi, R)
    d = np.stack([np.sin(th) * np.cos(az), np.sin(th) * np.sin(az), np.cos(th)])
    wi = (np.arange(R) % len(PHASE_LIMIT_WVLS)).astype(np.int32)
    return tbl, pt0, d, wi


def phase_limit_expect(kind, status, seg):
    """what a limit case must reach: (n_ok, n_limit) with the limit counted as the case names it"""
    n_ok = int((status == abi.OK).sum())
    if kind == 'grating_nan':
        nan_ok = np.isnan(seg).any(axis=(0, 1)) & (status == abi.OK)
        return n_ok, int(nan_ok.sum())
    want = abi.TIR if kind == 'doe_tir' else abi.EVANESCENT
    return n_ok, int((status == want).sum())


def record(kind, **kw):
    """append a measurement a test made to the file ROX_TEST_RECORDS names (tools/gpu_run.sh sets
    it to gpurun_out/<tag>/test_records.jsonl): deviations and flip counts worth keeping beside
    a green run (profiles/)"""
    path = os.environ.get('ROX_TEST_RECORDS')
    if not path:
        return
    with open(path, 'a') as f:
        f.write(json.dumps(dict(kind=kind, **kw)) + '\n')


# ---- tolerance mode (ROX_FAST_FP64): comparison and status-flip accounting -------------------
def scaled_err(ref, got):
    """max |ref - got| / max(1, |ref|) over the entries finite in both; NaN patterns must agree"""
    ref, got = np.asarray(ref, dtype=np.float64), np.asarray(got, dtype=np.float64)
    assert ref.shape == got.shape, (ref.shape, got.shape)
    rn, gn = np.isnan(ref), np.isnan(got)
    assert np.array_equal(rn, gn), f'NaN pattern differs at {np.argwhere(rn != gn)[:5].tolist()}'
    m = ~rn & np.isfinite(ref) & np.isfinite(got)
    if not m.any():
        return 0.0
    return float((np.abs(ref[m] - got[m]) / np.maximum(1.0, np.abs(ref[m]))).max())


def _rot(row):
    return np.array(list(row.rt), dtype=np.float64).reshape(3, 3)


def boundary_margin(tbl, wi, opts, seg, surf):
    """How close the ray whose EXACT packet is `seg` ([n_seg, 10], NaN where not produced) comes
    to a decision boundary at interface `surf`: the smallest of
      * the aperture margin   |r - (radius + fuzz)| / max(1, r)      (interface.py:113-122,
        surface.py:198-208; rectangles: |x| - (half width + fuzz) likewise),
      * the TIR margin        |n_out^2 - n_in^2 sin^2 I| / n_out^2   (raytrace.py:19-30),
      * the miss margin       |b^2 - ax2 cx2| / max(b^2, |ax2 cx2|, 1e-300) of a Spherical / Conic
        (profiles.py:321-336, 580-593).
    Only phantom-free packets (segment k = interface k)."""
    rows = tbl.rows
    row = rows[surf]
    margins = {}
    p_prev, d_prev = seg[surf - 1, 0:3], seg[surf - 1, 3:6]
    have_inc = not np.isnan(seg[surf, 0])
    # -- miss margin: needs only the previous segment
    if row.profile in (abi.SPHERICAL, abi.CONIC) and not np.isnan(p_prev).any():
        prow = rows[surf - 1]
        rt = _rot(prow)
        b4p = rt @ (p_prev - np.array(list(prow.t)))
        b4d = rt @ d_prev
        pp = b4p - np.dot(b4p, b4d) * b4d
        cv = row.cv
        if row.profile == abi.SPHERICAL:
            ax2, cx2, b = cv, cv * np.dot(pp, pp) - 2 * pp[2], cv * np.dot(b4d, pp) - b4d[2]
        else:
            ax2 = cv * (1. + row.cc * b4d[2] ** 2)
            cx2 = cv * (pp[0] ** 2 + pp[1] ** 2 + row.ec * pp[2] ** 2) - 2.0 * pp[2]
            b = cv * (b4d[0] * pp[0] + b4d[1] * pp[1] + row.ec * b4d[2] * pp[2]) - b4d[2]
        margins['miss'] = abs(b * b - ax2 * cx2) / max(b * b, abs(ax2 * cx2), 1e-300)
    if have_inc:
        x, y = seg[surf, 0], seg[surf, 1]
        fuzz = opts.fuzz
        # -- aperture margin
        m_ap = []
        if row.n_ap > 0:
            for k in range(row.n_ap):
                ap = row.ap[k]
                xx, yy = x - ap.x_offset, y - ap.y_offset
                if ap.kind == abi.AP_CIRCULAR:
                    r = np.hypot(xx, yy)
                    m_ap.append(abs(r - (ap.a + fuzz)) / max(1.0, r))
                elif ap.kind == abi.AP_RECTANGULAR:
                    m_ap.append(abs(abs(xx) - (ap.a + fuzz)) / max(1.0, abs(xx)))
                    m_ap.append(abs(abs(yy) - (ap.b + fuzz)) / max(1.0, abs(yy)))
        else:
            r = np.hypot(x, y)
            m_ap.append(abs(r - (row.max_aperture + fuzz)) / max(1.0, r))
        if m_ap:
            margins['aperture'] = min(m_ap)
        # -- TIR margin: the direction before the interface in ITS frame, the unit normal there
        if row.mode == abi.TRANSMIT and not np.isnan(seg[surf, 7]):
            prow = rows[surf - 1]
            b4d = _rot(prow) @ d_prev
            n = seg[surf, 7:10]
            cosI = np.dot(b4d, n) / np.linalg.norm(n)
            n_in, n_out = tbl.n_table[wi][surf - 1], tbl.n_table[wi][surf]
            margins['tir'] = abs(n_out * n_out - n_in * n_in * (1.0 - cosI * cosI)) / (n_out * n_out)
    return margins
