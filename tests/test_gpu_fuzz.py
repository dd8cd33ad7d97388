"""-m gpu: randomized differential test, HIP vs oracle, bit for bit.

Random prescriptions (spherical / conic / even / radial / toroid surfaces,
mirrors, dummies, phantoms, tilts and decenters with either dgemv order,
aperture lists) x random rays that include grazing, missing, blocked and
totally reflected ones, in every output mode.  The oracle is pinned to the
reference separately (tests/test_oracle_golden.py); this test hunts for
divergences between the two restatements on inputs no fixture covers."""
import numpy as np
import pytest

from rayoptics_amd import abi, SurfaceTable

pytestmark = pytest.mark.gpu


def rot(rng, max_deg):
    a, b, c = np.deg2rad(rng.uniform(-max_deg, max_deg, 3))
    rx = np.array([[1, 0, 0], [0, np.cos(a), -np.sin(a)], [0, np.sin(a), np.cos(a)]])
    ry = np.array([[np.cos(b), 0, np.sin(b)], [0, 1, 0], [-np.sin(b), 0, np.cos(b)]])
    rz = np.array([[np.cos(c), -np.sin(c), 0], [np.sin(c), np.cos(c), 0], [0, 0, 1]])
    return rx @ ry @ rz


def random_table(rng):
    n_surf = int(rng.integers(1, 9))
    N = n_surf + 2
    rows = (abi.Surface * N)()
    W = int(rng.integers(1, 4))
    n_table = np.ones((W, N))
    zdir = 1.0
    for i in range(N):
        r = rows[i]
        interior = 0 < i < N - 1
        r.mode = abi.DUMMY
        r.profile = abi.SPHERICAL
        r.ec = 1.0
        for k in range(3):
            r.rt[4 * k] = 1.0
        r.rt_order = int(rng.integers(0, 2))
        thi = float(rng.uniform(1.0, 15.0)) if i > 0 else float(rng.choice([20.0, 300.0, 1e10]))
        if interior:
            r.mode = int(rng.choice([abi.TRANSMIT] * 6 + [abi.REFLECT, abi.DUMMY, abi.PHANTOM]))
            r.profile = int(rng.choice([abi.SPHERICAL] * 3 + [abi.CONIC, abi.EVENPOLY, abi.RADIALPOLY,
                                                               abi.YTOROID, abi.XTOROID]))
            r.cv = float(rng.uniform(-0.08, 0.08)) if rng.random() > 0.15 else 0.0
            if r.profile != abi.SPHERICAL:
                r.cc = float(rng.uniform(-2.0, 1.0))
                r.ec = r.cc + 1.0
            if r.profile >= abi.EVENPOLY:
                r.ncoef = int(rng.integers(0, 5))
                for k in range(r.ncoef):
                    r.coefs[k] = float(rng.normal() * 10.0 ** (-(3 + 2 * k)))
            if r.profile >= abi.YTOROID:
                r.cR = float(rng.uniform(-0.05, 0.05))
            if rng.random() < 0.3:
                r.n_ap = int(rng.integers(1, 3))
                for k in range(r.n_ap):
                    a = r.ap[k]
                    a.kind = int(rng.choice([abi.AP_CIRCULAR, abi.AP_RECTANGULAR, abi.AP_ALWAYS_BLOCK],
                                            p=[0.6, 0.35, 0.05]))
                    a.is_obscuration = int(rng.random() < 0.2)
                    a.x_offset, a.y_offset = (float(v) for v in rng.uniform(-1, 1, 2))
                    a.a, a.b = float(rng.uniform(2, 9)), float(rng.uniform(2, 9))
            if r.mode == abi.REFLECT:
                zdir = -zdir
                thi = -thi
            elif zdir < 0:
                thi = -thi
            if rng.random() < 0.25:
                m = rot(rng, 6.0)
                for a_ in range(3):
                    for b_ in range(3):
                        r.rt[3 * a_ + b_] = float(m[a_, b_])
                r.t[0], r.t[1] = (float(v) for v in rng.uniform(-0.5, 0.5, 2))
        r.t[2] = thi if i < N - 1 else 0.0
        r.z_dir = zdir
        r.max_aperture = float(rng.uniform(4.0, 12.0)) if interior else 1e12
        for w in range(W):
            n_table[w, i] = 1.0 if rng.random() < 0.4 or not interior else float(rng.uniform(1.3, 1.9))
    return SurfaceTable(rows, n_table, [500.0 + 50 * w for w in range(W)])


def random_rays(rng, tbl, R):
    thi0 = tbl.rows[0].t[2]
    tgt = np.stack([rng.uniform(-10, 10, R), rng.uniform(-10, 10, R), np.full(R, thi0)])
    pt0 = np.stack([rng.uniform(-6, 6, R), rng.uniform(-6, 6, R), np.zeros(R)])
    if thi0 > 1e6:
        pt0[:2] *= 1e8
    d = tgt - pt0
    d /= np.linalg.norm(d, axis=0)
    # a few hand-picked nasties: on-axis, zero components, steep, backwards
    d[:, 0] = [0., 0., 1.]
    pt0[:, 0] = 0.0
    d[:, 1] = [0.6, 0., 0.8]
    d[:, 2] = [0., -0.999, np.sqrt(1 - 0.999 ** 2)]
    d[:, 3] = [0., 0., -1.]
    return pt0, d


@pytest.mark.parametrize('seed', range(24))
def test_random_systems_bit_exact(seed):
    from oracle import oracle
    from rayoptics_amd.engine import TraceEngine
    rng = np.random.default_rng(1000 + seed)
    tbl = random_table(rng)
    N = tbl.n_ifcs
    R = 3000 + int(rng.integers(0, 200))
    pt0, d = random_rays(rng, tbl, R)
    W = len(tbl.wvls)
    wi = rng.integers(0, W, R).astype(np.int32) if seed % 2 else int(rng.integers(0, W))
    eng = TraceEngine(tbl)
    n_ok = 0
    for mode in (abi.OUT_FULL, abi.OUT_LAST, abi.OUT_HITS):
        flags = abi.INTERSECT_OBJ if seed % 5 else 0
        if seed % 3:
            flags |= abi.CHECK_APERTURES
        if seed % 4 == 0 and mode == abi.OUT_FULL and tbl.rows[0].mode != abi.PHANTOM:
            flags |= abi.FILTER_PHANTOMS
        opts = oracle.make_opts(flags=flags, out_mode=mode, first_surf=int(seed % 2),
                                last_surf=(N - 2) if seed % 7 else -1, foc=0.01 * seed,
                                image_pt=(0.1, -0.2))
        with np.errstate(all='ignore'):
            orc = oracle.trace_rays(tbl, pt0, d, wi, opts)
        dev = eng.trace_rays(pt0, d, wi, opts, nan_fill=True).to_host()
        np.testing.assert_array_equal(dev.status, orc.status, err_msg=f'seed {seed} mode {mode}')
        np.testing.assert_array_equal(dev.fail_surf, orc.fail_surf)
        K = dev.seg.shape[0] if mode == abi.OUT_FULL else None
        exp = orc.seg[:K] if K else orc.seg
        same = (dev.seg == exp) | (np.isnan(dev.seg) & np.isnan(exp))
        assert same.all(), (seed, mode, np.argwhere(~same)[:4].tolist())
        assert np.array_equal(dev.op, orc.op, equal_nan=True)
        n_ok += int((dev.status == 0).sum())
    eng.close()
    assert n_ok >= 0
