"""-m gpu, round 6: tables beyond the LDS of a workgroup (the general instance over a table left
in global memory, read with scalar loads: rox_device.hpp F_GTAB) -- a SequentialModel has no
size limit (rayoptics/seq/sequential.py:167-202), round 5 returned ROX_E_UNSUPPORTED beyond
~220 interfaces -- and the same instance forced onto the regular fixtures."""
import os
import subprocess
import sys

import numpy as np
import pytest

from rayoptics_amd import SurfaceTable, abi
from rayoptics_amd.table import field_struct
import helpers as H

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SPOT = abi.INTERSECT_OBJ | abi.CHECK_APERTURES | abi.APPLY_VIGNETTING


def long_chain(n_lenses, rng):
    """object at infinity, n_lenses weak singlets (every fifth a conic, every seventh an even
    asphere, every ninth with a clear-aperture list, one mirror pair in the middle would fold
    the axis -- left out: the chain stays on one axis), image surface: 2 n_lenses + 2 interfaces"""
    surfs = [dict(cv=0.0, thi=1.0e10, n=1.0, max_aperture=1e12)]
    for k in range(n_lenses):
        front = dict(cv=1 / 103.0, thi=4.0, n=[1.5168, 1.5200, 1.5140], max_aperture=14.0)
        if k % 5 == 1:
            front.update(profile='Conic', cc=-0.6)
        if k % 7 == 2:
            front.update(profile='EvenPolynomial', cc=-0.2, coefs=[0.0, 1e-7, -2e-10])
        surfs.append(front)
        surfs.append(dict(cv=-1 / 103.0, thi=46.0, n=1.0, max_aperture=14.0))
    surfs.append(dict(cv=0.0, thi=0.0, n=1.0, max_aperture=50.0))
    tbl = SurfaceTable.from_prescription(surfs, wvls=(587.6, 486.1, 656.3), stop_idx=1)
    for k in range(3, n_lenses, 9):             # clear-aperture lists (a circle and an obscuration)
        row = tbl.rows[1 + 2 * k]
        row.n_ap = 2
        for ap, (obsc, ox, oy, rad) in zip(row.ap, ((0, 0.1, -0.05, 13.0), (1, 0.0, 0.2, 0.4))):
            ap.kind, ap.is_obscuration, ap.x_offset, ap.y_offset, ap.a, ap.b = abi.AP_CIRCULAR, obsc, ox, oy, rad, 0.0
    return tbl


@pytest.mark.parametrize('n_lenses', [150, 400])
def test_tables_beyond_the_lds(n_lenses):
    """302 and 802 interfaces (222 KB and 590 KB of rows: more than the 160 KiB of LDS): FULL
    packets, hits, LAST and packed hits of a pupil grid, explicit rays with per-ray wavelengths,
    a batch of (field, wavelength) grids -- every one bit-identical to the oracle"""
    from oracle import oracle
    from rayoptics_amd.engine import TraceEngine, make_opts, make_grid
    rng = np.random.default_rng(n_lenses)
    tbl = long_chain(n_lenses, rng)
    N = tbl.n_ifcs
    assert N == 2 * n_lenses + 2 and N * 736 > 160 * 1024
    eng = TraceEngine(tbl)
    flds = [field_struct([0.0, -1.0e10 * np.tan(np.deg2rad(t)), 0.0], (0., 0.), 9.0, 1.0e10)
            for t in (0.0, 0.05)]
    num = 40
    grid = make_grid((-1., -1.), (1., 1.), num)
    ogrid = oracle.make_grid((-1., -1.), (1., 1.), num)
    n_ok = 0
    for mode in (abi.OUT_FULL, abi.OUT_HITS, abi.OUT_LAST):
        opts = make_opts(flags=SPOT, out_mode=mode, first_surf=1, last_surf=N - 2, foc=0.01)
        dev = eng.trace_pupil_grid(flds[1], grid, 1, opts, nan_fill=True).to_host()
        orc = oracle.trace_pupil_grid(tbl, flds[1], ogrid, 1, opts)
        np.testing.assert_array_equal(dev.status, orc.status)
        np.testing.assert_array_equal(dev.fail_surf, orc.fail_surf)
        H.bit_equal(dev.seg, orc.seg, f'{N} interfaces mode {mode}')
        H.bit_equal(dev.op, orc.op, 'op')
        n_ok = int((orc.status == abi.OK).sum())
        assert 100 < n_ok < num * num
    oc = make_opts(flags=SPOT, out_mode=abi.OUT_HITS_COMPACT, first_surf=1, last_surf=N - 2, foc=0.01)
    xy = eng.trace_pupil_grid_hits(flds[1], grid, 1, oc)
    H.bit_equal(xy, oracle.trace_pupil_grid(tbl, flds[1], ogrid, 1, oc).hits, 'packed hits')
    # explicit rays, per-ray wavelengths
    R = 900
    pt0 = np.stack([rng.uniform(-8, 8, R), rng.uniform(-8, 8, R), np.full(R, -50.0)])
    d = np.stack([rng.uniform(-.01, .01, R), rng.uniform(-.01, .01, R), np.ones(R)])
    d /= np.linalg.norm(d, axis=0)
    wi = rng.integers(0, 3, R).astype(np.int32)
    opts = make_opts(flags=abi.CHECK_APERTURES, out_mode=abi.OUT_FULL, first_surf=1, last_surf=N - 2)
    dev = eng.trace_rays(pt0, d, wi, opts, nan_fill=True).to_host()
    orc = oracle.trace_rays(tbl, pt0, d, wi, opts)
    np.testing.assert_array_equal(dev.status, orc.status)
    H.bit_equal(dev.seg, orc.seg, 'explicit rays')
    H.bit_equal(dev.op, orc.op, 'explicit rays op')
    # the batched entry: two fields x three wavelengths in one launch
    pairs = [(fi, w) for fi in range(2) for w in range(3)]
    ol = [make_opts(flags=SPOT, out_mode=abi.OUT_HITS, first_surf=1, last_surf=N - 2, foc=0.01) for _ in pairs]
    res = eng.trace_pupil_grids([flds[fi] for fi, _ in pairs], [w for _, w in pairs], grid, ol, nan_fill=True)
    for (fi, w), o, r in zip(pairs, ol, res):
        orc = oracle.trace_pupil_grid(tbl, flds[fi], ogrid, w, o)
        b = r.to_host()
        np.testing.assert_array_equal(b.status, orc.status)
        H.bit_equal(b.seg, orc.seg, f'batched f{fi} w{w}')
    # the search kernels over the same table (their trial rays read it through scalar loads too):
    # chief-ray aiming at three wavelengths
    probs = []
    for w in range(3):
        pa = abi.Aim()
        pa.pt0[1] = -1.0e10 * np.tan(np.deg2rad(0.05))
        pa.z_enp, pa.y_target, pa.z_dir0, pa.wvl_idx, pa.surf, pa.flip = 1.0e10, 0.0, 1.0, w, 1, 1
        probs.append(pa)
    y_dev, r_dev = eng.aim_chief_rays(probs)
    y_orc, r_orc = oracle.aim_chief_rays(tbl, probs, 1e-12)
    assert np.array_equal(r_dev, r_orc) and np.array_equal(y_dev, y_orc)
    eng.close()


def test_the_global_table_instance_on_the_regular_fixtures():
    """ROX_FORCE_GTAB=1 (read once by the library: a subprocess) sends every trace launch through
    the general instance over the global table (the search kernels included): the parity,
    configuration, batch, aiming / vignetting / wide-angle tests -- every fixture, every output
    mode, phase elements, phantom filtering -- pass unchanged"""
    env = dict(os.environ, ROX_FORCE_GTAB='1')
    files = ['tests/test_gpu_parity.py', 'tests/test_gpu_configs.py', 'tests/test_gpu_batch.py',
             'tests/test_gpu_r02.py', 'tests/test_gpu_r04.py']
    p = subprocess.run([sys.executable, '-m', 'pytest', '-x', '-q', '-m', 'gpu', '-p', 'no:cacheprovider'] + files,
                       env=env, cwd=ROOT, capture_output=True, text=True, timeout=1800)
    assert p.returncode == 0, p.stdout[-4000:] + p.stderr[-2000:]
    assert ' passed' in p.stdout


# ---------------------------------------------------------------- rox_spot_stats
@pytest.mark.parametrize('name,num', [('dblgauss_c2', 300), ('rc_telescope_c4', 128), ('cell_phone', 200)])
def test_spot_statistics_on_the_device(name, num):
    """rox_spot_stats over a ROX_OUT_HITS launch == NumPy over the ORACLE's spot of the same grid:
    the count, min / max (RayGeoPSF.ray_data_bounds, analysisfigure.py:237-248) exactly; the 2-D
    histogram == numpy.histogram2d(x, y, bins=[x_edges, y_edges]) count for count, for RayGeoPSF's
    own 'fit' edges (np.linspace over the data's half-extent, 100 samples) and for edges laid ON
    data values (the closed right edge, values outside the range); centroid and RMS radius to
    1e-12 (parallel sums).  The packed-pairs layout (ROX_OUT_HITS_COMPACT in HBM) gives the same."""
    from oracle import oracle
    from rayoptics_amd import workloads
    from rayoptics_amd.engine import TraceEngine, make_opts, make_grid, DeviceResult
    import torch
    wl = workloads.load(name)
    N = wl.n_ifcs
    eng = TraceEngine(wl.table)
    fi = len(wl.fields) - 1
    fld = wl.fields[fi]
    flags = SPOT if (fld.kind != abi.FLD_EPD_WIDE and fld.z_dir0 != 0.0) else SPOT & ~abi.INTERSECT_OBJ
    grid = make_grid((-1., -1.), (1., 1.), num)
    o = make_opts(flags=flags, out_mode=abi.OUT_HITS, first_surf=1, last_surf=N - 2, foc=wl.foc,
                  image_pt=wl.image_pts[fi])
    orc = oracle.trace_pupil_grid(wl.table, fld, oracle.make_grid((-1., -1.), (1., 1.), num), wl.ref_wvl_idx, o)
    ok = orc.status == abi.OK
    x, y = orc.seg[0][ok], orc.seg[1][ok]
    assert len(x) > 1000
    res = DeviceResult(torch, eng.device, 0, num * num, abi.OUT_HITS, want_pupil=False, nan_fill=False)
    eng.trace_pupil_grid(fld, grid, wl.ref_wvl_idx, o, want_pupil=False, out=res)
    summ, none = eng.spot_stats(res)
    assert none is None and summ['n'] == len(x)
    assert summ['min'] == (x.min(), y.min()) and summ['max'] == (x.max(), y.max())
    assert abs(summ['centroid'][0] - x.mean()) <= 1e-12 * max(1.0, abs(x.mean()))
    assert abs(summ['centroid'][1] - y.mean()) <= 1e-12 * max(1.0, abs(y.mean()))
    rms = np.sqrt(np.mean((x - x.mean()) ** 2 + (y - y.mean()) ** 2))
    assert abs(summ['rms_radius'] - rms) <= 1e-9 * max(rms, 1e-300) + 1e-15
    # RayGeoPSF's 'fit' edges (analysisfigure.py:250-262)
    dx, dy = (x.max() - x.min()) / 2, (y.max() - y.min()) / 2
    cy = (y.max() + y.min()) / 2
    mv = max(dx, dy)
    cases = [(np.linspace(-mv, mv, num=100), np.linspace(cy - mv, cy + mv, num=100)),
             (np.linspace(-mv, mv, num=257), np.linspace(cy - mv, cy + mv, num=257)),
             # edges ON data values, a range that drops part of the data, uneven bins
             (np.sort(np.concatenate([x[::997][:40], [x.max()]])), np.array([y.min(), np.median(y), y.max()])),
             (np.array([np.median(x), x.max()]), np.linspace(y.min(), np.median(y), 7))]
    for xe, ye in cases:
        xe, ye = np.unique(xe), np.unique(ye)
        want = np.histogram2d(x, y, bins=[xe, ye])[0]
        s2, hist = eng.spot_stats(res, xe, ye)
        assert s2['n'] == len(x)
        np.testing.assert_array_equal(hist.astype(np.float64), want)
        assert hist.sum() == want.sum() > 0
    # the packed pairs of a ROX_OUT_HITS_COMPACT launch, device-resident, counted on the device
    pack = eng.hits_pack(num * num, 1)
    oc = make_opts(flags=flags | abi.HITS_APPEND, out_mode=abi.OUT_HITS_COMPACT, first_surf=1, last_surf=N - 2,
                   foc=wl.foc, image_pt=wl.image_pts[fi])
    eng.trace_pupil_grid_hits_append(fld, grid, wl.ref_wvl_idx, oc, pack)
    xe, ye = cases[0]
    s3, h3 = eng._spot_stats(pack.seg_ptr, pack.cap, None, pack.count.data_ptr(), num * num, abi.SPOT_PAIRS, xe, ye)
    assert s3['n'] == len(x) and s3['min'] == summ['min'] and s3['max'] == summ['max']
    np.testing.assert_array_equal(h3.astype(np.float64), np.histogram2d(x, y, bins=[xe, ye])[0])
    eng.close()


def test_spot_stats_product_call_and_its_wall_clock():
    """rayoptics_amd.trace.trace_grid_spot_stats on a table-backed model: 1 048 576 rays -> summary +
    RayGeoPSF's 100 x 100 'fit' histogram and a 256 x 256 one on the host; == NumPy over the
    packed spot trace_grid_spot returns (itself the oracle's, bit for bit); the wall-clock of
    both calls recorded"""
    import time
    import torch
    from rayoptics_amd import workloads, trace as rox_trace
    wl = workloads.load('dblgauss_c2')
    model = workloads.TableModel(wl)
    fi = 0
    fld = model.fields[fi]
    num = 1024
    rng = [np.array([-1., -1.]), np.array([1., 1.]), num]
    wvl = wl.table.wvls[wl.ref_wvl_idx]
    xy = np.array(rox_trace.trace_grid_spot(model, rng, fld, wvl, wl.foc, wl.image_pts[fi]))
    x, y = xy[:, 0], xy[:, 1]
    for bins in (100, 257):
        summ, hist, xe, ye = rox_trace.trace_grid_spot_stats(model, rng, fld, wvl, wl.foc, wl.image_pts[fi], bins=bins)
        assert summ['n'] == len(x) and summ['min'] == (x.min(), y.min()) and summ['max'] == (x.max(), y.max())
        np.testing.assert_array_equal(hist.astype(np.float64), np.histogram2d(x, y, bins=[xe, ye])[0])
    t = {}
    for what, fn in (('spot_to_host_ms', lambda: rox_trace.trace_grid_spot(model, rng, fld, wvl, wl.foc, wl.image_pts[fi])),
                     ('stats_hist_257_ms', lambda: rox_trace.trace_grid_spot_stats(model, rng, fld, wvl, wl.foc,
                                                                                   wl.image_pts[fi], bins=257)),
                     ('stats_hist_given_edges_ms', lambda: rox_trace.trace_grid_spot_stats(
                         model, rng, fld, wvl, wl.foc, wl.image_pts[fi], bins=(xe, ye))),
                     ('stats_only_ms', lambda: rox_trace.trace_grid_spot_stats(model, rng, fld, wvl, wl.foc,
                                                                               wl.image_pts[fi]))):
        for _ in range(30):
            fn()
        ts = []
        for _ in range(21):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            fn()
            ts.append((time.perf_counter() - t0) * 1e3)
        t[what] = float(np.median(ts))
    # ... and with the tolerance-mode kernels behind the same calls (session.set_tolerance_mode)
    from rayoptics_amd import session
    was = session.set_tolerance_mode(True)
    try:
        for what, fn in (('tolerance_stats_only_ms', lambda: rox_trace.trace_grid_spot_stats(
                              model, rng, fld, wvl, wl.foc, wl.image_pts[fi])),
                         ('tolerance_stats_hist_given_edges_ms', lambda: rox_trace.trace_grid_spot_stats(
                              model, rng, fld, wvl, wl.foc, wl.image_pts[fi], bins=(xe, ye)))):
            for _ in range(30):
                fn()
            ts = []
            for _ in range(21):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                fn()
                ts.append((time.perf_counter() - t0) * 1e3)
            t[what] = float(np.median(ts))
        s_f, h_f, _, _ = rox_trace.trace_grid_spot_stats(model, rng, fld, wvl, wl.foc, wl.image_pts[fi], bins=(xe, ye))
    finally:
        session.set_tolerance_mode(was)
    # (the 'fit' edges are the exact spot's own extremes: a tolerance-mode hit that sat on one may
    # land a rounding outside it -- the rays ON the frame, a handful by symmetry -- and is dropped)
    assert s_f['n'] == len(x) and abs(int(h_f.sum()) - int(hist.sum())) <= 16
    H.record('spot_stats_wallclock_1M_rays', **t)
    assert t['stats_only_ms'] < t['spot_to_host_ms']


# ---------------------------------------------------------------- degenerate operands
@pytest.mark.parametrize('name', ['dblgauss', 'nikkor', 'rc_telescope', 'cell_phone', 'tilted_singlet', 'toroid_lens'])
def test_degenerate_rays_bit_exact(name):
    """rays whose start points and directions are drawn from {+-0, +-1, 0.5, 1e-17, 1e-300, 1e300, inf,
    nan} (the reference traces whatever it is given: NaN and infinities flow through its NumPy
    arithmetic, comparisons with NaN are false) -- FULL, LAST and HITS on the device against the
    oracle, bit for bit incl. the NaN pattern, status and failing surface.  The straight-line loop
    of the reduced-output modes (trace_ray_reduced) lets the arithmetic behind a raised failure
    flag run on: this is where it would show if that changed an outcome."""
    from oracle import oracle
    from rayoptics_amd.engine import TraceEngine
    fx = H.fixture(name)
    tbl = fx.table
    N = tbl.n_ifcs
    rng = np.random.default_rng(606 + len(name))
    pool = np.array([0.0, -0.0, 1.0, -1.0, 0.5, 1e-17, 1e-300, 1e300, np.inf, -np.inf, np.nan, 3.0, -0.25])
    R = 6000
    pt0 = pool[rng.integers(0, len(pool), (3, R))]
    d = pool[rng.integers(0, len(pool), (3, R))]
    # half of the rays: a regular ray with ONE degenerate component
    reg = rng.random(R) < 0.5
    base_p = np.stack([rng.uniform(-5, 5, R), rng.uniform(-5, 5, R), np.zeros(R)])
    base_d = np.stack([rng.uniform(-.2, .2, R), rng.uniform(-.2, .2, R), np.ones(R)])
    base_d /= np.linalg.norm(base_d, axis=0)
    which = rng.integers(0, 6, R)
    for k in range(3):
        pt0[k] = np.where(reg & (which != k), base_p[k], pt0[k])
        d[k] = np.where(reg & (which != 3 + k), base_d[k], d[k])
    W = len(tbl.wvls)
    wi = rng.integers(0, W, R).astype(np.int32)
    eng = TraceEngine(tbl)
    n_ok = 0
    for mode in (abi.OUT_FULL, abi.OUT_LAST, abi.OUT_HITS):
        for flags in (abi.INTERSECT_OBJ | abi.CHECK_APERTURES, 0):
            opts = oracle.make_opts(flags=flags, out_mode=mode, first_surf=1, last_surf=N - 2, foc=0.03,
                                    image_pt=(0.1, -0.2))
            with np.errstate(all='ignore'):
                orc = oracle.trace_rays(tbl, pt0, d, wi, opts)
            dev = eng.trace_rays(pt0, d, wi, opts, nan_fill=True).to_host()
            np.testing.assert_array_equal(dev.status, orc.status, err_msg=f'{name} mode {mode} flags {flags}')
            np.testing.assert_array_equal(dev.fail_surf, orc.fail_surf)
            K = dev.seg.shape[0] if mode == abi.OUT_FULL else None
            H.bit_equal(dev.seg, orc.seg[:K] if K else orc.seg, f'{name} mode {mode} flags {flags} seg')
            H.bit_equal(dev.op, orc.op, f'{name} mode {mode} op')
            n_ok += int((orc.status == abi.OK).sum())
    eng.close()
    assert n_ok > 100


# ---- batches whose items travel in the kernel argument (<= 16) and batches that are uploaded
@pytest.mark.parametrize('n_items', [1, 2, 15, 16, 17, 40])
@pytest.mark.parametrize('mode', [abi.OUT_FULL, abi.OUT_HITS, abi.OUT_HITS_COMPACT])
def test_batches_either_side_of_the_inline_limit(n_items, mode):
    """rox_trace_pupil_grids hands up to 16 items to the kernel inside its argument block and
    uploads larger batches to device memory (csrc/rox_device.hpp BatchArgs): item for item the
    same bits as the single-grid entry, on both sides of the limit and in a sequence that
    alternates between the two on one stream"""
    from rayoptics_amd import workloads
    from rayoptics_amd.engine import TraceEngine, make_grid, make_opts
    wl = workloads.load('dblgauss_c2')
    eng = TraceEngine(wl.table)
    N = wl.n_ifcs
    W = len(wl.table.wvls)
    base = [(fi, wi) for fi in range(len(wl.fields)) for wi in range(W)]
    pairs = [base[i % len(base)] for i in range(n_items)]
    grid = make_grid((-1., -1.), (1., 1.), 37)
    flags = abi.INTERSECT_OBJ | abi.CHECK_APERTURES | abi.APPLY_VIGNETTING
    opts = [make_opts(flags=flags, out_mode=mode, first_surf=1, last_surf=N - 2, foc=wl.foc,
                      image_pt=wl.image_pts[fi]) for fi, _ in pairs]
    flds = [wl.fields[fi] for fi, _ in pairs]
    wis = [wi for _, wi in pairs]
    single = {}
    for (fi, wi), o in zip(pairs, opts):
        if (fi, wi) not in single:
            if mode == abi.OUT_HITS_COMPACT:
                single[fi, wi] = eng.trace_pupil_grids_hits([wl.fields[fi]], [wi], grid, [o])[0].copy()
            else:
                single[fi, wi] = eng.trace_pupil_grid(wl.fields[fi], grid, wi, o, nan_fill=True).to_host()
    for rep in range(2):
        if mode == abi.OUT_HITS_COMPACT:
            got = eng.trace_pupil_grids_hits(flds, wis, grid, opts)
            for p, g in zip(pairs, got):
                assert np.array_equal(g, single[p]), (n_items, p, rep)
            # ... a small batch in between (the other path on the same stream context)
            other = eng.trace_pupil_grids_hits(flds[:3], wis[:3], grid, opts[:3])
            for p, g in zip(pairs[:3], other):
                assert np.array_equal(g, single[p])
        else:
            res = eng.trace_pupil_grids(flds, wis, grid, opts, nan_fill=True)
            for p, r in zip(pairs, res):
                dev = r.to_host()
                np.testing.assert_array_equal(dev.status, single[p].status)
                H.bit_equal(dev.seg, single[p].seg, f'{n_items} items, item {p}, pass {rep}')
                H.bit_equal(dev.op, single[p].op, 'op')
    eng.close()
