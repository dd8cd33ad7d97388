"""-m gpu: the PRODUCT entry points (rayoptics_amd.trace / analyses drop-ins)
running over the HIP engine, on table-backed models built from the golden
fixtures (the reference itself is not on the GPU box): results against the
reference's own outputs stored in tests/golden and against the oracle.
Plus BASELINE configs[4] at full size on one GPU."""
import numpy as np
import pytest

from rayoptics_amd import abi
import helpers as H

pytestmark = pytest.mark.gpu
FLAGS = abi.INTERSECT_OBJ | abi.CHECK_APERTURES | abi.APPLY_VIGNETTING


def model_of(fx, fields, image_pts, wf=None, bbox=None, foc=0.0):
    from rayoptics_amd import workloads
    wl = workloads.SimpleWorkload(fx.table, fields, image_pts, foc=foc)
    m = workloads.TableModel(wl)
    if wf is not None:
        m.fields[0].rox_wavefront = wf
    if bbox is not None:
        m.fields[0]._vig_bbox = bbox
    return m


@pytest.mark.parametrize('name', ['dblgauss', 'singlet', 'rc_telescope', 'nikkor'])
def test_trace_grid_spot_product_function(name):
    """trace.trace_grid_spot (what SequentialModel.trace_grid is rebound to for
    SpotDiagramFigure) == the data the reference's SpotDiagramFigure computed"""
    from rayoptics_amd import trace, session
    fx = H.fixture(name)
    c = fx['spot']
    num = int(c['num'])
    for key in [k for k in c if k.endswith('_hits')]:
        fi, wi = key.split('_')[0], int(key.split('_')[1][1:])
        m = model_of(fx, [H.field_from_arr(c[f'{fi}_field'])], [tuple(c[f'{fi}_image_pt'])],
                     foc=float(c['foc']))
        xy = trace.trace_grid_spot(m, [np.array([-1., -1.]), np.array([1., 1.]), num], m.fields[0],
                                   fx.table.wvls[wi], float(c['foc']), c[f'{fi}_image_pt'])
        assert xy.dtype == np.float64 and xy.ndim == 2 and xy.shape[1] == 2
        np.testing.assert_array_equal(xy, c[key])
        # the caller may keep the array: a later call must not overwrite it
        keep = xy.copy()
        xy2 = trace.trace_grid_spot(m, [np.array([-1., -1.]), np.array([1., 1.]), num], m.fields[0],
                                    fx.table.wvls[wi], float(c['foc']) + 0.01, c[f'{fi}_image_pt'])
        np.testing.assert_array_equal(xy, keep)
        assert xy2.shape == xy.shape and not np.array_equal(xy2, xy)
    session.clear()


@pytest.mark.parametrize('name', ['dblgauss', 'nikkor'])
def test_trace_grid_spots_every_wavelength_in_one_launch(name):
    """trace.trace_grid_spots (the per-wavelength loop of SequentialModel.trace_grid as one
    rox_trace_pupil_grids launch) == the reference's SpotDiagramFigure data per wavelength"""
    from rayoptics_amd import trace, session
    fx = H.fixture(name)
    c = fx['spot']
    num = int(c['num'])
    by_field = {}
    for key in [k for k in c if k.endswith('_hits')]:
        fi, wi = key.split('_')[0], int(key.split('_')[1][1:])
        by_field.setdefault(fi, []).append(wi)
    n_multi = 0
    for fi, wis in by_field.items():
        wis = sorted(wis)
        m = model_of(fx, [H.field_from_arr(c[f'{fi}_field'])], [tuple(c[f'{fi}_image_pt'])],
                     foc=float(c['foc']))
        got = trace.trace_grid_spots(m, [np.array([-1., -1.]), np.array([1., 1.]), num], m.fields[0],
                                     [fx.table.wvls[w] for w in wis], float(c['foc']), c[f'{fi}_image_pt'])
        assert len(got) == len(wis)
        for w, xy in zip(wis, got):
            np.testing.assert_array_equal(xy, c[f'{fi}_w{w}_hits'])
        n_multi += len(wis) > 1
    assert n_multi >= 1
    session.clear()


from test_oracle_golden import OPD_CASES  # noqa: E402


@pytest.mark.parametrize('name,case', OPD_CASES)
def test_eval_wavefront_product_function(name, case):
    """analyses.eval_wavefront / trace_wavefront + focus_wavefront (RayGrid's pair)
    on the device == the grid the reference's eval_wavefront returned"""
    from rayoptics_amd import analyses, session
    from rayoptics_amd.table import wavefront_from_array
    fx = H.fixture(name)
    c = fx[case]
    m = model_of(fx, [H.field_from_arr(c['field'])], [(0., 0.)],
                 wf=wavefront_from_array(c['wavefront']), bbox=(c['start'], c['stop']))
    m._units_per_nm = 1.0 / (float(c['convert_to_opd']) * fx.table.wvls[int(c['wvl_idx'])])
    wvl = fx.table.wvls[int(c['wvl_idx'])]
    num = int(c['num'])
    got = analyses.eval_wavefront(m, m.fields[0], wvl, 0.0, num_rays=num)
    exp = c['opd_grid']
    assert got.shape == exp.shape
    np.testing.assert_array_equal(got[:, :, :2], exp[:, :, :2])
    assert np.array_equal(np.isnan(got[:, :, 2]), np.isnan(exp[:, :, 2]))
    ok = ~np.isnan(exp[:, :, 2])
    assert np.abs(got[:, :, 2][ok] - exp[:, :, 2][ok]).max() <= 1e-10 * float(c['convert_to_opd'])
    # RayGrid: trace_wavefront hands a deferred grid to focus_wavefront
    grid_pkg = analyses.trace_wavefront(m, m.fields[0], wvl, 0.0, num_rays=num)
    got2 = analyses.focus_wavefront(m, grid_pkg, m.fields[0], wvl, 0.0)
    if m.fields[0].rox_wavefront.kind == abi.WF_FINITE:
        np.testing.assert_array_equal(got2, got)
    else:
        # infinite reference sphere: RayGrid's pre-calc + calc route associates the
        # final sum differently from eval_wavefront's full calc (waveabr.py:427-488 vs
        # :356-424) -- an ulp apart, and bit-identical to the oracle's split variant
        from oracle import oracle
        from test_oracle_golden import opd_opts
        o = opd_opts(c)
        o.wf.kind = abi.WF_INF_SPLIT
        grid = oracle.make_grid(c['start'], c['stop'], num)
        orc = oracle.trace_pupil_grid(fx.table, m.fields[0].rox_field, grid, int(c['wvl_idx']), o)
        exp2 = np.where(orc.status == 0, float(c['convert_to_opd']) * orc.seg[0], np.nan)
        np.testing.assert_array_equal(got2[:, :, 2].ravel(), exp2)
        np.testing.assert_allclose(got2[:, :, 2], got[:, :, 2], rtol=0, atol=1e-9, equal_nan=True)
    session.clear()


def test_trace_rays_soa_and_deferred_ray_list():
    """analyses.trace_rays_soa (array form of trace_list_of_rays) and the
    _DeferredRayList of trace_pupil_coords / focus_pupil_coords over the HIP engine"""
    from oracle import oracle
    from rayoptics_amd import analyses, session
    fx = H.fixture('dblgauss')
    cr = fx['rays_ap']
    c = fx['grid_f2']
    N = fx.table.n_ifcs
    m = model_of(fx, [H.field_from_arr(c['field'])], [(0.0, 18.0)])
    wv = [fx.table.wvls[int(i)] for i in np.atleast_1d(cr['wvl_idx'])]
    if len(wv) == 1:
        wv = wv[0]
    pk = analyses.trace_rays_soa(m, cr['pt0'], cr['dir0'], wv, check_apertures=True)
    np.testing.assert_array_equal(pk.status, cr['status'])
    np.testing.assert_array_equal(pk.fail_surf, cr['fail_surf'])
    # (device buffers are not pre-filled here: slots past a failure hold garbage
    # that the views never read -- compare what the reference appended)
    got = np.where(np.isnan(cr['seg']), np.nan, pk.seg[:cr['seg'].shape[0]])
    H.assert_soa_close(cr['seg'], got, 'seg')
    H.assert_soa_close(cr['op'], pk.op, 'op')
    r_ok = int(np.flatnonzero(pk.status == abi.OK)[0])
    ray, op, _w = pk.pkg(r_ok)
    assert len(ray) == N and abs(op - cr['op'][r_ok]) <= 1e-10 * max(1.0, abs(cr['op'][r_ok]))
    # RayList: deferred list -> one HITS_COMPACT launch per refocus
    rng = np.random.default_rng(8)
    pupil = rng.uniform(-1.05, 1.05, (500, 2))
    lst = analyses.trace_pupil_coords(m, [p.copy() for p in pupil], m.fields[0], fx.table.wvls[1], 0.0)
    assert isinstance(lst, analyses._DeferredRayList)
    for foc in (0.0, 0.05):
        xy = analyses.focus_pupil_coords(m, lst, m.fields[0], fx.table.wvls[1], foc)
        opts = oracle.make_opts(flags=FLAGS, out_mode=abi.OUT_HITS_COMPACT, first_surf=1,
                                last_surf=N - 2, foc=foc, image_pt=(0.0, 18.0))
        orc = oracle.trace_pupil_list(fx.table, m.fields[0].rox_field, pupil[:, 0].copy(),
                                      pupil[:, 1].copy(), 1, opts)
        np.testing.assert_array_equal(xy, orc.hits)
    session.clear()


def test_single_ray_trace_product_function():
    """trace.raytrace_trace (what rayoptics.raytr.raytrace.trace is rebound to):
    one ray per call through a pinned block, list-of-lists result, the raised
    TraceError with the partial packet -- against the golden explicit rays the
    reference traced (tests/golden) and bit-exactly against the oracle"""
    from oracle import oracle
    from rayoptics_amd import trace, session
    from rayoptics_amd.traceerror import TraceError
    for name in ('dblgauss', 'cell_phone'):
        fx = H.fixture(name)
        cr = fx['rays_ap']
        N = fx.table.n_ifcs
        m = model_of(fx, [H.field_from_arr(fx['grid_f2']['field'])], [(0., 0.)])
        sm = m['seq_model']
        R = cr['pt0'].shape[1]
        wi_all = np.broadcast_to(np.atleast_1d(cr['wvl_idx']), (R,))
        n_err = 0
        for r in range(0, R, max(1, R // 48)):
            pt0, d0 = cr['pt0'][:, r].copy(), cr['dir0'][:, r].copy()
            wi = int(wi_all[r])
            opts = oracle.make_opts(flags=abi.INTERSECT_OBJ | abi.CHECK_APERTURES,
                                    first_surf=1, last_surf=N - 2)
            o = oracle.trace_rays(fx.table, pt0.reshape(3, 1), d0.reshape(3, 1), wi, opts)
            try:
                ray, op, wvl = trace.raytrace_trace(sm, pt0, d0, fx.table.wvls[wi],
                                                    check_apertures=True)
            except TraceError as e:
                n_err += 1
                assert int(o.status[0]) != abi.OK and e.surf == int(o.fail_surf[0])
                ray, op, wvl = e.ray_pkg
            else:
                assert int(o.status[0]) == abi.OK and len(ray) == N
            assert int(cr['status'][r]) == int(o.status[0])
            assert isinstance(ray, list) and wvl == fx.table.wvls[wi]
            assert op == float(o.op[0])
            for k, seg in enumerate(ray):
                np.testing.assert_array_equal(seg[0], o.seg[k, 0:3, 0])
                np.testing.assert_array_equal(seg[1], o.seg[k, 3:6, 0])
                assert seg[2] == o.seg[k, 6, 0]
                np.testing.assert_array_equal(seg[3], o.seg[k, 7:10, 0])
        assert n_err >= 1
    session.clear()


def test_trace_raw_on_an_explicit_path_list():
    """trace.raytrace_trace_raw (what raytrace.trace_raw is rebound to): a path of
    (Intfc, Gap, Tfrm, Index, Z_Dir) tuples -- stand-in objects with the attributes the
    reference's carry -- flattened per call, the handle cached by the table's bytes;
    first_surf / last_surf defaulting to 0 / None as in trace_raw"""
    from oracle import oracle
    from rayoptics_amd import SurfaceTable, session, trace
    from rayoptics_amd.traceerror import TraceError

    class Spherical:
        def __init__(self, cv):
            self.cv = cv

    class Conic(Spherical):
        def __init__(self, cv, cc):
            self.cv, self.cc, self.ec = cv, cc, cc + 1.0

    class Circular:
        def __init__(self, radius):
            self.radius = radius

    class Ifc:
        def __init__(self, profile, mode='transmit', max_aperture=1.0, ca=None):
            self.profile, self.interact_mode, self.max_aperture = profile, mode, max_aperture
            self.clear_apertures = ca or []

    eye, zero = np.identity(3), np.zeros(3)
    th = np.deg2rad(3.0)
    tilt = np.array([[1, 0, 0], [0, np.cos(th), -np.sin(th)], [0, np.sin(th), np.cos(th)]])

    def path():     # a generator, as seq_model.path() hands one out
        yield (Ifc(Spherical(0.0), 'dummy'), None, (eye, np.array([0., 0., 50.])), 1.0, 1.0)
        yield (Ifc(Conic(0.02, -0.4), max_aperture=9.0, ca=[Circular(9.0)]), None,
               (tilt.T, np.array([0., 0.3, 4.])), 1.5168, 1.0)
        yield (Ifc(Spherical(-0.015), max_aperture=9.0), None, (eye, np.array([0., 0., 60.])), 1.0, 1.0)
        yield (Ifc(Spherical(0.0), 'dummy'), None, None, 1.0, 1.0)
    tbl = SurfaceTable.from_paths([list(path())], [550.0])
    rng = np.random.default_rng(2)
    n_err = 0
    for k in range(40):
        pt0 = np.array([rng.uniform(-6, 6), rng.uniform(-6, 6), 0.])
        d0 = np.array([rng.uniform(-.15, .15), rng.uniform(-.15, .15), 1.])
        d0 /= np.linalg.norm(d0)
        ca = bool(k % 2)
        opts = oracle.make_opts(flags=abi.INTERSECT_OBJ | (abi.CHECK_APERTURES if ca else 0))
        o = oracle.trace_rays(tbl, pt0.reshape(3, 1), d0.reshape(3, 1), 0, opts)
        try:
            ray, op, wvl = trace.raytrace_trace_raw(path(), pt0, d0, 550.0, check_apertures=ca)
            assert int(o.status[0]) == abi.OK and len(ray) == 4
        except TraceError as e:
            n_err += 1
            assert int(o.status[0]) != abi.OK and e.surf == int(o.fail_surf[0])
            ray, op, wvl = e.ray_pkg
        assert op == float(o.op[0]) and wvl == 550.0
        for j, seg in enumerate(ray):
            np.testing.assert_array_equal(seg[0], o.seg[j, 0:3, 0])
            np.testing.assert_array_equal(seg[1], o.seg[j, 3:6, 0])
            assert seg[2] == o.seg[j, 6, 0]
    assert n_err >= 1
    assert len(session._by_table) == 1              # one handle for the 40 calls
    session.clear()


def test_config5_full_size_on_one_gpu():
    """BASELINE configs[4] in full: 9 fields x 5 wavelengths x 2048 x 2048 pupil
    grids (188.7 M rays) of the 44-interface lithography lens imported from
    rayoptics/zemax/tests/US05831776-1.zmx, traced on ONE GPU through the multi-GPU
    path's own code (dist.partition / trace_blocks: 45 packed-hit launches appended
    into one buffer, counts kept on the device).  Per grid: a 64-row block of the
    unpacked HITS trace bit-exact vs the oracle, the packed block identical to the
    survivors of that HITS trace in ray order; and the whole-job invariants (every
    ray accounted for, survivors finite, vignetted fraction sane)"""
    import torch
    from oracle import oracle
    from rayoptics_amd import workloads, dist as rdist
    from rayoptics_amd.engine import TraceEngine, DeviceResult, make_opts, make_grid
    wl = workloads.load('litho_c5')
    assert wl.n_ifcs == 44 and len(wl.fields) == 9 and len(wl.table.wvls) == 5
    eng = TraceEngine(wl.table)
    num = 2048
    plan = rdist.partition(9, 5, num, 1)
    cap = rdist.rays_of(plan[0], num)
    assert cap == 45 * num * num and len(plan[0]) == 45
    pack = rdist.trace_blocks(eng, plan[0], num, wl.fields, wl.image_pts, wl.foc)
    counts = pack.counts()
    assert pack.rays == cap and len(counts) == 45
    n_ok = int(counts.sum())
    assert int(pack.count.item()) == n_ok
    assert 0.55 * cap < n_ok < 0.85 * cap                   # the circular pupil in the square grid
    assert bool(torch.isfinite(pack.xy[:n_ok]).all().item())
    N = wl.n_ifcs
    rng = np.random.default_rng(5)
    hits = DeviceResult(torch, eng.device, 0, num * num, abi.OUT_HITS, want_pupil=False, nan_fill=True)
    offs = np.concatenate([[0], np.cumsum(counts)])
    for g, b in enumerate(plan[0]):
        opts = make_opts(flags=FLAGS, out_mode=abi.OUT_HITS, first_surf=1, last_surf=N - 2,
                         foc=wl.foc, image_pt=wl.image_pts[b.fi])
        eng.trace_pupil_grid(wl.fields[b.fi], make_grid((-1., -1.), (1., 1.), num), b.wi, opts,
                             want_pupil=False, out=hits)
        ok = hits.status == 0
        assert int((hits.status == 255).sum().item()) == 0      # every ray was traced
        assert int(ok.sum().item()) == counts[g]
        assert torch.equal(hits.seg[:, ok].T.contiguous(), pack.xy[offs[g]:offs[g + 1]]), b
        i0 = int(rng.integers(0, num - 64))
        grid = oracle.make_grid((-1., -1.), (1., 1.), num, row_begin=i0, row_count=64)
        orc = oracle.trace_pupil_grid(wl.table, wl.fields[b.fi], grid, b.wi, opts)
        lo, hi = i0 * num, (i0 + 64) * num
        np.testing.assert_array_equal(hits.status[lo:hi].cpu().numpy(), orc.status)
        got = hits.seg[:, lo:hi].cpu().numpy()
        same = (got == orc.seg) | (~(orc.status == 0))[None, :]
        assert same.all(), (b, int((~same).sum()))
    eng.close()


def test_ingested_prescriptions_trace_like_the_oracle():
    """BASELINE's '.zmx import' / 'CODE V .seq' systems: tables parsed by
    rayoptics_amd.ingest from the reference's prescription files (both EVENASPH
    .zmx files, the 44-interface lithography lens, the CODE V double Gauss and
    Ritchey-Chretien, the decentered three-mirror telescope and off-axis parabola,
    a COORDBRK fold, two .roa models), traced on the device vs the oracle"""
    import json
    import os
    from oracle import oracle
    from rayoptics_amd import SurfaceTable
    from rayoptics_amd.engine import TraceEngine
    with open(os.path.join(H.GOLDEN, 'ingest_tables.json')) as f:
        tabs = json.load(f)
    rng = np.random.default_rng(21)
    for key, rec in tabs.items():
        tbl = SurfaceTable.from_dict(rec['table'])
        N = tbl.n_ifcs
        eng = TraceEngine(tbl)
        R = 3000
        ap1 = tbl.rows[1].max_aperture
        z0 = tbl.rows[0].t[2]
        z0 = z0 if np.isfinite(z0) and abs(z0) < 1e6 else 1e3
        base = abi.INTERSECT_OBJ | abi.CHECK_APERTURES
        if key in ('seq_threemir', 'seq_codv_35571'):
            # decentered / tilted mirrors; the .seq files size no apertures (the reference
            # derives them by tracing): a fixed bundle, no aperture checks
            ap1, z0, base = 10.0, 1e3, abi.INTERSECT_OBJ
        # rays from the axial object point (or a far point) into the first aperture
        tgt = np.stack([rng.uniform(-ap1, ap1, R), rng.uniform(-ap1, ap1, R), np.full(R, z0)])
        pt0 = np.zeros((3, R))
        pt0[2] = tbl.rows[0].t[2] - z0
        d = tgt - np.stack([np.zeros(R), np.zeros(R), np.zeros(R)])
        d /= np.linalg.norm(d, axis=0)
        wi = (np.arange(R) % len(tbl.wvls)).astype(np.int32)
        for mode in (abi.OUT_FULL, abi.OUT_HITS):
            opts = oracle.make_opts(flags=base, out_mode=mode,
                                    first_surf=1, last_surf=N - 2, foc=0.0)
            with np.errstate(all='ignore'):
                orc = oracle.trace_rays(tbl, pt0, d, wi, opts)
            dev = eng.trace_rays(pt0, d, wi, opts, nan_fill=True).to_host()
            np.testing.assert_array_equal(dev.status, orc.status, err_msg=key)
            np.testing.assert_array_equal(dev.fail_surf, orc.fail_surf, err_msg=key)
            same = (dev.seg == orc.seg) | (np.isnan(dev.seg) & np.isnan(orc.seg))
            assert same.all(), (key, mode, int((~same).sum()))
            assert np.array_equal(dev.op, orc.op, equal_nan=True), key
        assert (orc.status == abi.OK).sum() > 20, (key, int((orc.status == abi.OK).sum()))
        eng.close()


def test_decentered_roa_parsed_here_traces_like_the_oracle():
    """tests/golden/decentered.roa (DecenterData records, a coordinate break, aperture lists)
    parsed on this box by rayoptics_amd.ingest -- no reference involved -- gives the table of
    the `tilted_singlet` golden fixture, and the device traces it like the oracle"""
    import os
    from oracle import oracle
    from rayoptics_amd import ingest
    from rayoptics_amd.engine import TraceEngine
    fx = H.fixture('tilted_singlet')
    pres = ingest.read_roa(os.path.join(H.GOLDEN, 'decentered.roa'))
    tbl = pres.to_table(wvls=fx.table.wvls, index_of=ingest.reference_fallback_index)
    tbl.n_table[:] = fx.table.n_table
    for a, b in zip(tbl.rows, fx.table.rows):
        assert bytes(a) == bytes(b)
    rng = np.random.default_rng(5)
    R = 4096 + 33
    pt0, d = H.random_rays(rng, R, tbl.rows[0].t[2], spread=7.0)
    wi = (np.arange(R) % len(tbl.wvls)).astype(np.int32)
    eng = TraceEngine(tbl)
    opts = oracle.make_opts(flags=abi.INTERSECT_OBJ | abi.CHECK_APERTURES, out_mode=abi.OUT_FULL,
                            first_surf=1, last_surf=tbl.n_ifcs - 2)
    orc = oracle.trace_rays(tbl, pt0, d, wi, opts)
    dev = eng.trace_rays(pt0, d, wi, opts, nan_fill=True).to_host()
    np.testing.assert_array_equal(dev.status, orc.status)
    H.bit_equal(dev.seg, orc.seg, 'decentered.roa')
    assert 100 < int((orc.status == 0).sum()) < R
    eng.close()


@pytest.mark.parametrize('script,args,expect', [('spot_diagram.py', ['96'], 'rms spot radius'),
                                                 ('wavefront_psf.py', ['32', '128'], 'Strehl'),
                                                 ('spot_stats.py', ['128'], 'rms spot radius')])
def test_examples_run(script, args, expect):
    """the stand-alone examples run as written (subprocess, from the repo root)"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, 'examples', script)] + args,
                       capture_output=True, text=True, timeout=600, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if expect in ln]
    assert len(lines) >= 2, r.stdout[-1000:]
