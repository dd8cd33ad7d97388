"""-m gpu: the HIP kernels, called through the C ABI, against
  * the CPU oracle on the same inputs -- BIT-EXACT (binary64, same operation
    order, explicit fma at the BLAS sites), and
  * the reference's own outputs stored in tests/golden -- within 1e-10
    (north_star's tolerance on ray intercepts, applied relative to max(1,|ref|)).
"""
import ctypes as C

import numpy as np
import pytest

from rayoptics_amd import abi
import helpers as H

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def engines():
    from rayoptics_amd.engine import TraceEngine
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = TraceEngine(H.fixture(name).table)
        return cache[name]
    yield get
    for e in cache.values():
        e.close()


def bit_equal(a, b, what):
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    same = (a == b) | (np.isnan(a) & np.isnan(b))
    assert same.all(), f'{what}: {np.count_nonzero(~same)} of {same.size} entries differ, first at {np.argwhere(~same)[:3].tolist()}'
    # signed zeros too
    assert np.array_equal(np.signbit(a[~np.isnan(a)]), np.signbit(b[~np.isnan(b)])), f'{what}: zero signs differ'


def assert_same_as_oracle(dev, orc, what):
    np.testing.assert_array_equal(dev.status, orc.status, err_msg=what)
    np.testing.assert_array_equal(dev.fail_surf, orc.fail_surf, err_msg=what)
    bit_equal(dev.seg, orc.seg, what + ' seg')
    bit_equal(dev.op, orc.op, what + ' op')
    if getattr(orc, 'pupil', None) is not None and dev.pupil is not None:
        bit_equal(dev.pupil, orc.pupil, what + ' pupil')


from test_oracle_golden import RAY_CASES, GRID_CASES  # noqa: E402


@pytest.mark.parametrize('name,case', RAY_CASES)
def test_explicit_rays(engines, name, case):
    from oracle import oracle
    fx = H.fixture(name)
    c = fx[case]
    wi = c['wvl_idx'] if 'wvl_idx' in c else 0
    opts = H.make_opts(c)
    dev = engines(name).trace_rays(c['pt0'], c['dir0'], wi, opts, nan_fill=True).to_host()
    orc = oracle.trace_rays(fx.table, c['pt0'], c['dir0'], wi, opts)
    assert_same_as_oracle(dev, orc, f'{name}/{case}')
    H.assert_result_matches(c, dev)                 # and vs the reference itself


@pytest.mark.parametrize('name,case', GRID_CASES)
def test_pupil_grid(engines, name, case):
    from oracle import oracle
    fx = H.fixture(name)
    c = fx[case]
    fld = H.field_from_arr(c['field'])
    grid = oracle.make_grid(c['start'], c['stop'], int(c['num']), int(c['kind']))
    opts = H.make_opts(c)
    dev = engines(name).trace_pupil_grid(fld, grid, int(c['wvl_idx']), opts, nan_fill=True).to_host()
    orc = oracle.trace_pupil_grid(fx.table, fld, grid, int(c['wvl_idx']), opts)
    assert_same_as_oracle(dev, orc, f'{name}/{case}')
    np.testing.assert_array_equal(dev.pupil, c['pupil'])
    H.assert_result_matches(c, dev)


@pytest.mark.parametrize('name', ['dblgauss', 'singlet', 'rc_telescope', 'nikkor'])
def test_spot_hits(engines, name):
    """HITS mode == SpotDiagramFigure's data (reference consumer, unchanged)"""
    from oracle import oracle
    fx = H.fixture(name)
    c = fx['spot']
    num = int(c['num'])
    N = fx.table.n_ifcs
    for key in [k for k in c if k.endswith('_hits')]:
        fi, wi = key.split('_')[0], int(key.split('_')[1][1:])
        fld = H.field_from_arr(c[f'{fi}_field'])
        opts = oracle.make_opts(flags=abi.INTERSECT_OBJ | abi.CHECK_APERTURES | abi.APPLY_VIGNETTING,
                                out_mode=abi.OUT_HITS, first_surf=1, last_surf=N - 2,
                                foc=float(c['foc']), image_pt=tuple(c[f'{fi}_image_pt']))
        grid = oracle.make_grid((-1., -1.), (1., 1.), num)
        dev = engines(name).trace_pupil_grid(fld, grid, wi, opts, nan_fill=True).to_host()
        ok = dev.status == abi.OK
        H.assert_soa_close(c[key], dev.seg[:, ok].T, key, require_exact=True)


@pytest.mark.parametrize('out_mode', [abi.OUT_LAST, abi.OUT_HITS])
def test_reduced_outputs_match_full(engines, out_mode):
    from oracle import oracle
    fx = H.fixture('dblgauss')
    c = fx['rays_ap']
    N = fx.table.n_ifcs
    opts = oracle.make_opts(flags=int(c['flags']), out_mode=out_mode, first_surf=1,
                            last_surf=N - 2, foc=0.0125, image_pt=(0.01, -0.02))
    dev = engines('dblgauss').trace_rays(c['pt0'], c['dir0'], c['wvl_idx'], opts, nan_fill=True).to_host()
    orc = oracle.trace_rays(fx.table, c['pt0'], c['dir0'], c['wvl_idx'], opts)
    assert_same_as_oracle(dev, orc, f'out_mode {out_mode}')


def test_list_of_rays_last(engines):
    fx = H.fixture('dblgauss')
    c = fx['list_last']
    N = fx.table.n_ifcs
    from oracle import oracle
    opts = oracle.make_opts(flags=abi.INTERSECT_OBJ | abi.CHECK_APERTURES,
                            out_mode=abi.OUT_LAST, first_surf=1, last_surf=N - 2)
    dev = engines('dblgauss').trace_rays(c['pt0'], c['dir0'], c['wvl_idx'], opts, nan_fill=True).to_host()
    np.testing.assert_array_equal(dev.status, c['status'])
    H.assert_soa_close(c['last'], dev.seg, 'last', require_exact=True)


def test_filter_out_phantoms(engines):
    """raytrace.py:185-188: the coordinate break's segment is dropped and its
    path length folded into the previous segment"""
    from oracle import oracle
    fx = H.fixture('tilted_singlet')
    c = fx['rays_ap']
    assert fx.table.has_phantoms()
    opts = H.make_opts(c)
    opts.flags |= abi.FILTER_PHANTOMS
    eng = engines('tilted_singlet')
    assert eng.num_segments(opts.flags) == fx.table.n_ifcs - 1
    dev = eng.trace_rays(c['pt0'], c['dir0'], c['wvl_idx'], opts, nan_fill=True).to_host()
    orc = oracle.trace_rays(fx.table, c['pt0'], c['dir0'], c['wvl_idx'], opts)
    np.testing.assert_array_equal(dev.status, orc.status)
    bit_equal(dev.seg, orc.seg[:dev.seg.shape[0]], 'filtered seg')
    bit_equal(dev.op, orc.op, 'filtered op')


def test_intersect_obj_false_and_surface_ranges(engines):
    """wide-angle entry (trace.py:302-303) and non-default first/last_surf"""
    from oracle import oracle
    fx = H.fixture('dblgauss_finite')
    c = fx['rays_ap']
    for flags, fs, ls in [(abi.CHECK_APERTURES, 0, -1), (abi.INTERSECT_OBJ, 2, 7),
                          (abi.INTERSECT_OBJ | abi.CHECK_APERTURES, 3, 3)]:
        opts = oracle.make_opts(flags=flags, first_surf=fs, last_surf=ls)
        dev = engines('dblgauss_finite').trace_rays(c['pt0'], c['dir0'], c['wvl_idx'], opts, nan_fill=True).to_host()
        orc = oracle.trace_rays(fx.table, c['pt0'], c['dir0'], c['wvl_idx'], opts)
        assert_same_as_oracle(dev, orc, f'flags={flags} fs={fs} ls={ls}')


def test_empty_and_ragged_batches(engines):
    from oracle import oracle
    fx = H.fixture('dblgauss')
    c = fx['rays_ap']
    eng = engines('dblgauss')
    opts = H.make_opts(c)
    res = eng.trace_rays(np.zeros((3, 0)), np.zeros((3, 0)), 0, opts)
    assert res.seg.shape[-1] == 0
    for R in (1, 63, 65, 257):
        dev = eng.trace_rays(c['pt0'][:, :R], c['dir0'][:, :R], c['wvl_idx'][:R], opts, nan_fill=True).to_host()
        orc = oracle.trace_rays(fx.table, c['pt0'][:, :R], c['dir0'][:, :R], c['wvl_idx'][:R], opts)
        assert_same_as_oracle(dev, orc, f'R={R}')


def test_host_pointer_mode(engines):
    """ROX_HOST_POINTERS: plain numpy buffers straight through the C ABI"""
    from oracle import oracle
    from rayoptics_amd.engine import load_library
    fx = H.fixture('rc_telescope')
    c = fx['grid_f4']
    eng = engines('rc_telescope')
    fld = H.field_from_arr(c['field'])
    grid = oracle.make_grid(c['start'], c['stop'], int(c['num']), int(c['kind']))
    opts = H.make_opts(c)
    orc = oracle.trace_pupil_grid(fx.table, fld, grid, int(c['wvl_idx']), opts)
    opts.flags |= abi.HOST_POINTERS
    res = oracle.HostResult(fx.table.n_ifcs, orc.R, abi.OUT_FULL, want_pupil=True)
    out = res.out_struct()
    rc = load_library().rox_trace_pupil_grid(eng._handle, C.byref(fld), C.byref(grid),
                                             int(c['wvl_idx']), C.byref(opts), C.byref(out), None)
    assert rc == 0, load_library().rox_last_error()
    assert_same_as_oracle(res, orc, 'host pointers')


def test_full_size_grid_bit_exact_vs_oracle(engines):
    """BASELINE.json config 2 at full size: 1024x1024 pupil grid through the
    13-interface double Gauss, every ray compared with the oracle (HITS and
    status), plus size-independent properties."""
    from oracle import oracle
    fx = H.fixture('dblgauss')
    c = fx['grid_f2']
    N = fx.table.n_ifcs
    fld = H.field_from_arr(c['field'])
    grid = oracle.make_grid((-1., -1.), (1., 1.), 1024)
    opts = oracle.make_opts(flags=abi.INTERSECT_OBJ | abi.CHECK_APERTURES | abi.APPLY_VIGNETTING,
                            out_mode=abi.OUT_HITS, first_surf=1, last_surf=N - 2)
    dev = engines('dblgauss').trace_pupil_grid(fld, grid, 1, opts, nan_fill=True).to_host()
    orc = oracle.trace_pupil_grid(fx.table, fld, grid, 1, opts)
    assert_same_as_oracle(dev, orc, '1024x1024 HITS')
    # properties: the system is symmetric in x for a y-field, so the spot is too
    ok = (dev.status == 0).reshape(1024, 1024)
    assert 0.3 < ok.mean() < 0.99
    # FULL packets of the same grid: every segment chain is geometrically
    # consistent (next point = point + dst * direction, in the next frame)
    opts.out_mode = abi.OUT_FULL
    full = engines('dblgauss').trace_pupil_grid(fld, grid, 1, opts, nan_fill=True)
    seg = full.seg
    import torch
    good = (full.status == 0)
    t_z = torch.tensor([fx.table.rows[k].t[2] for k in range(N)], dtype=torch.float64, device=seg.device)
    for k in range(1, N - 1):
        p, d, dst = seg[k, 0:3][:, good], seg[k, 3:6][:, good], seg[k, 6][good]
        nxt = seg[k + 1, 0:3][:, good].clone()
        nxt[2] += t_z[k]
        err = (p + dst * d - nxt).abs().max().item()
        assert err < 1e-9, (k, err)
        assert ((d * d).sum(0) - 1).abs().max().item() < 1e-12


def test_row_blocks_tile_the_full_grid(engines):
    """multi-GPU sharding unit: pupil row blocks reproduce the full grid's rays
    bit for bit (coordinates still come from the full accumulate-by-step axis)"""
    from oracle import oracle
    fx = H.fixture('dblgauss')
    c = fx['grid_f2']
    N = fx.table.n_ifcs
    fld = H.field_from_arr(c['field'])
    opts = H.make_opts(c)
    num = 37
    eng = engines('dblgauss')
    full = eng.trace_pupil_grid(fld, oracle.make_grid((-1., -1.), (1., 1.), num), 2, opts,
                                nan_fill=True).to_host()
    for r0, rc in [(0, 5), (5, 20), (25, 12)]:
        g = oracle.make_grid((-1., -1.), (1., 1.), num, row_begin=r0, row_count=rc)
        part = eng.trace_pupil_grid(fld, g, 2, opts, nan_fill=True).to_host()
        orc = oracle.trace_pupil_grid(fx.table, fld, g, 2, opts)
        assert_same_as_oracle(part, orc, f'rows {r0}+{rc}')
        sl = slice(r0 * num, (r0 + rc) * num)
        bit_equal(part.seg, full.seg[:, :, sl], 'row block vs full grid')
        bit_equal(part.pupil, full.pupil[:, sl], 'row block pupil')


def test_sharded_spot_single_rank(engines):
    """dist.trace_spot_sharded with world size 1 (no process group) on the GPU"""
    from oracle import oracle
    from rayoptics_amd import workloads
    from rayoptics_amd.dist import trace_spot_sharded
    from rayoptics_amd.engine import TraceEngine
    wl = workloads.load('rc_telescope_c4')
    eng = TraceEngine(wl.table)
    N = wl.n_ifcs
    for by in ('rows', 'field'):
        tm = {}
        out = trace_spot_sharded(eng, wl.fields, wl.image_pts, 1, 24, wl.foc, by=by, timings=tm)
        assert len(out) == 5
        for (fi, wi), xy in out.items():
            opts = oracle.make_opts(flags=abi.INTERSECT_OBJ | abi.CHECK_APERTURES | abi.APPLY_VIGNETTING,
                                    out_mode=abi.OUT_HITS_COMPACT, first_surf=1, last_surf=N - 2,
                                    foc=wl.foc, image_pt=wl.image_pts[fi])
            ref = oracle.trace_pupil_grid(wl.table, wl.fields[fi],
                                          oracle.make_grid((-1., -1.), (1., 1.), 24), wi, opts)
            bit_equal(xy, ref.hits, f'field {fi}')
        assert tm['pairs_total'] == sum(len(v) for v in out.values()) < 5 * 24 * 24
    eng.close()


def test_slim_fp64_paths_equal_ieee_operators():
    """the exponent-band-guarded sqrt and shared-reciprocal division used by
    the kernels are bit-identical to sqrt() and `/` -- 2^27 operand sets over
    the whole exponent range, zeros, denormals, band edges, inf, nan"""
    from rayoptics_amd.engine import load_library
    import torch
    assert torch.cuda.is_available()
    torch.zeros(1, device='cuda')
    lib = load_library()
    counts = (C.c_uint64 * 4)()
    for seed in (1, 2):
        rc = lib.rox_selftest_fp64(1 << 27, seed, counts)
        assert rc == 0, lib.rox_last_error()
        assert counts[0] == 0 and counts[1] == 0, list(counts)
        assert counts[2] > (1 << 27) * 0.5          # the guarded paths were exercised


from test_oracle_golden import OPD_CASES, opd_opts, check_opd_grid  # noqa: E402


@pytest.mark.parametrize('name,case', OPD_CASES)
def test_opd_mode(engines, name, case):
    """ROX_OUT_OPD: wave_abr_full_calc_finite_pup fused into the trace epilogue"""
    from oracle import oracle
    fx = H.fixture(name)
    c = fx[case]
    fld = H.field_from_arr(c['field'])
    grid = oracle.make_grid(c['start'], c['stop'], int(c['num']))
    opts = opd_opts(c)
    dev = engines(name).trace_pupil_grid(fld, grid, int(c['wvl_idx']), opts, nan_fill=True).to_host()
    orc = oracle.trace_pupil_grid(fx.table, fld, grid, int(c['wvl_idx']), opts)
    assert_same_as_oracle(dev, orc, f'{name}/{case}')
    check_opd_grid(c, dev, exact=True)


def test_opd_full_size_wavefront(engines):
    """1024x1024 OPD map of the double Gauss edge field: every ray vs the oracle"""
    from oracle import oracle
    fx = H.fixture('dblgauss')
    c = fx['opd_f2']
    fld = H.field_from_arr(c['field'])
    grid = oracle.make_grid(c['start'], c['stop'], 1024)
    opts = opd_opts(c)
    dev = engines('dblgauss').trace_pupil_grid(fld, grid, int(c['wvl_idx']), opts, nan_fill=True).to_host()
    orc = oracle.trace_pupil_grid(fx.table, fld, grid, int(c['wvl_idx']), opts)
    assert_same_as_oracle(dev, orc, 'OPD 1024x1024')
    ok = dev.status == 0
    assert ok.mean() > 0.3
    waves = float(c['convert_to_opd']) * dev.seg[0][ok]
    assert np.abs(waves).max() < 200           # a real wavefront, not garbage


def test_edge_grids_and_pitches(engines):
    """num = 1 and 2 grids (step = x/0 is never used for a ray), explicit row
    pitches (ld > R, odd), and per-ray wavelengths over a ragged batch"""
    from oracle import oracle
    from rayoptics_amd.engine import DeviceResult
    import torch
    fx = H.fixture('dblgauss')
    c = fx['grid_f2']
    eng = engines('dblgauss')
    fld = H.field_from_arr(c['field'])
    opts = H.make_opts(c)
    for num, start, stop in [(1, (0.25, -0.5), (1., 1.)), (2, (-0.3, -0.3), (0.3, 0.3)), (3, (0., 0.), (0., 0.))]:
        grid = oracle.make_grid(start, stop, num)
        with np.errstate(all='ignore'):
            orc = oracle.trace_pupil_grid(fx.table, fld, grid, 0, opts)
        dev = eng.trace_pupil_grid(fld, grid, 0, opts, nan_fill=True).to_host()
        assert_same_as_oracle(dev, orc, f'num={num}')
    # explicit pitches
    grid = oracle.make_grid((-1., -1.), (1., 1.), 23)
    orc = oracle.trace_pupil_grid(fx.table, fld, grid, 1, opts)
    for ld in (529, 530, 1001, 4096):
        out = DeviceResult(torch, eng.device, eng.num_segments(opts.flags), 529, abi.OUT_FULL,
                           want_pupil=True, nan_fill=True, ld=ld)
        dev = eng.trace_pupil_grid(fld, grid, 1, opts, out=out).to_host()
        assert_same_as_oracle(dev, orc, f'ld={ld}')
    # per-ray wavelengths, ragged
    cr = fx['rays_ap']
    R = 333
    wi = (np.arange(R) % 3).astype(np.int32)
    o2 = H.make_opts(cr)
    dev = eng.trace_rays(cr['pt0'][:, :R], cr['dir0'][:, :R], wi, o2, nan_fill=True).to_host()
    orc = oracle.trace_rays(fx.table, cr['pt0'][:, :R], cr['dir0'][:, :R], wi, o2)
    assert_same_as_oracle(dev, orc, 'per-ray wavelengths')


def test_argument_errors_are_reported_not_crashes(engines):
    from rayoptics_amd.engine import EngineError, make_opts, make_grid
    fx = H.fixture('dblgauss')
    eng = engines('dblgauss')
    fld = H.field_from_arr(fx['grid_f2']['field'])
    with pytest.raises(EngineError, match='wvl_idx'):
        eng.trace_pupil_grid(fld, make_grid((-1, -1), (1, 1), 4), 7, make_opts())
    with pytest.raises(EngineError, match='row block'):
        eng.trace_pupil_grid(fld, make_grid((-1, -1), (1, 1), 4, row_begin=3, row_count=2), 0, make_opts())
    with pytest.raises(EngineError, match='OPD'):
        eng.trace_pupil_grid(fld, make_grid((-1, -1), (1, 1), 4), 0, make_opts(out_mode=abi.OUT_OPD))
    with pytest.raises(EngineError, match='out_mode'):
        eng.trace_pupil_grid(fld, make_grid((-1, -1), (1, 1), 4), 0, make_opts(out_mode=9))


def test_large_table_many_wavelengths():
    """44 interfaces x 5 wavelengths staged in LDS (the largest prescription
    shape in the reference tree), per-ray wavelength gather"""
    from oracle import oracle
    from rayoptics_amd import SurfaceTable
    from rayoptics_amd.engine import TraceEngine
    surfs = [dict(cv=0.0, thi=500.0, n=1.0, max_aperture=1e9)]
    rng = np.random.default_rng(11)
    for k in range(21):
        n = [1.5 + 0.01 * k + 0.002 * w for w in range(5)]
        surfs.append(dict(cv=0.004 * (1 + k % 3), thi=3.0, n=n, max_aperture=20.0,
                          profile='Conic' if k % 4 == 1 else 'Spherical', cc=-0.4))
        surfs.append(dict(cv=-0.003 * (1 + k % 2), thi=8.0, n=1.0, max_aperture=20.0))
    surfs.append(dict(cv=0.0, thi=0.0, n=1.0, max_aperture=100.0))
    tbl = SurfaceTable.from_prescription(surfs, wvls=(450., 500., 550., 600., 650.))
    assert tbl.n_ifcs == 44
    eng = TraceEngine(tbl)
    R = 5000
    pt0 = np.stack([rng.uniform(-5, 5, R), rng.uniform(-5, 5, R), np.zeros(R)])
    d = np.stack([rng.uniform(-.02, .02, R), rng.uniform(-.02, .02, R), np.ones(R)])
    d /= np.linalg.norm(d, axis=0)
    wi = rng.integers(0, 5, R).astype(np.int32)
    opts = oracle.make_opts(flags=abi.INTERSECT_OBJ | abi.CHECK_APERTURES, first_surf=1, last_surf=42)
    dev = eng.trace_rays(pt0, d, wi, opts, nan_fill=True).to_host()
    orc = oracle.trace_rays(tbl, pt0, d, wi, opts)
    assert_same_as_oracle(dev, orc, '44 interfaces x 5 wvls')
    assert (dev.status == 0).mean() > 0.2
    eng.close()


def test_cell_phone_doc_table_kat_on_device(engines):
    """the HIP Newton path against the ray table printed in the reference's
    documentation (Cell_Phone_lens.rst:311-328)"""
    from test_oracle_golden import CELL_PHONE_DOC_MARGINAL as doc
    fx = H.fixture('cell_phone')
    N = fx.table.n_ifcs
    from oracle import oracle
    opts = oracle.make_opts(flags=abi.INTERSECT_OBJ, first_surf=1, last_surf=N - 2)
    dev = engines('cell_phone').trace_rays(np.array([[0.], [1.], [0.]]), np.array([[0.], [0.], [1.]]),
                                           1, opts, nan_fill=True).to_host()
    assert dev.status[0] == abi.OK
    seg = dev.seg[:, :, 0]
    np.testing.assert_allclose(seg[:, 1], doc[:, 0], atol=6e-6)
    np.testing.assert_allclose(seg[:, 2], doc[:, 1], atol=6e-6, rtol=6e-5)
    np.testing.assert_allclose(seg[:, 4], doc[:, 2], atol=6e-7)
    np.testing.assert_allclose(seg[:, 5], doc[:, 3], atol=6e-7)
    np.testing.assert_allclose(seg[1:, 6], doc[1:, 4], rtol=6e-5, atol=6e-6)


def test_pupil_list_entry(engines):
    """rox_trace_pupil_list (analyses.trace_ray_list): explicit pupil
    coordinates, device and host-pointer forms, vs the oracle"""
    from oracle import oracle
    from rayoptics_amd.engine import load_library
    fx = H.fixture('dblgauss')
    c = fx['grid_f2']
    eng = engines('dblgauss')
    fld = H.field_from_arr(c['field'])
    rng = np.random.default_rng(4)
    R = 1500
    px, py = rng.uniform(-1.1, 1.1, R), rng.uniform(-1.1, 1.1, R)
    for mode in (abi.OUT_FULL, abi.OUT_LAST, abi.OUT_HITS):
        opts = H.make_opts(c, out_mode=mode, foc=0.02, image_pt=(0.3, 18.0))
        orc = oracle.trace_pupil_list(fx.table, fld, px, py, 2, opts)
        dev = eng.trace_pupil_list(fld, px, py, 2, opts, nan_fill=True).to_host()
        assert_same_as_oracle(dev, orc, f'pupil list mode {mode}')
    # host pointers straight through the C ABI
    opts = H.make_opts(c)
    orc = oracle.trace_pupil_list(fx.table, fld, px, py, 0, opts)
    opts.flags |= abi.HOST_POINTERS
    res = oracle.HostResult(fx.table.n_ifcs, R, abi.OUT_FULL, want_pupil=True)
    out = res.out_struct()
    rc = load_library().rox_trace_pupil_list(eng._handle, C.byref(fld), R, px.ctypes.data,
                                             py.ctypes.data, 0, C.byref(opts), C.byref(out), None)
    assert rc == 0, load_library().rox_last_error()
    assert_same_as_oracle(res, orc, 'pupil list, host pointers')
    # explicit rays, host pointers
    cr = fx['rays_ap']
    o2 = H.make_opts(cr)
    orc = oracle.trace_rays(fx.table, cr['pt0'], cr['dir0'], cr['wvl_idx'], o2)
    o2.flags |= abi.HOST_POINTERS
    Rr = cr['pt0'].shape[1]
    res = oracle.HostResult(fx.table.n_ifcs, Rr, abi.OUT_FULL)
    out = res.out_struct()
    p0 = np.ascontiguousarray(cr['pt0']); d0 = np.ascontiguousarray(cr['dir0'])
    wi = np.ascontiguousarray(cr['wvl_idx'], dtype=np.int32)
    rc = load_library().rox_trace_rays(eng._handle, Rr, p0.ctypes.data, d0.ctypes.data, wi.ctypes.data, 0,
                                       C.byref(o2), C.byref(out), None)
    assert rc == 0, load_library().rox_last_error()
    assert_same_as_oracle(res, orc, 'explicit rays, host pointers')


@pytest.mark.parametrize('name', ['dblgauss', 'dblgauss_finite', 'singlet', 'rc_telescope', 'nikkor',
                                  'cell_phone', 'tilted_singlet', 'toroid_lens', 'telecentric'])
def test_random_ray_differential_set(engines, name):
    """SURVEY 8(d)'s differential set: 2^20 random rays per fixture (rng 20260925), landing
    uniformly in the first surface's aperture disc from 10...1000 units in front of it with
    directions normalise(U(-.3,.3), U(-.3,.3), 1), random wavelengths; FULL packets, status,
    failing surface and op bit-exact against the oracle, in chunks of 2^18 rays"""
    import threading
    from oracle import oracle
    fx = H.fixture(name)
    tbl = fx.table
    N = tbl.n_ifcs
    eng = engines(name)
    rng = np.random.default_rng(20260925)
    ap1 = float(tbl.rows[1].max_aperture)
    thi0 = float(tbl.rows[0].t[2])
    chunk, n_chunks = 1 << 18, 4
    flags = abi.CHECK_APERTURES                             # rays start in front of surface 1
    opts = oracle.make_opts(flags=flags, first_surf=1, last_surf=N - 2)
    n_ok = 0
    for c in range(n_chunks):
        R = chunk
        rad = ap1 * np.sqrt(rng.uniform(0, 1, R))
        phi = rng.uniform(0, 2 * np.pi, R)
        d = np.stack([rng.uniform(-.3, .3, R), rng.uniform(-.3, .3, R), np.ones(R)])
        d /= np.linalg.norm(d, axis=0)
        dz = rng.uniform(10., 1000., R)
        pt0 = np.stack([rad * np.cos(phi) - dz * d[0] / d[2], rad * np.sin(phi) - dz * d[1] / d[2],
                        thi0 - dz])
        wi = rng.integers(0, len(tbl.wvls), R).astype(np.int32)
        # the oracle is single-threaded C behind ctypes (the GIL is released): 8 slices
        parts = [None] * 8
        bounds = [(R * k) // 8 for k in range(9)]

        def work(k):
            s = slice(bounds[k], bounds[k + 1])
            with np.errstate(all='ignore'):
                parts[k] = oracle.trace_rays(tbl, np.ascontiguousarray(pt0[:, s]),
                                             np.ascontiguousarray(d[:, s]),
                                             np.ascontiguousarray(wi[s]), opts)
        thr = [threading.Thread(target=work, args=(k,)) for k in range(8)]
        for t in thr:
            t.start()
        dev = eng.trace_rays(pt0, d, wi, opts, nan_fill=True).to_host()
        for t in thr:
            t.join()
        status = np.concatenate([p.status for p in parts])
        np.testing.assert_array_equal(dev.status, status, err_msg=f'{name} chunk {c}')
        np.testing.assert_array_equal(dev.fail_surf, np.concatenate([p.fail_surf for p in parts]))
        op = np.concatenate([p.op for p in parts])
        assert np.array_equal(dev.op, op, equal_nan=True), (name, c)
        seg = np.concatenate([p.seg for p in parts], axis=2)
        same = (dev.seg == seg) | (np.isnan(dev.seg) & np.isnan(seg))
        assert same.all(), (name, c, int((~same).sum()))
        n_ok += int((status == abi.OK).sum())
    assert n_ok > 50, (name, n_ok)          # (the RC telescope passes ~1e-4 of such rays)
