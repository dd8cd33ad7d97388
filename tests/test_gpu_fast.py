"""-m gpu: ROX_FAST_FP64, the tolerance mode of the reduced-output modes (include/roxtrace.h;
csrc/rox_device.hpp "tolerance mode").  north_star's bar is 1e-10 on ray intercepts against the
reference's NumPy path; the default kernels are bit-exact, these are not, so here the bar is
asserted directly:

  * every fixture system and every BASELINE configuration, HITS / LAST / OPD / FAN / packed
    hits: fast vs the oracle (small grids) and vs the bit-exact device path (full-size grids)
    <= 1e-10 * max(1, |ref|) on every value (TOL below; observed <= ~1e-12, printed);
  * a ray's status may differ only where its decision margin is within rounding of zero: flips
    are counted, and every flipped ray must lie within 1e-10 (scaled) of the aperture edge /
    TIR limit / miss boundary it was decided at (helpers.boundary_margin, computed from the
    ORACLE's FULL packet of that ray); rays laid within a few ulp of aperture edges and of the
    critical angle make sure the accounting is exercised;
  * ROX_OUT_FULL packets in tolerance mode: every segment of every ray, partial records
    included, within the same bar; under ROX_FILTER_PHANTOMS they stay bit-exact; a batch must
    agree on the flag."""
import numpy as np
import pytest

from rayoptics_amd import abi
import helpers as H

pytestmark = pytest.mark.gpu

TOL = 1e-10
SPOT = abi.INTERSECT_OBJ | abi.CHECK_APERTURES | abi.APPLY_VIGNETTING
WORKLOADS = ['singlet_c1', 'dblgauss_c2', 'zmx_evenasph_c3', 'nikkor_c3', 'rc_telescope_c4', 'litho_c5',
             'cell_phone', 'tt_cassegrain', 'tt_landscape', 'tt_paraboloid', 'tt_singlet_seq',
             'tt_triplet', 'tt_two_mirrors_conic', 'tt_two_sph_mirrors']


def _flags(wl, fi):
    f = wl.fields[fi]
    wide = f.kind == abi.FLD_EPD_WIDE or f.z_dir0 == 0.0
    return (SPOT & ~abi.INTERSECT_OBJ) | (0 if wide else abi.INTERSECT_OBJ)


def _opts(wl, fi, mode, fast, **kw):
    from rayoptics_amd.engine import make_opts
    return make_opts(flags=_flags(wl, fi) | (abi.FAST_FP64 if fast else 0), out_mode=mode, first_surf=1,
                     last_surf=wl.n_ifcs - 2, foc=wl.foc, image_pt=wl.image_pts[fi], **kw)


def check_flips(tbl, wi, opts, ref, got, full_of, what, on_boundary=None):
    """ref / got: exact and tolerance-mode results of the same rays.  Returns (n_flips, worst
    scaled error of the rays both paths treat alike).  full_of(r) -> the oracle's FULL packet
    [n_seg, 10] of ray r.  on_boundary: rays known to sit within 1e-10 of a decision boundary
    although both paths decide them alike (a ray AT the critical angle leaves along the
    surface: its image intercept is ~1e8 and moves by 1e7 per ulp of the radicand) -- their
    status is compared, their values are not."""
    flip = (ref.status != got.status) | (ref.fail_surf != got.fail_surf)
    same = ~flip
    if on_boundary is not None:
        same &= ~on_boundary
    ok = same & (ref.status == abi.OK)
    err = 0.0
    if ok.any():
        err = max(err, H.scaled_err(ref.seg[..., ok], got.seg[..., ok]))
    if same.any():
        err = max(err, H.scaled_err(ref.op[same], got.op[same]))
    assert err <= TOL, f'{what}: scaled error {err:.3e} > {TOL}'
    for r in np.flatnonzero(flip):
        # decided differently: at the first interface either path stopped at
        surfs = [int(s) for s in (ref.fail_surf[r], got.fail_surf[r]) if s >= 0]
        assert surfs, (what, r)
        k = min(surfs)
        m = H.boundary_margin(tbl, wi, opts, full_of(int(r)), k)
        assert m and min(m.values()) <= TOL, f'{what}: ray {r} flipped at interface {k}, margins {m}'
    return int(flip.sum()), err


def _full_packet_fn(wl, fi, wi, grid_def, opts):
    """oracle FULL packet of ray r of a product grid (traces the one pupil row that holds it)"""
    from oracle import oracle
    start, stop, num = grid_def

    def full_of(r):
        g = oracle.make_grid(start, stop, num, row_begin=r // num, row_count=1)
        o = oracle.make_opts(flags=opts.flags & ~abi.FAST_FP64, out_mode=abi.OUT_FULL,
                             first_surf=opts.first_surf, last_surf=opts.last_surf, eps=opts.eps, fuzz=opts.fuzz)
        res = oracle.trace_pupil_grid(wl.table, wl.fields[fi], g, wi, o)
        return res.seg[:, :, r % num]
    return full_of


@pytest.mark.parametrize('name', WORKLOADS)
def test_fast_modes_against_the_oracle(name):
    """64 x 64 grids of every field (first and last wavelength): HITS, LAST and packed hits in
    tolerance mode against oracle/rox_oracle.c"""
    from oracle import oracle
    from rayoptics_amd import workloads
    from rayoptics_amd.engine import TraceEngine, make_grid
    wl = workloads.load(name)
    eng = TraceEngine(wl.table)
    num = 64
    gdef = ((-1., -1.), (1., 1.), num)
    grid = make_grid(*gdef)
    W = len(wl.table.wvls)
    worst, flips, n_ok = 0.0, 0, 0
    for fi in range(len(wl.fields)):
        for wi in sorted({0, W - 1}):
            for mode in (abi.OUT_HITS, abi.OUT_LAST):
                o = _opts(wl, fi, mode, True)
                orc = oracle.trace_pupil_grid(wl.table, wl.fields[fi], grid, wi, o)
                dev = eng.trace_pupil_grid(wl.fields[fi], grid, wi, o, nan_fill=True).to_host()
                f, e = check_flips(wl.table, wi, o, orc, dev, _full_packet_fn(wl, fi, wi, gdef, o),
                                   f'{name} f{fi} w{wi} mode {mode}')
                flips += f
                worst = max(worst, e)
                n_ok += int((orc.status == abi.OK).sum())
            oc = _opts(wl, fi, abi.OUT_HITS_COMPACT, True)
            xy = eng.trace_pupil_grid_hits(wl.fields[fi], grid, wi, oc)
            orc_c = oracle.trace_pupil_grid(wl.table, wl.fields[fi], grid, wi, oc)
            if f == 0:
                assert xy.shape == orc_c.hits.shape
                assert H.scaled_err(orc_c.hits, xy) <= TOL
    eng.close()
    assert n_ok > 500
    H.record('fast_vs_oracle', workload=name, grid=num, worst_scaled_error=worst, status_flips=flips, rays_through=n_ok)


FULL_SIZE = [('dblgauss_c2', 1024, [0, 1, 2]), ('zmx_evenasph_c3', 512, [0, 1, 2]), ('nikkor_c3', 512, [0, 2]),
             ('rc_telescope_c4', 256, [0, 2, 4]), ('litho_c5', 512, [0, 4, 8]), ('cell_phone', 512, [0, 2])]


@pytest.mark.parametrize('name,num,fields', FULL_SIZE)
def test_fast_hits_at_baseline_sizes_against_the_exact_device_path(name, num, fields):
    """the BASELINE configurations' own grids: tolerance-mode HITS vs the bit-exact HITS of the same
    launch (itself equal to the oracle bit for bit: tests/test_gpu_timed_launch.py,
    test_gpu_configs.py), every ray"""
    from rayoptics_amd import workloads
    from rayoptics_amd.engine import TraceEngine, make_grid
    wl = workloads.load(name)
    eng = TraceEngine(wl.table)
    gdef = ((-1., -1.), (1., 1.), num)
    grid = make_grid(*gdef)
    worst, flips, rays = 0.0, 0, 0
    for fi in fields:
        wi = wl.ref_wvl_idx
        ox, of = _opts(wl, fi, abi.OUT_HITS, False), _opts(wl, fi, abi.OUT_HITS, True)
        ex = eng.trace_pupil_grid(wl.fields[fi], grid, wi, ox, nan_fill=True).to_host()
        fa = eng.trace_pupil_grid(wl.fields[fi], grid, wi, of, nan_fill=True).to_host()
        f, e = check_flips(wl.table, wi, of, ex, fa, _full_packet_fn(wl, fi, wi, gdef, of), f'{name} f{fi}')
        flips += f
        worst = max(worst, e)
        rays += num * num
        assert (ex.status == abi.OK).sum() > num
    eng.close()
    H.record('fast_vs_exact_device', workload=name, grid=num, fields=len(fields), rays=rays,
             worst_scaled_error=worst, status_flips=flips)
    assert flips <= rays // 10000


from test_oracle_golden import OPD_CASES, opd_opts  # noqa: E402


@pytest.mark.parametrize('name,case', OPD_CASES)
def test_fast_opd_and_fan(name, case):
    """ROX_OUT_OPD grids and ROX_OUT_FAN fans in tolerance mode against the oracle: OPD within
    1e-10 system units of optical path (~2e-7 waves), finite and infinite reference spheres"""
    from oracle import oracle
    from rayoptics_amd.engine import TraceEngine
    fx = H.fixture(name)
    c = fx[case]
    eng = TraceEngine(fx.table)
    fld = H.field_from_arr(c['field'])
    wi = int(c['wvl_idx'])
    o = opd_opts(c)
    o.flags |= abi.FAST_FP64
    grid = oracle.make_grid(c['start'], c['stop'], 96)
    orc = oracle.trace_pupil_grid(fx.table, fld, grid, wi, o)
    dev = eng.trace_pupil_grid(fld, grid, wi, o, nan_fill=True).to_host()
    same = (orc.status == dev.status)
    assert (~same).sum() <= 2, int((~same).sum())
    ok = same & (orc.status == abi.OK)
    assert ok.sum() > 100
    worst = H.scaled_err(orc.seg[..., ok], dev.seg[..., ok])
    assert worst <= TOL, worst
    for xy in (0, 1):
        start, stop = np.zeros(2), np.zeros(2)
        start[xy], stop[xy] = -1.0, 1.0
        g = oracle.make_grid(start, stop, 33, abi.GRID_FAN)
        of = opd_opts(c)
        of.out_mode = abi.OUT_FAN
        of.flags |= abi.APPLY_VIGNETTING | abi.FAST_FP64
        of.foc, of.image_pt[0], of.image_pt[1] = 0.02, 0.01, -0.03
        orc = oracle.trace_pupil_grid(fx.table, fld, g, wi, of)
        dev = eng.trace_pupil_grid(fld, g, wi, of, nan_fill=True).to_host()
        np.testing.assert_array_equal(dev.status, orc.status)
        okf = orc.status == abi.OK
        worst = max(worst, H.scaled_err(orc.seg[..., okf], dev.seg[..., okf]))
    assert worst <= TOL, worst
    eng.close()
    H.record('fast_opd_fan_vs_oracle', fixture=name, case=case, worst_scaled_error=worst)


def check_full_packets(tbl, wi, opts, ref, got, what):
    """FULL packets in tolerance mode against the exact ones: every ray both paths end alike --
    through, or stopped at the same interface for the same reason -- has the same segments written
    (the NaN pattern of a NaN-filled buffer) and every value of them within TOL, the partial
    record of a blocked / reflected ray included; flips are accounted for as check_flips does"""
    def full_of(r):
        return ref.seg[:, :, r]
    flips, err = check_flips(tbl, wi, opts, ref, got, full_of, what)
    same = (ref.status == got.status) & (ref.fail_surf == got.fail_surf)
    if same.any():
        err = max(err, H.scaled_err(ref.seg[..., same], got.seg[..., same]))
        if ref.pupil is not None and got.pupil is not None:
            H.bit_equal(ref.pupil, got.pupil, what + ' pupil')
    assert err <= TOL, f'{what}: scaled error {err:.3e} (partial packets included)'
    return flips, err


@pytest.mark.parametrize('name', WORKLOADS)
def test_full_packets_in_tolerance_mode_against_the_oracle(name):
    """ROX_OUT_FULL with ROX_FAST_FP64: 48 x 48 grids of every field (first and last wavelength),
    apertures checked -- so that blocked rays leave their partial records -- against the oracle's
    packets; the single launch and the batched one give the same bits"""
    from oracle import oracle
    from rayoptics_amd import workloads
    from rayoptics_amd.engine import TraceEngine, make_grid
    wl = workloads.load(name)
    eng = TraceEngine(wl.table)
    num = 48
    grid = make_grid((-1., -1.), (1., 1.), num)
    W = len(wl.table.wvls)
    worst, flips, n_ok, n_part = 0.0, 0, 0, 0
    pairs = [(fi, wi) for fi in range(len(wl.fields)) for wi in sorted({0, W - 1})]
    ol = [_opts(wl, fi, abi.OUT_FULL, True) for fi, _ in pairs]
    batch = eng.trace_pupil_grids([wl.fields[fi] for fi, _ in pairs], [wi for _, wi in pairs], grid, ol,
                                  nan_fill=True)
    for (fi, wi), o, rb in zip(pairs, ol, batch):
        orc = oracle.trace_pupil_grid(wl.table, wl.fields[fi], grid, wi, o)
        dev = eng.trace_pupil_grid(wl.fields[fi], grid, wi, o, nan_fill=True).to_host()
        f, e = check_full_packets(wl.table, wi, o, orc, dev, f'{name} f{fi} w{wi} FULL')
        flips += f
        worst = max(worst, e)
        n_ok += int((orc.status == abi.OK).sum())
        n_part += int(((orc.status == abi.BLOCKED) | (orc.status == abi.TIR)).sum())
        b = rb.to_host()
        np.testing.assert_array_equal(dev.status, b.status)
        H.bit_equal(dev.seg, b.seg, f'{name} f{fi} w{wi}: batched vs single, tolerance-mode FULL')
    eng.close()
    assert n_ok > 300
    H.record('fast_full_vs_oracle', workload=name, grid=num, worst_scaled_error=worst, status_flips=flips,
             rays_through=n_ok, partial_records=n_part)


def test_full_tolerance_kernels_on_every_workload():
    """The host sends a FULL launch to the tolerance-mode kernels only for systems made mostly of
    aspheres (roxtrace.hip use_fast(); of the workloads above: the phone lens) -- elsewhere they
    are slower than the store-bound exact kernels.  ROX_FAST_FP64_FULL=1 (read once by the
    library: a subprocess) sends every FULL launch with the flag there: the test above, on every
    workload, against the same bar; and the phone lens indeed runs on them by default (its
    packets differ from the exact ones in the last bits; the double Gauss's do not)"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, ROX_FAST_FP64_FULL='1')
    p = subprocess.run([sys.executable, '-m', 'pytest', '-x', '-q', '-m', 'gpu', '-p', 'no:cacheprovider',
                        'tests/test_gpu_fast.py', '-k', 'full_packets_in_tolerance_mode'],
                       env=env, cwd=root, capture_output=True, text=True, timeout=1800)
    assert p.returncode == 0, p.stdout[-4000:] + p.stderr[-2000:]
    assert f'{len(WORKLOADS)} passed' in p.stdout, p.stdout[-500:]
    from rayoptics_amd import workloads
    from rayoptics_amd.engine import TraceEngine, make_grid
    grid = make_grid((-1., -1.), (1., 1.), 40)
    differs = {}
    for name in ('cell_phone', 'dblgauss_c2'):
        wl = workloads.load(name)
        eng = TraceEngine(wl.table)
        a = eng.trace_pupil_grid(wl.fields[0], grid, 0, _opts(wl, 0, abi.OUT_FULL, True), nan_fill=True).to_host()
        b = eng.trace_pupil_grid(wl.fields[0], grid, 0, _opts(wl, 0, abi.OUT_FULL, False), nan_fill=True).to_host()
        differs[name] = not np.array_equal(a.seg, b.seg, equal_nan=True)
        assert H.scaled_err(b.seg[..., b.status == abi.OK], a.seg[..., b.status == abi.OK]) <= TOL
        eng.close()
    assert differs == {'cell_phone': True, 'dblgauss_c2': False}


def test_full_packets_under_phantom_filtering_stay_exact_and_batches_must_agree():
    """ROX_FILTER_PHANTOMS + ROX_OUT_FULL keeps the exact kernels whatever ROX_FAST_FP64 says (the
    late append of a filtered segment, raytrace.py:185-191, is not in trace_ray_fast); a batch must
    agree on the flag"""
    from oracle import oracle
    from rayoptics_amd import workloads
    from rayoptics_amd.engine import TraceEngine, make_grid, EngineError, make_opts
    grid = make_grid((-1., -1.), (1., 1.), 96)
    # (the phone lens: the system whose FULL launches the flag would otherwise move)
    wl = workloads.load('cell_phone')
    eng = TraceEngine(wl.table)
    o = make_opts(flags=_flags(wl, 1) | abi.FAST_FP64 | abi.FILTER_PHANTOMS, out_mode=abi.OUT_FULL,
                  first_surf=1, last_surf=wl.n_ifcs - 2, foc=wl.foc, image_pt=wl.image_pts[1])
    dev = eng.trace_pupil_grid(wl.fields[1], grid, 1, o, nan_fill=True).to_host()
    orc = oracle.trace_pupil_grid(wl.table, wl.fields[1], grid, 1, o)
    np.testing.assert_array_equal(dev.status, orc.status)
    H.bit_equal(dev.seg, orc.seg, 'FULL with ROX_FAST_FP64 | ROX_FILTER_PHANTOMS')
    H.bit_equal(dev.op, orc.op, 'op')
    eng.close()
    wl = workloads.load('dblgauss_c2')
    eng = TraceEngine(wl.table)
    with pytest.raises(EngineError, match='ROX_FAST_FP64'):
        eng.trace_pupil_grids([wl.fields[0], wl.fields[1]], [0, 0], grid,
                              [_opts(wl, 0, abi.OUT_HITS, True), _opts(wl, 1, abi.OUT_HITS, False)])
    # a batch in tolerance mode == the single launches in tolerance mode (same kernels' arithmetic)
    pairs = [(fi, wi) for fi in range(3) for wi in range(len(wl.table.wvls))]
    ol = [_opts(wl, fi, abi.OUT_HITS, True) for fi, _ in pairs]
    res = eng.trace_pupil_grids([wl.fields[fi] for fi, _ in pairs], [wi for _, wi in pairs], grid, ol,
                                nan_fill=True)
    for (fi, wi), oo, r in zip(pairs, ol, res):
        one = eng.trace_pupil_grid(wl.fields[fi], grid, wi, oo, nan_fill=True).to_host()
        b = r.to_host()
        np.testing.assert_array_equal(one.status, b.status)
        H.bit_equal(one.seg, b.seg, f'batched fast f{fi} w{wi}')
    eng.close()


def _edge_table(n_glass, max_ap):
    """object | flat glass-to-air interface 2 mm behind it | image: rays parallel to the axis land
    where they start (aperture edges), tilted rays meet the critical angle"""
    from rayoptics_amd import SurfaceTable
    return SurfaceTable.from_prescription(
        [dict(cv=0., thi=2., n=n_glass, max_aperture=1e12),
         dict(cv=0., thi=3., n=1.0, max_aperture=max_ap),
         dict(cv=0., thi=0., n=1.0, max_aperture=1e12)], wvls=(550.,))


def _ulps(x, k):
    for _ in range(abs(k)):
        x = np.nextafter(x, np.inf if k > 0 else -np.inf)
    return x


def test_flipped_rays_lie_on_their_boundaries():
    """rays laid within +-8 ulp of an aperture edge and of the critical angle: tolerance mode may
    decide them the other way (it forms x^2 + y^2 with one rounding and the TIR radicand from
    mu = n_in / n_out); every such ray is shown to sit on the boundary it was decided at, the
    others agree to 1e-10"""
    from oracle import oracle
    from rayoptics_amd.engine import TraceEngine
    n_glass, max_ap, fuzz = 1.5, 4.3, 1e-5
    tbl = _edge_table(n_glass, max_ap)
    eng = TraceEngine(tbl)
    pts, dirs = [], []
    t = max_ap + fuzz
    for ang in np.linspace(0.0, 2 * np.pi, 181):            # the aperture edge
        for k in range(-8, 9):
            pts.append((_ulps(t * np.cos(ang), k), t * np.sin(ang), 0.0))
            dirs.append((0.0, 0.0, 1.0))
    sc = 1.0 / n_glass                                       # sin of the critical angle
    for ang in np.linspace(0.0, 2 * np.pi, 181):            # the TIR limit
        for k in range(-8, 9):
            s_ = _ulps(sc, k)
            dirs.append((s_ * np.cos(ang), s_ * np.sin(ang), np.sqrt(1.0 - s_ * s_)))
            pts.append((0.1 * np.cos(ang), 0.05, 0.0))
    pt0, d = np.array(pts).T.copy(), np.array(dirs).T.copy()
    R = pt0.shape[1]
    flags = abi.INTERSECT_OBJ | abi.CHECK_APERTURES
    o_full = oracle.make_opts(flags=flags, out_mode=abi.OUT_FULL, first_surf=1, last_surf=1, fuzz=fuzz)
    full = oracle.trace_rays(tbl, pt0, d, 0, o_full)
    # the rays of the second group graze the TIR limit by construction (margin ~1e-15): shown here
    near = np.zeros(R, dtype=bool)
    for r in range(R // 2, R):
        m = H.boundary_margin(tbl, 0, o_full, full.seg[:, :, r], 1)
        near[r] = m.get('tir', 1.0) <= TOL
    assert near[R // 2:].all() and not near[:R // 2].any()
    total_flips = 0
    for mode in (abi.OUT_HITS, abi.OUT_LAST):
        o = oracle.make_opts(flags=flags | abi.FAST_FP64, out_mode=mode, first_surf=1, last_surf=1, fuzz=fuzz)
        orc = oracle.trace_rays(tbl, pt0, d, 0, o)
        dev = eng.trace_rays(pt0, d, 0, o, nan_fill=True).to_host()
        f, e = check_flips(tbl, 0, o, orc, dev, lambda r: full.seg[:, :, r], f'edge rays mode {mode}',
                           on_boundary=near)
        total_flips += f
        assert 0 < int((orc.status == abi.BLOCKED).sum()) < R // 2
        assert 0 < int((orc.status == abi.TIR).sum()) < R // 2
    eng.close()
    H.record('fast_edge_rays', rays=2 * R, status_flips=total_flips, every_flip_on_its_boundary=True)
