"""Pins the CPU oracle (oracle/rox_oracle.c) against
  (a) the reference's own golden vectors / known answers, and
  (b) fixtures produced by running the reference itself
      (tests/golden/make_golden.py).
CPU only: this is the `-m "not gpu"` half of the parity story; the `-m gpu`
tests then hold the HIP kernels to the oracle bit for bit."""
import ctypes as C

import numpy as np
import pytest

from rayoptics_amd import abi
from oracle import oracle
import helpers as H


def test_codev_marginal_ray_kat():
    """rayoptics/raytr/tests/test_sequential.py:34-77 with
    rayoptics/raytr/tests/marginal_ray.py (CODE V, 6 decimals)"""
    fx = H.fixture('dblgauss_seq')
    c = fx['marginal']
    res = oracle.trace_rays(fx.table, c['pt0'], c['dir0'], 0, oracle.make_opts())
    assert res.status[0] == abi.OK
    codev = c['codev']          # rows: x y z tanX tanY dist
    seg = res.seg[:, :, 0]
    for i in range(13):
        np.testing.assert_allclose(seg[i, 0:3], codev[i, 0:3], rtol=3e-6, atol=2e-6)
        np.testing.assert_allclose([seg[i, 3] / seg[i, 5], seg[i, 4] / seg[i, 5]],
                                   codev[i, 3:5], rtol=3e-6, atol=1e-6)
        if 1 < i < 12:      # test_sequential.py:73 skips object/image space
            np.testing.assert_allclose(seg[i, 6], codev[i, 5], rtol=3e-6)
    # and bit-for-bit against what the reference code itself produced
    assert H.assert_result_matches(c, res, require_exact=True) == 1.0


def test_profiles_kat():
    """rayoptics/elem/tests/test_profiles.py:127-154: double-Gauss S1,
    Spherical closed form vs Spencer-Murty Newton, s = 5.866433424372758"""
    r1 = 56.20238                       # test_profiles.py:128-133
    cv = 1 / r1
    p0 = np.array([0., 25.0, 0.])
    d = np.array([0., 0., 1.])
    sag1 = r1 - np.sqrt(r1 * r1 - 25.0 * 25.0)
    L = oracle.lib()
    out = {}
    for prof in (abi.SPHERICAL, abi.EVENPOLY):
        sf = abi.Surface()
        sf.profile, sf.cv, sf.cc, sf.ec = prof, cv, 0.0, 1.0
        s = C.c_double()
        p1 = np.zeros(3)
        nrm = np.zeros(3)
        st = L.rox_oracle_intersect(C.byref(sf), p0.ctypes.data, d.ctypes.data,
                                    1e-12, 1.0, C.byref(s), p1.ctypes.data,
                                    nrm.ctypes.data)
        assert st == 0
        out[prof] = (s.value, p1, nrm)
    np.testing.assert_allclose(out[abi.SPHERICAL][0], 5.866433424372758, rtol=1e-14)
    np.testing.assert_allclose(out[abi.SPHERICAL][1], [0., 25.0, sag1], rtol=1e-14)
    nrm_truth = -(np.array([0., 25.0, sag1]) - np.array([0, 0, r1]))
    nrm_truth /= np.linalg.norm(nrm_truth)
    np.testing.assert_allclose(out[abi.SPHERICAL][2], nrm_truth, rtol=1e-14)
    np.testing.assert_allclose(out[abi.EVENPOLY][0], 5.866433424372758, rtol=1e-13)
    np.testing.assert_allclose(out[abi.SPHERICAL][1], out[abi.EVENPOLY][1], rtol=1e-13, atol=1e-13)
    np.testing.assert_allclose(out[abi.SPHERICAL][2], out[abi.EVENPOLY][2], rtol=1e-13, atol=1e-13)


RAY_CASES = [('dblgauss_seq', 'bundle'), ('dblgauss', 'rays_ap'),
             ('dblgauss', 'rays_noap'), ('dblgauss_finite', 'rays_ap'),
             ('rc_telescope', 'rays_ap'), ('nikkor', 'rays_ap'),
             ('cell_phone', 'rays_ap'), ('tilted_singlet', 'rays_ap'),
             ('toroid_lens', 'rays_ap')]


@pytest.mark.parametrize('name,case', RAY_CASES)
def test_explicit_rays_vs_reference(name, case):
    fx = H.fixture(name)
    c = fx[case]
    wi = c['wvl_idx'] if 'wvl_idx' in c else 0
    res = oracle.trace_rays(fx.table, c['pt0'], c['dir0'], wi, H.make_opts(c))
    exact = H.assert_result_matches(c, res)
    # the FMA-chain model of NumPy's BLAS sites makes the restatement bit-exact,
    # tilted / decentered systems included (per-row dgemv summation order)
    if True:
        assert exact == 1.0, f'{name}/{case}: bit-exact fraction {exact}'
    assert len(set(c['status'].tolist())) >= 1


GRID_CASES = [('dblgauss', 'grid_f0'), ('dblgauss', 'grid_f2'), ('dblgauss', 'fan_f1'),
              ('dblgauss_finite', 'grid_f2'), ('singlet', 'grid64'),
              ('singlet', 'grid_f1'), ('rc_telescope', 'grid_f0'),
              ('rc_telescope', 'grid_f4'), ('nikkor', 'grid_f1'),
              ('cell_phone', 'grid_f2'), ('tilted_singlet', 'grid_f1'),
              ('toroid_lens', 'grid_f1')]


@pytest.mark.parametrize('name,case', GRID_CASES)
def test_pupil_grid_vs_reference_driver(name, case):
    """trace.trace_grid / trace_fan -> trace_safe -> trace_base ->
    apply_vignetting + ray_start_from_osp -> rt.trace, all reference code"""
    fx = H.fixture(name)
    c = fx[case]
    fld = H.field_from_arr(c['field'])
    grid = oracle.make_grid(c['start'], c['stop'], int(c['num']), int(c['kind']))
    res = oracle.trace_pupil_grid(fx.table, fld, grid, int(c['wvl_idx']), H.make_opts(c))
    np.testing.assert_array_equal(res.pupil, c['pupil'])    # accumulate-by-step + vignetting
    exact = H.assert_result_matches(c, res)
    assert exact == 1.0, f'{name}/{case}: bit-exact fraction {exact}'


@pytest.mark.parametrize('name', ['dblgauss', 'singlet', 'rc_telescope', 'nikkor'])
def test_spot_diagram_vs_reference_figure(name):
    """HITS output == what SpotDiagramFigure.update_data() computed"""
    fx = H.fixture(name)
    c = fx['spot']
    num = int(c['num'])
    N = fx.table.n_ifcs
    nchecked = 0
    for key in c:
        if not key.endswith('_hits'):
            continue
        fi, wi = key.split('_')[0], int(key.split('_')[1][1:])
        fld = H.field_from_arr(c[f'{fi}_field'])
        opts = oracle.make_opts(flags=abi.INTERSECT_OBJ | abi.CHECK_APERTURES | abi.APPLY_VIGNETTING,
                                out_mode=abi.OUT_HITS, first_surf=1, last_surf=N - 2,
                                foc=float(c['foc']), image_pt=tuple(c[f'{fi}_image_pt']))
        grid = oracle.make_grid((-1., -1.), (1., 1.), num)
        res = oracle.trace_pupil_grid(fx.table, fld, grid, wi, opts)
        ok = res.status == abi.OK
        got = res.seg[:, ok].T          # form='list', append_if_none=False
        exp = c[key]
        assert got.shape == exp.shape, (key, got.shape, exp.shape)
        H.assert_soa_close(exp, got, key, require_exact=True)
        nchecked += 1
    assert nchecked >= 2


def test_list_of_rays_last_segment():
    """analyses.trace_list_of_rays(output_filter='last')"""
    fx = H.fixture('dblgauss')
    c = fx['list_last']
    N = fx.table.n_ifcs
    opts = oracle.make_opts(flags=abi.INTERSECT_OBJ | abi.CHECK_APERTURES,
                            out_mode=abi.OUT_LAST, first_surf=1, last_surf=N - 2)
    res = oracle.trace_rays(fx.table, c['pt0'], c['dir0'], c['wvl_idx'], opts)
    np.testing.assert_array_equal(res.status, c['status'])
    H.assert_soa_close(c['last'], res.seg, 'last', require_exact=True)
    H.assert_soa_close(c['op'], np.where(res.status == 0, res.op, np.nan), 'op', require_exact=True)


OPD_CASES = [('dblgauss', 'opd_f0'), ('dblgauss', 'opd_f2'), ('rc_telescope', 'opd_f3'),
             ('nikkor', 'opd_f1'), ('tilted_singlet', 'opd_f1'), ('toroid_lens', 'opd_f1'),
             # infinite reference sphere: wave_abr_full_calc_inf_ref (waveabr.py:356-424)
             ('telecentric', 'opd_f0'), ('telecentric', 'opd_f2')]


def opd_opts(c):
    from rayoptics_amd.table import wavefront_from_array
    return oracle.make_opts(flags=int(c['flags']), out_mode=abi.OUT_OPD,
                            first_surf=int(c['first_surf']), last_surf=int(c['last_surf']),
                            wf=wavefront_from_array(c['wavefront']))


def check_opd_grid(c, res, exact):
    """res: HostResult-like (seg[1][R], status, pupil) vs analyses.eval_wavefront"""
    num = int(c['num'])
    exp = c['opd_grid']                  # [num][num][3] = px, py, opd in waves (nan if no ray)
    got = np.where(res.status == 0, float(c['convert_to_opd']) * res.seg[0], np.nan).reshape(num, num)
    np.testing.assert_array_equal(res.pupil[0].reshape(num, num), exp[:, :, 0])
    np.testing.assert_array_equal(res.pupil[1].reshape(num, num), exp[:, :, 1])
    assert np.array_equal(np.isnan(got), np.isnan(exp[:, :, 2]))
    m = ~np.isnan(got)
    assert m.sum() > num
    # OPD is a difference of optical paths of O(100) system units: 1e-10
    # absolute on the path lengths is ~2e-7 waves
    assert np.abs(got[m] - exp[:, :, 2][m]).max() <= 1e-10 * float(c['convert_to_opd'])
    frac = float(np.mean(got[m] == exp[:, :, 2][m]))
    if exact:
        assert frac == 1.0, frac
    return frac


@pytest.mark.parametrize('name,case', OPD_CASES)
def test_opd_vs_reference_eval_wavefront(name, case):
    """OUT_OPD == analyses.eval_wavefront -> waveabr.wave_abr_full_calc"""
    fx = H.fixture(name)
    c = fx[case]
    fld = H.field_from_arr(c['field'])
    grid = oracle.make_grid(c['start'], c['stop'], int(c['num']))
    res = oracle.trace_pupil_grid(fx.table, fld, grid, int(c['wvl_idx']), opd_opts(c))
    check_opd_grid(c, res, exact=True)


# docs/source/examples/Cell_Phone_lens/Cell_Phone_lens.rst:311-328 -- list_ray of
# rt.trace(sm, [0,1,0], [0,0,1], central wvl) through the 8-asphere
# RadialPolynomial phone lens (rayoptics/optical/tests/cell_phone_camera.roa):
# rows = Y, Z, M, N, Len as printed (X = L = 0)
CELL_PHONE_DOC_MARGINAL = np.array([
    [1.00000, 0.0, 0.000000, 1.000000, 1e10],
    [1.00000, 0.0, 0.000000, 1.000000, 0.26119],
    [1.00000, 0.26119, -0.163284, 0.986579, 0.93632],
    [0.84711, -0.0050525, -0.272278, 0.962219, 0.86687],
    [0.61108, -0.10094, -0.024063, 0.999710, 0.79796],
    [0.59188, -0.053212, -0.171810, 0.985130, 0.16841],
    [0.56295, 0.012694, -0.122925, 0.992416, 0.89598],
    [0.45281, 0.01188, -0.158261, 0.987397, 0.2017],
    [0.42089, 0.051033, -0.178956, 0.983857, 0.83614],
    [0.27126, 0.023675, -0.185004, 0.982738, 0.6882],
    [0.14394, 0.0, -0.122034, 0.992526, 0.40301],
    [0.09476, 0.0, -0.185004, 0.982738, 0.65124],
    [-0.02573, 0.0, -0.185004, 0.982738, 0.0]])


def test_cell_phone_doc_table_kat():
    """the reference's documentation prints this ray to 5-6 digits: a known
    answer for the Newton (Spencer-Murty) path on RadialPolynomial aspheres"""
    fx = H.fixture('cell_phone')
    N = fx.table.n_ifcs
    assert N == 13
    wi = fx.table.wvls.index(587.5618) if 587.5618 in fx.table.wvls else 1
    pt0 = np.array([[0.], [1.], [0.]])
    dir0 = np.array([[0.], [0.], [1.]])
    res = oracle.trace_rays(fx.table, pt0, dir0, wi,
                            oracle.make_opts(flags=abi.INTERSECT_OBJ, first_surf=1, last_surf=N - 2))
    assert res.status[0] == abi.OK
    seg = res.seg[:, :, 0]
    doc = CELL_PHONE_DOC_MARGINAL
    np.testing.assert_allclose(seg[:, 0], 0.0, atol=1e-15)
    np.testing.assert_allclose(seg[:, 1], doc[:, 0], atol=6e-6)
    np.testing.assert_allclose(seg[:, 2], doc[:, 1], atol=6e-6, rtol=6e-5)
    np.testing.assert_allclose(seg[:, 4], doc[:, 2], atol=6e-7)
    np.testing.assert_allclose(seg[:, 5], doc[:, 3], atol=6e-7)
    np.testing.assert_allclose(seg[1:, 6], doc[1:, 4], rtol=6e-5, atol=6e-6)
