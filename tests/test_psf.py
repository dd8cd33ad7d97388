"""analyses.calc_psf (rayoptics/raytr/analyses.py:848-875): the oracle's restatement
against PSFs the reference itself computed (tests/golden/psf.npz, made by
tests/golden/make_golden.py --only-psf), the drop-in's host logic, and -- on the
GPU -- rox_calc_psf (pruned DFT on the fp64 matrix cores) against both.

Tolerance: the PSF is normalised to a peak of 1; the reference goes through
pocketfft, the restatements through a direct DFT, so agreement is to rounding of
the transform (a few 1e-15 of the peak), asserted at 1e-12."""
import ctypes as C
import os

import numpy as np
import pytest

from rayoptics_amd import abi

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'psf.npz')
ATOL = 1e-12


def cases():
    z = np.load(GOLDEN)
    names = sorted(set(k.split('/')[0] for k in z.files if '/' in k))
    return [(n, z[f'{n}/opd'], int(z[f'{n}/dims'][0]), int(z[f'{n}/dims'][1]), z[f'{n}/psf'])
            for n in names]


CASES = cases()


@pytest.mark.parametrize('name,opd,ndim,maxdim,psf', CASES, ids=[c[0] for c in CASES])
def test_oracle_psf_matches_the_reference(name, opd, ndim, maxdim, psf):
    from oracle import oracle
    got = oracle.calc_psf(opd, ndim, maxdim)
    assert got.shape == psf.shape == (maxdim, maxdim)
    assert got.max() == 1.0
    np.testing.assert_allclose(got, psf, rtol=0, atol=ATOL)


def test_oracle_psf_rejects_what_the_reference_rejects():
    from oracle import oracle
    w = np.zeros((7, 7))
    with pytest.raises(ValueError):
        oracle.calc_psf(w, 7, 32)               # odd ndim: the slice shapes differ
    with pytest.raises(ValueError):
        oracle.calc_psf(np.zeros((8, 8)), 8, 8)  # block does not fit


@pytest.mark.needs_reference
def test_calc_psf_dropin_against_the_live_reference():
    """install() rebinds analyses.calc_psf / update_psf_data; with the oracle standing in
    for the device the reference's own callers get the reference's PSF"""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), 'golden'))
    import refmodels as rm  # noqa: F401
    import rayoptics.raytr.analyses as ranalyses
    from oracle import oracle
    from oracle_engine import OracleEngine
    from rayoptics_amd import analyses, install, session
    session._set_engine_factory(OracleEngine)
    analyses.PSF_BACKEND = oracle.calc_psf
    opm = rm.dblgauss()
    fld = opm['osp']['fov'].fields[1]
    try:
        theirs_grid = ranalyses.RayGrid(opm, f=fld, wl=587.6, num_rays=16)
        theirs_grid.maxdim = 48
        theirs = ranalyses.update_psf_data(theirs_grid)
        install.install()
        ours_grid = ranalyses.RayGrid(opm, f=fld, wl=587.6, num_rays=16)
        ours_grid.maxdim = 48
        ours = ranalyses.update_psf_data(ours_grid)
        direct = ranalyses.calc_psf(ours_grid.grid[2], 16, 48)
    finally:
        install.uninstall()
        session._set_engine_factory(None)
        analyses.PSF_BACKEND = None
    assert ours.shape == theirs.shape == (48, 48)
    np.testing.assert_allclose(ours, theirs, rtol=0, atol=ATOL)
    np.testing.assert_array_equal(direct, ours)


# ------------------------------------------------------------------ on the device
@pytest.mark.gpu
@pytest.mark.parametrize('name,opd,ndim,maxdim,psf', CASES, ids=[c[0] for c in CASES])
def test_device_psf(name, opd, ndim, maxdim, psf):
    from oracle import oracle
    from rayoptics_amd import analyses
    got = analyses.calc_psf(opd, ndim, maxdim)              # host arrays through ROX_HOST_POINTERS
    assert got.shape == (maxdim, maxdim) and got.max() == 1.0
    np.testing.assert_allclose(got, psf, rtol=0, atol=ATOL)             # the reference
    np.testing.assert_allclose(got, oracle.calc_psf(opd, ndim, maxdim), rtol=0, atol=ATOL)


@pytest.mark.gpu
def test_device_psf_resident_and_large():
    """device-pointer form (the OPD grid of a ROX_OUT_OPD launch never leaves HBM),
    sizes that are not multiples of the GEMM tiles, argument errors"""
    import torch
    from oracle import oracle
    from rayoptics_amd.engine import calc_psf, load_library
    rng = np.random.default_rng(3)
    for ndim, maxdim in ((2, 5), (10, 17), (34, 129), (128, 512), (250, 1000)):
        y, x = np.mgrid[-1:1:ndim * 1j, -1:1:ndim * 1j]
        opd = 1.5 * (x * x + y * y) + 0.4 * x * y * y + 0.05 * rng.standard_normal((ndim, ndim))
        if ndim >= 8:
            opd[x * x + y * y > 1.0] = np.nan
        dev = calc_psf(torch.from_numpy(opd).cuda(), ndim, maxdim)
        assert dev.is_cuda and tuple(dev.shape) == (maxdim, maxdim)
        got = dev.cpu().numpy()
        host = calc_psf(opd, ndim, maxdim)
        np.testing.assert_array_equal(got, host)
        if maxdim <= 512:
            np.testing.assert_allclose(got, oracle.calc_psf(opd, ndim, maxdim), rtol=0, atol=ATOL)
        # size-independent properties: peak 1, non-negative
        assert got.max() == 1.0 and got.min() >= 0.0
    lib = load_library()
    buf = np.zeros((8, 8))
    out = np.zeros((8, 8))
    assert lib.rox_calc_psf(buf.ctypes.data, 7, 32, out.ctypes.data, abi.HOST_POINTERS, None) == -1
    assert lib.rox_calc_psf(buf.ctypes.data, 8, 8, out.ctypes.data, abi.HOST_POINTERS, None) == -1
    assert b'rox_calc_psf' in lib.rox_last_error()


@pytest.mark.gpu
@pytest.mark.parametrize('tag', ['dblgauss_f0', 'dblgauss_f2', 'dblgauss_f1_odd_size'])
def test_trace_to_psf_on_the_device(tag):
    """the whole analysis on the device: pupil grid trace + OPD epilogue
    (analyses.eval_wavefront on a table-backed model) -> rox_calc_psf, against the PSF the
    reference computed from its own eval_wavefront grid"""
    import json
    from rayoptics_amd import SurfaceTable, analyses, session, workloads
    from rayoptics_amd.table import wavefront_from_array
    import helpers as H
    z = np.load(GOLDEN)
    tbl = SurfaceTable.from_dict(json.loads(str(z['dblgauss_table_json'])))
    ndim, maxdim = (int(v) for v in z[f'{tag}/dims'])
    wi = int(z[f'{tag}/wvl_idx'])
    wl = workloads.SimpleWorkload(tbl, [H.field_from_arr(z[f'{tag}/field'])], [(0., 0.)])
    m = workloads.TableModel(wl)
    m.fields[0].rox_wavefront = wavefront_from_array(z[f'{tag}/wavefront'])
    m.fields[0]._vig_bbox = (z[f'{tag}/bbox'][0], z[f'{tag}/bbox'][1])
    m._units_per_nm = 1.0 / (float(z[f'{tag}/convert_to_opd']) * tbl.wvls[wi])
    grid = analyses.eval_wavefront(m, m.fields[0], tbl.wvls[wi], 0.0, num_rays=ndim)
    opd = np.rollaxis(grid, 2)[2]                       # RayGrid.update_data's view
    exp_opd = z[f'{tag}/opd']
    assert np.array_equal(np.isnan(opd), np.isnan(exp_opd))
    ok = ~np.isnan(exp_opd)
    assert np.abs(opd[ok] - exp_opd[ok]).max() <= 1e-10 * float(z[f'{tag}/convert_to_opd'])
    psf = analyses.calc_psf(opd, ndim, maxdim)
    np.testing.assert_allclose(psf, z[f'{tag}/psf'], rtol=0, atol=1e-9)
    session.clear()


@pytest.mark.gpu
def test_psf_calls_from_several_threads():
    """host-array calls from several Python threads (NULL stream, shared workspace)"""
    import threading
    from rayoptics_amd.engine import calc_psf
    rng = np.random.default_rng(9)
    jobs = []
    for ndim, maxdim in ((8, 20), (16, 64), (32, 100), (64, 256)):
        opd = 0.7 * rng.standard_normal((ndim, ndim))
        jobs.append((opd, ndim, maxdim, calc_psf(opd, ndim, maxdim)))
    errs = []

    def worker(k):
        opd, ndim, maxdim, want = jobs[k]
        for _ in range(30):
            if not np.array_equal(calc_psf(opd, ndim, maxdim), want):
                errs.append(k)
                return
    ths = [threading.Thread(target=worker, args=(k,)) for k in range(len(jobs))]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    assert not errs, errs

