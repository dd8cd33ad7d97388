"""-m gpu: the reference-shaped result views (RayPkg / RaySeg / TraceError
objects) served from HIP-engine outputs, checked against the reference's own
packets stored in tests/golden (the reference itself is not on the GPU box)."""
import numpy as np
import pytest

from rayoptics_amd import abi
import helpers as H

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('name,case', [('dblgauss', 'grid_f2'), ('tilted_singlet', 'grid_f1'),
                                       ('rc_telescope', 'grid_f4')])
def test_lazy_packets_match_reference_packets(name, case):
    from rayoptics_amd.engine import TraceEngine, make_grid
    from rayoptics_amd.raypkg import HostPackets
    from rayoptics_amd import traceerror as te
    fx = H.fixture(name)
    c = fx[case]
    eng = TraceEngine(fx.table)
    fld = H.field_from_arr(c['field'])
    opts = H.make_opts(c)
    grid = make_grid(c['start'], c['stop'], int(c['num']), int(c['kind']))
    res = eng.trace_pupil_grid(fld, grid, int(c['wvl_idx']), opts)      # no nan_fill
    wvl = fx.table.wvls[int(c['wvl_idx'])]
    pk = HostPackets(res.to_host(), fx.table, opts.flags, abi.OUT_FULL, wvl)
    exp = c['seg']
    kinds = {abi.MISSED_SURFACE: te.TraceMissedSurfaceError, abi.TIR: te.TraceTIRError,
             abi.BLOCKED: te.TraceRayBlockedError}
    n_err = 0
    for r in range(exp.shape[2]):
        nseg = int(np.sum(~np.isnan(exp[:, 6, r])))         # segments the reference appended
        assert pk.nseg(r) == nseg
        ray, op, w = pk.pkg(r)
        assert len(ray) == nseg and w == wvl
        if c['status'][r] == abi.OK:
            assert abs(op - c['op'][r]) <= 1e-10 * max(1.0, abs(c['op'][r]))
        for k, sg in enumerate(ray):
            got = np.concatenate([sg[0], sg[1], [sg[2]], sg[3]])
            err = np.abs(got - exp[k, :, r]) / np.maximum(1.0, np.abs(exp[k, :, r]))
            assert err.max() <= H.ATOL
        if nseg:
            np.testing.assert_array_equal(ray[-1][0], ray[nseg - 1][0])
        if c['status'][r] != abi.OK:
            e = pk.error(r)
            assert isinstance(e, kinds[int(c['status'][r])])
            assert e.surf == c['fail_surf'][r]
            assert len(e.ray_pkg.ray) == nseg
            n_err += 1
    assert n_err > 0
    eng.close()


def test_library_is_the_hip_one():
    """the parity tests above ran on the in-tree HIP library, not a fallback"""
    import os
    from rayoptics_amd import engine
    lib = engine.load_library()
    assert os.path.basename(engine.LIB_PATH) == 'libroxtrace.so'
    with open('/proc/self/maps') as f:
        assert any('ray-optics_amd/libroxtrace.so' in line for line in f)
    assert lib.rox_abi_version() == abi.ABI_VERSION
