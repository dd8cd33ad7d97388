"""run by tests/test_gpu_r02.py::test_chunked_launches in a subprocess with
ROX_RAYS_PER_LAUNCH set (the library reads it once): a ragged batch over the
multi-launch path, every output mode (OPD and FAN epilogues, host-pointer staging
included), vs the oracle."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

import rayoptics_amd  # noqa: E402,F401
from rayoptics_amd import abi  # noqa: E402
from rayoptics_amd.engine import TraceEngine  # noqa: E402
from oracle import oracle  # noqa: E402
import helpers as H  # noqa: E402

assert os.environ.get('ROX_RAYS_PER_LAUNCH') == '4096'


def same(a, b, what):
    a, b = np.asarray(a), np.asarray(b)
    ok = (a.shape == b.shape) and bool(((a == b) | (np.isnan(a) & np.isnan(b))).all())
    assert ok, what


for name in ('dblgauss', 'tilted_singlet'):
    fx = H.fixture(name)
    c = fx['grid_f2'] if name == 'dblgauss' else fx['grid_f1']
    eng = TraceEngine(fx.table)
    fld = H.field_from_arr(c['field'])
    N = fx.table.n_ifcs
    num = 131                               # 17161 rays = 4 launches + a ragged fifth
    grid = oracle.make_grid((-1., -1.), (1., 1.), num)
    for mode in (abi.OUT_FULL, abi.OUT_LAST, abi.OUT_HITS, abi.OUT_HITS_COMPACT):
        opts = H.make_opts(c, out_mode=mode, foc=0.02, image_pt=(0.1, 0.3))
        orc = oracle.trace_pupil_grid(fx.table, fld, grid, 0, opts)
        if mode == abi.OUT_HITS_COMPACT:
            same(eng.trace_pupil_grid_hits(fld, grid, 0, opts), orc.hits, f'{name} compact')
            same(eng.trace_pupil_grid_hits(fld, grid, 0, opts), orc.hits, f'{name} compact again')
            continue
        dev = eng.trace_pupil_grid(fld, grid, 0, opts, nan_fill=True).to_host()
        assert np.array_equal(dev.status, orc.status) and np.array_equal(dev.fail_surf, orc.fail_surf)
        same(dev.seg, orc.seg, f'{name} mode {mode} seg')
        same(dev.op, orc.op, f'{name} mode {mode} op')
        same(dev.pupil, orc.pupil, f'{name} mode {mode} pupil')
    # OPD and FAN epilogues over several launches (dblgauss: finite reference sphere)
    if name == 'dblgauss':
        from test_oracle_golden import opd_opts
        co = fx['opd_f2']
        f2 = H.field_from_arr(co['field'])
        g2 = oracle.make_grid(co['start'], co['stop'], 97)      # 9409 rays: 3 launches
        for mode in (abi.OUT_OPD, abi.OUT_FAN):
            o = opd_opts(co)
            o.out_mode = mode
            o.foc, o.image_pt[0], o.image_pt[1] = 0.01, 0.0, 18.0
            orc = oracle.trace_pupil_grid(fx.table, f2, g2, int(co['wvl_idx']), o)
            dev = eng.trace_pupil_grid(f2, g2, int(co['wvl_idx']), o, nan_fill=True).to_host()
            assert np.array_equal(dev.status, orc.status)
            same(dev.seg, np.asarray(orc.seg).reshape(np.asarray(dev.seg).shape), f'mode {mode} seg')
        # plain host buffers (ROX_HOST_POINTERS) over several launches, both staging paths
        import ctypes as C
        for num2 in (70, 190):                                  # 4900 rays (pinned block), 36100 (arena)
            g3 = oracle.make_grid((-1., -1.), (1., 1.), num2)
            o = H.make_opts(c, out_mode=abi.OUT_FULL)
            orc = oracle.trace_pupil_grid(fx.table, fld, g3, 0, o)
            o.flags |= abi.HOST_POINTERS
            res = oracle.HostResult(N, num2 * num2, abi.OUT_FULL, want_pupil=True)
            res.seg[:] = 3.0
            out = res.out_struct()
            rc = eng.lib.rox_trace_pupil_grid(eng._handle, C.byref(fld), C.byref(g3), 0, C.byref(o),
                                              C.byref(out), None)
            assert rc == 0, eng.lib.rox_last_error()
            assert np.array_equal(res.status, orc.status)
            same(res.seg, orc.seg, f'host pointers {num2} seg')
            same(res.pupil, orc.pupil, f'host pointers {num2} pupil')
    # explicit rays with per-ray wavelengths, OPD-free modes
    if name == 'dblgauss':
        cr = fx['rays_ap']
        reps = 9
        pt0 = np.tile(cr['pt0'], (1, reps))
        d0 = np.tile(cr['dir0'], (1, reps))
        wi = np.tile(np.asarray(cr['wvl_idx'], dtype=np.int32) if np.ndim(cr['wvl_idx']) else
                     np.full(cr['pt0'].shape[1], int(cr['wvl_idx']), np.int32), reps)
        o = H.make_opts(cr)
        dev = eng.trace_rays(pt0, d0, wi, o, nan_fill=True).to_host()
        orc = oracle.trace_rays(fx.table, pt0, d0, wi, o)
        assert np.array_equal(dev.status, orc.status)
        same(dev.seg, orc.seg, 'rays seg')
        same(dev.op, orc.op, 'rays op')
    # rox_trace_pupil_grids when an item needs more than one launch (num * num > the limit): the
    # entry falls back to the plain path item by item; below the limit it is one batched launch
    if name == 'dblgauss':
        for num3, what in ((131, 'fallback'), (60, 'batched')):
            g4 = oracle.make_grid((-1., -1.), (1., 1.), num3)
            flds = [fld, H.field_from_arr(fx['grid_f2']['field']), fld]
            wis = [0, 1, 2]
            for mode in (abi.OUT_FULL, abi.OUT_HITS_COMPACT):
                optl = [H.make_opts(c, out_mode=mode, foc=0.01 * k, image_pt=(0.1, 0.3)) for k in range(3)]
                want = [oracle.trace_pupil_grid(fx.table, f, g4, w, o) for f, w, o in zip(flds, wis, optl)]
                if mode == abi.OUT_HITS_COMPACT:
                    got = eng.trace_pupil_grids_hits(flds, wis, g4, optl)
                    for g, w in zip(got, want):
                        same(g, w.hits, f'grids {what} compact')
                else:
                    got = eng.trace_pupil_grids(flds, wis, g4, optl, nan_fill=True)
                    for g, w in zip(got, want):
                        h = g.to_host()
                        assert np.array_equal(h.status, w.status)
                        same(h.seg, w.seg, f'grids {what} seg')
    eng.close()
print('chunked ok')
