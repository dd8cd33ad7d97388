"""-m gpu: rox_trace_pupil_grids -- the (field x wavelength) grids of one system traced in
ONE launch (blockIdx.y = item).  Every item must equal, bit for bit, what the oracle gives
for that (field, wavelength) and what the single-grid entry gives on the device."""
import numpy as np
import pytest

from rayoptics_amd import abi
import helpers as H

pytestmark = pytest.mark.gpu

SPOT = abi.INTERSECT_OBJ | abi.CHECK_APERTURES | abi.APPLY_VIGNETTING


def _flags(wl, fi):
    f = wl.fields[fi]
    wide = f.kind == abi.FLD_EPD_WIDE or f.z_dir0 == 0.0
    return (SPOT & ~abi.INTERSECT_OBJ) | (0 if wide else abi.INTERSECT_OBJ)


def _items(wl, pairs, mode, **kw):
    from rayoptics_amd.engine import make_opts
    N = wl.n_ifcs
    return [make_opts(flags=_flags(wl, fi), out_mode=mode, first_surf=1, last_surf=N - 2,
                      foc=wl.foc, image_pt=wl.image_pts[fi], **kw) for fi, _wi in pairs]


def _same(dev, orc, mode, what):
    np.testing.assert_array_equal(dev.status, orc.status, err_msg=what)
    if mode != abi.OUT_HITS_COMPACT:
        np.testing.assert_array_equal(dev.fail_surf, orc.fail_surf, err_msg=what)
        H.bit_equal(dev.op, orc.op, what + ' op')
    H.bit_equal(dev.seg, orc.seg, what + ' seg')


@pytest.mark.parametrize('name,num', [('dblgauss_c2', 96), ('nikkor_c3', 64), ('cell_phone', 50),
                                      ('rc_telescope_c4', 77), ('zmx_evenasph_c3', 64)])
@pytest.mark.parametrize('mode', [abi.OUT_FULL, abi.OUT_HITS, abi.OUT_LAST])
def test_batched_grids_equal_the_oracle_item_by_item(name, num, mode):
    from oracle import oracle
    from rayoptics_amd import workloads
    from rayoptics_amd.engine import TraceEngine, make_grid
    wl = workloads.load(name)
    eng = TraceEngine(wl.table)
    W = len(wl.table.wvls)
    pairs = [(fi, wi) for fi in range(len(wl.fields)) for wi in range(W)]
    grid = make_grid((-1., -1.), (1., 1.), num)
    opts = _items(wl, pairs, mode)
    res = eng.trace_pupil_grids([wl.fields[fi] for fi, _ in pairs], [wi for _, wi in pairs], grid, opts,
                                nan_fill=True)
    assert len(res) == len(pairs)
    for (fi, wi), o, r in zip(pairs, opts, res):
        orc = oracle.trace_pupil_grid(wl.table, wl.fields[fi], grid, wi, o)
        dev = r.to_host()
        _same(dev, orc, mode, f'{name} field {fi} wvl {wi}')
        if mode == abi.OUT_FULL:
            H.bit_equal(dev.pupil, orc.pupil, 'pupil')
    # ... and the single-grid entry on the device gives the same packets
    fi, wi = pairs[-1]
    one = eng.trace_pupil_grid(wl.fields[fi], grid, wi, opts[-1], nan_fill=True).to_host()
    last = res[-1].to_host()
    H.bit_equal(one.seg, last.seg, 'single vs batched')
    eng.close()


@pytest.mark.parametrize('name,num', [('dblgauss_c2', 21), ('dblgauss_c2', 200), ('cell_phone', 64),
                                      ('rc_telescope_c4', 256)])
def test_batched_packed_hits(name, num):
    """ROX_OUT_HITS_COMPACT per item (own tickets and look-back states): every item's
    (R_ok, 2) array in ray order, into pinned host memory, twice in a row on one stream
    (the states are re-armed by the launch itself)"""
    from oracle import oracle
    from rayoptics_amd import workloads
    from rayoptics_amd.engine import TraceEngine, make_grid
    wl = workloads.load(name)
    eng = TraceEngine(wl.table)
    W = len(wl.table.wvls)
    pairs = [(fi, wi) for fi in range(len(wl.fields)) for wi in range(W)]
    grid = make_grid((-1., -1.), (1., 1.), num)
    opts = _items(wl, pairs, abi.OUT_HITS_COMPACT)
    want = [oracle.trace_pupil_grid(wl.table, wl.fields[fi], grid, wi, o).hits
            for (fi, wi), o in zip(pairs, opts)]
    for rep in range(3):
        got = eng.trace_pupil_grids_hits([wl.fields[fi] for fi, _ in pairs], [wi for _, wi in pairs],
                                         grid, opts)
        for (fi, wi), g, w in zip(pairs, got, want):
            assert g.shape == w.shape, (name, fi, wi, rep, g.shape, w.shape)
            assert np.array_equal(g, w), (name, fi, wi, rep)
    # a different batch size on the same stream context afterwards
    got = eng.trace_pupil_grids_hits([wl.fields[0]], [0], grid, opts[:1])
    assert np.array_equal(got[0], want[0])
    eng.close()


def test_batched_wavefront_maps():
    """ROX_OUT_OPD: every item carries its own reference sphere (rox_opts.wf); the stored
    wavefront cases of one model that share a grid go through one launch"""
    from oracle import oracle
    from rayoptics_amd.engine import TraceEngine
    from test_oracle_golden import OPD_CASES, opd_opts
    by_model = {}
    for name, case in OPD_CASES:
        by_model.setdefault(name, []).append(case)
    done = 0
    for name, cases in by_model.items():
        fx = H.fixture(name)
        groups = {}
        for case in cases:
            c = fx[case]
            key = (tuple(np.asarray(c['start'], float)), tuple(np.asarray(c['stop'], float)), int(c['num']))
            groups.setdefault(key, []).append(c)
        eng = TraceEngine(fx.table)
        for (start, stop, num), cs in groups.items():
            cs = cs + cs[:1]            # (at least two items; the repeat must give the same map)
            grid = oracle.make_grid(start, stop, num)
            flds = [H.field_from_arr(c['field']) for c in cs]
            wis = [int(c['wvl_idx']) for c in cs]
            opts = [opd_opts(c) for c in cs]
            res = eng.trace_pupil_grids(flds, wis, grid, opts, nan_fill=True)
            for c, f, w, o, r in zip(cs, flds, wis, opts, res):
                orc = oracle.trace_pupil_grid(fx.table, f, grid, w, o)
                dev = r.to_host()
                np.testing.assert_array_equal(dev.status, orc.status)
                H.bit_equal(dev.seg, orc.seg, f'{name} OPD batch')
                done += 1
        eng.close()
    assert done >= 4


def test_batched_fans_all_wavelengths_in_one_launch():
    """ROX_OUT_FAN over a GRID_FAN grid, one item per wavelength with its own reference sphere,
    focus and image point -- what the rebound SequentialModel.trace_fan launches for
    RayFanFigure -- item by item equal to the oracle"""
    from oracle import oracle
    from rayoptics_amd.engine import TraceEngine
    from test_oracle_golden import OPD_CASES, opd_opts
    done = 0
    for name in ('dblgauss', 'telecentric'):
        fx = H.fixture(name)
        cs = [fx[case] for n, case in OPD_CASES if n == name]
        eng = TraceEngine(fx.table)
        for xy in (0, 1):
            start, stop = np.zeros(2), np.zeros(2)
            start[xy], stop[xy] = -1.0, 1.0
            grid = oracle.make_grid(start, stop, 21, abi.GRID_FAN)
            flds, wis, opts = [], [], []
            for k, c in enumerate(cs):
                o = opd_opts(c)
                o.out_mode = abi.OUT_FAN
                o.flags |= abi.APPLY_VIGNETTING
                o.foc, o.image_pt[0], o.image_pt[1] = 0.01 * k, 0.01, -0.02 * k
                flds.append(H.field_from_arr(c['field']))
                wis.append(int(c['wvl_idx']))
                opts.append(o)
            res = eng.trace_pupil_grids(flds, wis, grid, opts, nan_fill=True)
            for f, w, o, r in zip(flds, wis, opts, res):
                orc = oracle.trace_pupil_grid(fx.table, f, grid, w, o)
                dev = r.to_host()
                np.testing.assert_array_equal(dev.status, orc.status)
                H.bit_equal(dev.seg, orc.seg, f'{name} fan xy={xy}')
                H.bit_equal(dev.pupil, orc.pupil, 'pupil')
                done += 1
        eng.close()
    assert done >= 8


def test_batches_from_two_threads_on_two_streams():
    """two host threads, each with its own HIP stream, enqueue batches of different shapes on
    ONE handle at the same time (item slots, per-item tickets and look-back states live in the
    per-stream context): every item still equals the oracle"""
    import threading
    import torch
    from oracle import oracle
    from rayoptics_amd import workloads
    from rayoptics_amd.engine import TraceEngine, make_grid
    wl = workloads.load('dblgauss_c2')
    eng = TraceEngine(wl.table)
    W = len(wl.table.wvls)
    jobs = {0: (37, [(fi, wi) for fi in range(3) for wi in range(W)]),
            1: (120, [(2, 0), (0, 1), (1, 2), (2, 2), (0, 0)])}
    want, got, errs = {}, {0: [], 1: []}, []
    for k, (num, pairs) in jobs.items():
        grid = make_grid((-1., -1.), (1., 1.), num)
        opts = _items(wl, pairs, abi.OUT_HITS_COMPACT)
        want[k] = [oracle.trace_pupil_grid(wl.table, wl.fields[fi], grid, wi, o).hits
                   for (fi, wi), o in zip(pairs, opts)]

    def work(k):
        try:
            num, pairs = jobs[k]
            grid = make_grid((-1., -1.), (1., 1.), num)
            opts = _items(wl, pairs, abi.OUT_HITS_COMPACT)
            st = torch.cuda.Stream(device=eng.device)
            with torch.cuda.stream(st):
                for _ in range(12):
                    got[k].append([g.copy() for g in eng.trace_pupil_grids_hits(
                        [wl.fields[fi] for fi, _ in pairs], [wi for _, wi in pairs], grid, opts)])
        except Exception as e:          # noqa: BLE001
            errs.append(repr(e))
    th = [threading.Thread(target=work, args=(k,)) for k in jobs]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errs, errs
    for k in jobs:
        assert len(got[k]) == 12
        for rep in got[k]:
            for g, w in zip(rep, want[k]):
                assert g.shape == w.shape and np.array_equal(g, w), k
    eng.close()


def test_batch_argument_checks():
    from rayoptics_amd import workloads
    from rayoptics_amd.engine import TraceEngine, make_grid, EngineError
    wl = workloads.load('dblgauss_c2')
    eng = TraceEngine(wl.table)
    grid = make_grid((-1., -1.), (1., 1.), 16)
    pairs = [(0, 0), (1, 0)]
    opts = _items(wl, pairs, abi.OUT_HITS)
    flds = [wl.fields[0], wl.fields[1]]
    mixed = [opts[0], _items(wl, pairs[1:], abi.OUT_FULL)[0]]
    with pytest.raises(EngineError, match='same for every item'):
        eng.trace_pupil_grids(flds, [0, 0], grid, mixed)
    with pytest.raises(EngineError, match='out of range'):
        eng.trace_pupil_grids(flds, [0, 99], grid, opts)
    # an empty batch is a no-op; one item takes the plain path
    assert eng.trace_pupil_grids([], [], grid, []) == []
    r = eng.trace_pupil_grids(flds[:1], [0], grid, opts[:1])
    assert len(r) == 1
    eng.close()
