"""-m gpu: the launches bench.py TIMES, bit-checked at full size against the oracle.

bench.py's main leg is `dblgauss_c2`, field 0, wi = ref_wvl_idx, 1024 x 1024, FULL packets with
the pupil pair, rays generated on the device (bench.py main(): `opts`, `out`, `step`).  Here
that exact launch -- same workload object, same options, same DeviceResult shape -- is
compared with the oracle on every ray: segments, op, status, failing surface, pupil.  The
oracle traces the grid as pupil-row blocks on threads (ctypes releases the GIL; a block
walks the x axis from the start like the full grid, so its rays are the grid's rays).
`bench.work_of`'s counts -- the figures `value` and `roofline.achieved` are computed from --
are recomputed from the ORACLE's status arrays and pinned to the numbers DESIGN quotes.
One item of each batched configs_leg launch gets the same treatment."""
import os
import sys
import threading

import numpy as np
import pytest

from rayoptics_amd import abi

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SPOT = abi.INTERSECT_OBJ | abi.CHECK_APERTURES | abi.APPLY_VIGNETTING


def oracle_blocks(table, fld, num, wi, opts, n_blocks):
    """[(row_begin, row_count, HostResult)] of the num x num grid, traced on threads"""
    from oracle import oracle
    oracle.lib()
    bounds = [(num * k) // n_blocks for k in range(n_blocks + 1)]
    out = [None] * n_blocks

    def work(k):
        g = oracle.make_grid((-1., -1.), (1., 1.), num, row_begin=bounds[k],
                             row_count=bounds[k + 1] - bounds[k])
        out[k] = (bounds[k], bounds[k + 1] - bounds[k],
                  oracle.trace_pupil_grid(table, fld, g, wi, opts))
    thr = [threading.Thread(target=work, args=(k,)) for k in range(n_blocks)]
    for t in thr:
        t.start()
    for t in thr:
        t.join()
    return out


def same_bits(a, b, what):
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    ok = (a == b) | (np.isnan(a) & np.isnan(b))
    assert ok.all(), f'{what}: {np.count_nonzero(~ok)} of {ok.size} differ, first {np.argwhere(~ok)[:3].tolist()}'


def counts(status, fail_surf, N, full):
    """bench.work_of restated on NumPy arrays (intersections performed, algorithmic bytes)"""
    ok = status == abi.OK
    fs = fail_surf.astype(np.int64)
    inters = int(ok.sum()) * (N - 1) + int(fs[~ok].sum())
    if not full:
        return inters, status.size * 19
    missed = status == abi.MISSED_SURFACE
    nseg = int(ok.sum()) * N + int(fs[missed].sum()) + int((fs[~ok & ~missed] + 1).sum())
    return inters, nseg * 80 + status.size * (8 + 1 + 2 + 16)


def test_the_main_leg_launch_bit_exact_and_its_work_counts():
    import torch
    from rayoptics_amd import workloads
    from rayoptics_amd.engine import TraceEngine, make_opts, make_grid, DeviceResult
    sys.path.insert(0, ROOT)
    import bench
    wl = workloads.load('dblgauss_c2')
    N = wl.n_ifcs
    num = 1024
    R = num * num
    eng = TraceEngine(wl.table)
    fi, wi = 0, wl.ref_wvl_idx              # rank 0 of bench.py
    fld = wl.fields[fi]
    grid = make_grid((-1., -1.), (1., 1.), num)
    opts = make_opts(flags=SPOT, out_mode=abi.OUT_FULL, first_surf=1, last_surf=N - 2)
    # bench.py allocates with nan_fill=False; NaN-filled here so that the slots the launch must
    # NOT write (segments past a failure) are checked too -- the launch is the same
    out = DeviceResult(torch, eng.device, eng.num_segments(SPOT), R, abi.OUT_FULL,
                       want_pupil=True, nan_fill=True)
    eng.trace_pupil_grid(fld, grid, wi, opts, out=out)
    torch.cuda.synchronize()
    d_inters, d_bytes = bench.work_of(out.status, out.fail_surf, N, abi, full=True)
    dev = out.to_host()
    n_thr = max(1, min(32, os.cpu_count() or 1))
    status = np.empty(R, dtype=np.uint8)
    fsurf = np.empty(R, dtype=np.int16)
    for r0, rc, orc in oracle_blocks(wl.table, fld, num, wi, opts, n_thr):
        sl = slice(r0 * num, (r0 + rc) * num)
        np.testing.assert_array_equal(dev.status[sl], orc.status)
        np.testing.assert_array_equal(dev.fail_surf[sl], orc.fail_surf)
        same_bits(dev.seg[:, :, sl], orc.seg, f'seg rows {r0}..')
        same_bits(dev.op[sl], orc.op, f'op rows {r0}..')
        same_bits(dev.pupil[:, sl], orc.pupil, f'pupil rows {r0}..')
        status[sl], fsurf[sl] = orc.status, orc.fail_surf
    o_inters, o_bytes = counts(status, fsurf, N, full=True)
    assert (d_inters, d_bytes) == (o_inters, o_bytes)
    # the figures DESIGN section 4 and the bench line quote for this launch
    assert o_inters == 11_158_584
    assert o_bytes == 1_004_884_352
    assert int((status == abi.OK).sum()) == 821_936
    eng.close()


@pytest.mark.parametrize('name,num,item,mode', [
    ('nikkor_c3', 512, (2, 0), abi.OUT_FULL),
    ('nikkor_c3', 512, (1, 2), abi.OUT_HITS),
    ('zmx_evenasph_c3', 512, (2, 1), abi.OUT_FULL),
    ('rc_telescope_c4', 256, (4, None), abi.OUT_FULL),
    ('rc_telescope_c4', 256, (3, None), abi.OUT_HITS),
])
def test_an_item_of_each_batched_config_launch(name, num, item, mode):
    """bench.py configs_leg: every (field, wavelength) grid of a configuration in ONE launch
    (rox_trace_pupil_grids) with the leg's own options; the named item of that launch ==
    the oracle on every ray (C5's 64-row blocks: tests/test_gpu_product.py)"""
    import torch
    from rayoptics_amd import workloads
    from rayoptics_amd.engine import TraceEngine, make_opts, make_grid, DeviceResult
    wl = workloads.load(name)
    N = wl.n_ifcs
    eng = TraceEngine(wl.table)
    nf, nw = len(wl.fields), len(wl.table.wvls)
    wis = [wl.ref_wvl_idx] if item[1] is None else list(range(nw))
    pairs = [(f, w) for f in range(nf) for w in wis]
    R = num * num
    grid = make_grid((-1., -1.), (1., 1.), num)
    wide = [abi.INTERSECT_OBJ if (f.kind != abi.FLD_EPD_WIDE and f.z_dir0 != 0.0) else 0
            for f in wl.fields]
    full = mode == abi.OUT_FULL
    ress = [DeviceResult(torch, eng.device, eng.num_segments(0), R, mode, want_pupil=full,
                         nan_fill=True) for _ in pairs]
    optl = [make_opts(flags=(SPOT & ~abi.INTERSECT_OBJ) | wide[f], out_mode=mode, first_surf=1,
                      last_surf=N - 2, foc=wl.foc, image_pt=wl.image_pts[f]) for f, _w in pairs]
    eng.trace_pupil_grids([wl.fields[f] for f, _w in pairs], [w for _f, w in pairs], grid, optl,
                          outs=ress)
    torch.cuda.synchronize()
    want = (item[0], wl.ref_wvl_idx if item[1] is None else item[1])
    k = pairs.index(want)
    dev = ress[k].to_host()
    n_ok = 0
    for r0, rc, orc in oracle_blocks(wl.table, wl.fields[want[0]], num, want[1], optl[k],
                                     max(1, min(16, os.cpu_count() or 1))):
        sl = slice(r0 * num, (r0 + rc) * num)
        np.testing.assert_array_equal(dev.status[sl], orc.status)
        np.testing.assert_array_equal(dev.fail_surf[sl], orc.fail_surf)
        same_bits(dev.seg[..., sl], orc.seg, f'{name} item {want} seg rows {r0}..')
        same_bits(dev.op[sl], orc.op, f'{name} item {want} op')
        if full:
            same_bits(dev.pupil[:, sl], orc.pupil, 'pupil')
        n_ok += int((orc.status == abi.OK).sum())
    assert n_ok > 0.2 * R
    eng.close()
