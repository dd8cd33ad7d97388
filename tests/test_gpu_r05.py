"""-m gpu, round 5: the small-workgroup kernels (roxtrace.hip want_small) give the bytes the
large-workgroup kernels give; the reworked asphere evaluation against the oracle at full size;
many threads over more models than the session keeps handles for."""
import os
import subprocess
import sys
import threading
import time

import numpy as np
import pytest

from rayoptics_amd import abi

pytestmark = pytest.mark.gpu

SPOT = abi.INTERSECT_OBJ | abi.CHECK_APERTURES | abi.APPLY_VIGNETTING
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_SMALL_SCRIPT = r'''
import hashlib, json, sys
import numpy as np
sys.path.insert(0, %(root)r)
import rayoptics_amd
from rayoptics_amd import abi, workloads
from rayoptics_amd.engine import TraceEngine, make_opts, make_grid
SPOT = abi.INTERSECT_OBJ | abi.CHECK_APERTURES | abi.APPLY_VIGNETTING
out = {}
for name, num in (('dblgauss_c2', 97), ('rc_telescope_c4', 256), ('nikkor_c3', 130), ('cell_phone', 65),
                  ('zmx_evenasph_c3', 200)):
    wl = workloads.load(name)
    N = wl.n_ifcs
    eng = TraceEngine(wl.table)
    grid = make_grid((-1., -1.), (1., 1.), num)
    fis = list(range(len(wl.fields)))
    for mode in (abi.OUT_FULL, abi.OUT_HITS, abi.OUT_LAST):
        h = hashlib.sha256()
        optl = [make_opts(flags=SPOT, out_mode=mode, first_surf=1, last_surf=N - 2, foc=wl.foc,
                          image_pt=wl.image_pts[f]) for f in fis]
        # one launch per grid, then the same grids batched into one launch
        singles = [eng.trace_pupil_grid(wl.fields[f], grid, wl.ref_wvl_idx, optl[f], nan_fill=True).to_host()
                   for f in fis]
        batch = [r.to_host() for r in eng.trace_pupil_grids([wl.fields[f] for f in fis],
                                                            [wl.ref_wvl_idx] * len(fis), grid, optl,
                                                            nan_fill=True)]
        for a, b in zip(singles, batch):
            for x, y in ((a.seg, b.seg), (a.op, b.op), (a.status, b.status), (a.fail_surf, b.fail_surf)):
                assert np.array_equal(x, y, equal_nan=True), (name, mode, 'batch differs from single launches')
            for x in (a.seg, a.op, a.status, a.fail_surf):
                h.update(np.ascontiguousarray(x).tobytes())
        out['%%s/%%d' %% (name, mode)] = h.hexdigest()
    # explicit rays (the list seam): a few hundred rays = a small launch
    rng = np.random.default_rng(5)
    R = 333
    pt0 = np.stack([rng.uniform(-2, 2, R), rng.uniform(-2, 2, R), np.full(R, -50.0)])
    d = np.stack([rng.uniform(-.02, .02, R), rng.uniform(-.02, .02, R), np.ones(R)])
    d /= np.sqrt((d * d).sum(0))
    o = make_opts(flags=abi.INTERSECT_OBJ | abi.CHECK_APERTURES, out_mode=abi.OUT_FULL, first_surf=1, last_surf=N - 2)
    r = eng.trace_rays(pt0, d, wl.ref_wvl_idx, o, nan_fill=True).to_host()
    h = hashlib.sha256()
    for x in (r.seg, r.op, r.status, r.fail_surf):
        h.update(np.ascontiguousarray(x).tobytes())
    out[name + '/rays'] = h.hexdigest()
    eng.close()
print(json.dumps(out))
'''


def test_small_and_large_workgroups_give_the_same_bytes():
    """ROX_SMALL_BLOCKS is read once per process: the same launches in two processes, one
    forcing the small-workgroup kernels and one forbidding them -- FULL / HITS / LAST packets,
    single and batched launches, explicit rays, five instances -- must hash the same"""
    import json
    res = {}
    for v in ('0', '1'):
        env = dict(os.environ, ROX_SMALL_BLOCKS=v)
        p = subprocess.run([sys.executable, '-c', _SMALL_SCRIPT % {'root': ROOT}], env=env, cwd=ROOT,
                           capture_output=True, text=True, timeout=900)
        assert p.returncode == 0, p.stderr[-3000:]
        res[v] = json.loads(p.stdout.strip().splitlines()[-1])
    assert res['0'] == res['1']
    assert len(res['0']) == 5 * 4


@pytest.mark.parametrize('name,fi', [('zmx_evenasph_c3', 0), ('zmx_evenasph_c3', 2), ('nikkor_c3', 1),
                                     ('cell_phone', 2)])
def test_asphere_instances_full_size_against_the_oracle(name, fi):
    """a 1024 x 1024 grid through the Newton instances (shared power chain, branch on
    (cc + 1) == ec, slim Spencer-Murty quotient): every packet component of every ray equals
    the oracle's -- compared in row blocks so that the oracle's share stays in seconds"""
    from oracle import oracle
    from rayoptics_amd import workloads
    from rayoptics_amd.engine import TraceEngine, make_opts, make_grid
    wl = workloads.load(name)
    N = wl.n_ifcs
    eng = TraceEngine(wl.table)
    wi = wl.ref_wvl_idx
    num = 1024
    o = make_opts(flags=SPOT, out_mode=abi.OUT_FULL, first_surf=1, last_surf=N - 2, foc=wl.foc,
                  image_pt=wl.image_pts[fi])
    dev = eng.trace_pupil_grid(wl.fields[fi], make_grid((-1., -1.), (1., 1.), num), wi, o,
                               nan_fill=True).to_host()
    oo = oracle.make_opts(flags=SPOT, out_mode=abi.OUT_FULL, first_surf=1, last_surf=N - 2, foc=wl.foc,
                          image_pt=wl.image_pts[fi])
    for row0 in (0, 311, 512, 1000):
        rows = 24
        orc = oracle.trace_pupil_grid(wl.table, wl.fields[fi],
                                      oracle.make_grid((-1., -1.), (1., 1.), num, row_begin=row0,
                                                       row_count=rows), wi, oo)
        sl = slice(row0 * num, (row0 + rows) * num)
        assert np.array_equal(dev.status[sl], orc.status)
        assert np.array_equal(dev.fail_surf[sl], orc.fail_surf)
        assert np.array_equal(dev.seg[:, :, sl], orc.seg, equal_nan=True)
        assert np.array_equal(dev.op[sl], orc.op, equal_nan=True)
    eng.close()


def _perturbed_models(n):
    """n table-backed models with distinct surface tables (curvatures of the double Gauss and of
    the phone lens nudged): distinct device handles for the session cache"""
    from rayoptics_amd import workloads
    from rayoptics_amd.table import SurfaceTable
    models = []
    for k in range(n):
        base = workloads.load('dblgauss_c2' if k % 2 == 0 else 'cell_phone')
        d = base.table.to_dict()
        t = SurfaceTable.from_dict(d)
        t.rows[2].cv = t.rows[2].cv * (1.0 + 1e-4 * (k + 1))
        wl = workloads.SimpleWorkload(t, base.fields, base.image_pts, foc=base.foc,
                                      ref_wvl_idx=base.ref_wvl_idx, name=f'm{k}')
        models.append(workloads.TableModel(wl))
    return models


def test_eight_threads_over_twenty_models_with_sixteen_handles():
    """session.engine_for from 8 threads over 20 distinct models while the cache keeps 16
    handles: engines are evicted (closed) under threads that still hold them and re-open on
    their next call; PinnedPool leases come and go on every thread.  No crash, no error, every
    result equal to the oracle's."""
    from oracle import oracle
    from rayoptics_amd import session
    from rayoptics_amd.engine import make_opts, make_grid, TraceEngine
    session.clear()
    assert session.MAX_ENGINES == 16
    models = _perturbed_models(20)
    grid = make_grid((-1., -1.), (1., 1.), 48)
    want = []
    for m in models:
        wl = m.workload
        N = wl.n_ifcs
        oo = oracle.make_opts(flags=SPOT, out_mode=abi.OUT_HITS_COMPACT, first_surf=1, last_surf=N - 2,
                              foc=wl.foc, image_pt=wl.image_pts[1])
        want.append(oracle.trace_pupil_grid(wl.table, wl.fields[1], oracle.make_grid((-1., -1.), (1., 1.), 48),
                                            wl.ref_wvl_idx, oo).hits)
    errors, counts = [], [0] * 8
    stop = time.time() + 10.0
    reopened = [0]

    def worker(tid):
        rng = np.random.default_rng(100 + tid)
        try:
            while time.time() < stop:
                k = int(rng.integers(len(models)))
                m = models[k]
                wl = m.workload
                eng = session.engine_for(m)
                assert isinstance(eng, TraceEngine)
                if rng.random() < 0.2:
                    time.sleep(0.002)           # hold the engine while others evict it
                    if not eng.is_open:
                        reopened[0] += 1
                o = make_opts(flags=SPOT, out_mode=abi.OUT_HITS_COMPACT, first_surf=1,
                              last_surf=wl.n_ifcs - 2, foc=wl.foc, image_pt=wl.image_pts[1])
                xy = eng.trace_pupil_grid_hits(wl.fields[1], grid, wl.ref_wvl_idx, o)
                if not np.array_equal(xy, want[k]):
                    errors.append((tid, k, 'result differs from the oracle'))
                    return
                counts[tid] += 1
        except Exception as e:      # noqa: BLE001
            errors.append((tid, repr(e)))

    ths = [threading.Thread(target=worker, args=(t,)) for t in range(8)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    session.clear()
    assert not errors, errors[:3]
    assert min(counts) > 20, counts
    assert len(session._cache) == 0


@pytest.mark.parametrize('name', ['zmx_evenasph_c3', 'cell_phone', 'nikkor_c3'])
def test_patch_lane_mapping_leaves_every_ray_where_it_was(name, monkeypatch):
    """ROX_PATCH8=1 (an experiment switch, read per call; off in the product because it measured
    slower): reduced-output launches of the Newton instances give a wave an 8 x 8 pupil patch
    when the grid divides into such tiles (512, 192: yes -- with 512- and 256-thread workgroups;
    200, a 24-row block of 512: no / yes): HITS, LAST and a batched HITS launch against the
    oracle, ray for ray"""
    monkeypatch.setenv('ROX_PATCH8', '1')
    from oracle import oracle
    from rayoptics_amd import workloads
    from rayoptics_amd.engine import TraceEngine, make_opts, make_grid
    wl = workloads.load(name)
    N = wl.n_ifcs
    eng = TraceEngine(wl.table)
    wi = wl.ref_wvl_idx
    fi = len(wl.fields) - 1
    for mode in (abi.OUT_HITS, abi.OUT_LAST):
        for gkw in (dict(num=512), dict(num=192), dict(num=200), dict(num=512, row_begin=40, row_count=24),
                    dict(num=512, row_begin=3, row_count=13)):
            o = make_opts(flags=SPOT, out_mode=mode, first_surf=1, last_surf=N - 2, foc=wl.foc,
                          image_pt=wl.image_pts[fi])
            dev = eng.trace_pupil_grid(wl.fields[fi], make_grid((-1., -1.), (1., 1.), **gkw), wi, o,
                                       nan_fill=True).to_host()
            oo = oracle.make_opts(flags=SPOT, out_mode=mode, first_surf=1, last_surf=N - 2, foc=wl.foc,
                                  image_pt=wl.image_pts[fi])
            orc = oracle.trace_pupil_grid(wl.table, wl.fields[fi], oracle.make_grid((-1., -1.), (1., 1.), **gkw),
                                          wi, oo)
            assert np.array_equal(dev.status, orc.status), (mode, gkw)
            assert np.array_equal(dev.seg, orc.seg, equal_nan=True), (mode, gkw)
            assert np.array_equal(dev.op, orc.op, equal_nan=True)
            assert np.array_equal(dev.pupil, orc.pupil)
    # batched: every field in one launch
    fis = list(range(len(wl.fields)))
    optl = [make_opts(flags=SPOT, out_mode=abi.OUT_HITS, first_surf=1, last_surf=N - 2, foc=wl.foc,
                      image_pt=wl.image_pts[f]) for f in fis]
    res = eng.trace_pupil_grids([wl.fields[f] for f in fis], [wi] * len(fis),
                                make_grid((-1., -1.), (1., 1.), 256), optl, nan_fill=True)
    for f, r in zip(fis, res):
        h = r.to_host()
        oo = oracle.make_opts(flags=SPOT, out_mode=abi.OUT_HITS, first_surf=1, last_surf=N - 2, foc=wl.foc,
                              image_pt=wl.image_pts[f])
        orc = oracle.trace_pupil_grid(wl.table, wl.fields[f], oracle.make_grid((-1., -1.), (1., 1.), 256), wi, oo)
        assert np.array_equal(h.status, orc.status)
        assert np.array_equal(h.seg, orc.seg, equal_nan=True)
    eng.close()


def _flat_table(aps, max_aperture=50.0):
    """object plane, one flat dummy carrying the clear apertures `aps`, image plane"""
    from rayoptics_amd import SurfaceTable
    rows = (abi.Surface * 3)()
    for i, r in enumerate(rows):
        r.mode, r.profile, r.ec = abi.DUMMY, abi.SPHERICAL, 1.0
        for k in range(3):
            r.rt[4 * k] = 1.0
        r.t[2] = 5.0 if i < 2 else 0.0
        r.z_dir = 1.0
        r.max_aperture = 1e12
    rows[1].max_aperture = max_aperture
    rows[1].n_ap = len(aps)
    for k, (kind, obsc, ox, oy, a, b) in enumerate(aps):
        ap = rows[1].ap[k]
        ap.kind, ap.is_obscuration, ap.x_offset, ap.y_offset, ap.a, ap.b = kind, obsc, ox, oy, a, b
    return SurfaceTable(rows, np.ones((1, 3)), [550.0])


@pytest.mark.parametrize('aps', [
    [],                                                                     # max_aperture alone
    [(abi.AP_CIRCULAR, 0, 0.0, 0.0, 3.7, 0.0)],
    [(abi.AP_CIRCULAR, 0, 0.31, -0.77, 2.9, 0.0)],
    [(abi.AP_CIRCULAR, 1, -0.4, 0.2, 1.3, 0.0)],                            # an obscuration
    [(abi.AP_CIRCULAR, 0, 0.1, 0.1, 4.1, 0.0), (abi.AP_CIRCULAR, 1, 0.0, 0.3, 0.9, 0.0),
     (abi.AP_RECTANGULAR, 0, 0.2, 0.0, 3.0, 2.5)],
])
def test_aperture_edges_decided_as_the_square_root_decides_them(aps):
    """the aperture tests compare x^2 + y^2 with a staged threshold instead of taking the square
    root (rox_device.hpp sqrt_le_threshold / stage_aperture_thresholds).  Rays parallel to the
    axis land on a flat surface exactly where they start, so their start points are laid within
    +-6 ulp of every circular edge (radius + fuzz, along 97 directions) and of the rectangular
    ones: blocked / passed must be the oracle's -- which takes the square root, like the
    reference -- for every one of them, with two fuzz values"""
    from oracle import oracle
    from rayoptics_amd.engine import TraceEngine
    max_ap = 6.3
    tbl = _flat_table(aps, max_ap)
    eng = TraceEngine(tbl)
    n_edge = 0
    for fuzz in (1e-5, 1e-4):
        pts = []
        circles = [(0.0, 0.0, max_ap)] if not aps else [(ox, oy, a) for k, _o, ox, oy, a, _b in aps
                                                         if k == abi.AP_CIRCULAR]
        for ox, oy, rad in circles:
            t = rad + fuzz
            for ang in np.linspace(0.0, 2 * np.pi, 97):
                c, s_ = np.cos(ang), np.sin(ang)
                for scale in (1.0, np.nextafter(1.0, 2.0), np.nextafter(1.0, 0.0)):
                    x, y = ox + t * scale * c, oy + t * scale * s_
                    for kx in range(-6, 7, 3):
                        xx = x
                        for _ in range(abs(kx)):
                            xx = np.nextafter(xx, np.inf if kx > 0 else -np.inf)
                        pts.append((xx, y))
        for k, _o, ox, oy, a, b in aps:
            if k == abi.AP_RECTANGULAR:
                for sx in (-1, 1):
                    e = ox + sx * (a + fuzz)
                    for kx in range(-4, 5):
                        xx = e
                        for _ in range(abs(kx)):
                            xx = np.nextafter(xx, np.inf if kx > 0 else -np.inf)
                        pts.append((xx, oy + 0.3))
        pts = np.array(pts)
        R = len(pts)
        pt0 = np.stack([pts[:, 0], pts[:, 1], np.zeros(R)])
        d = np.stack([np.zeros(R), np.zeros(R), np.ones(R)])
        for mode in (abi.OUT_FULL, abi.OUT_HITS):
            opts = oracle.make_opts(flags=abi.INTERSECT_OBJ | abi.CHECK_APERTURES, out_mode=mode, first_surf=1,
                                    last_surf=1, fuzz=fuzz, image_pt=(0.0, 0.0))
            orc = oracle.trace_rays(tbl, pt0, d, 0, opts)
            dev = eng.trace_rays(pt0, d, 0, opts, nan_fill=True).to_host()
            assert np.array_equal(dev.status, orc.status), (fuzz, mode, int((dev.status != orc.status).sum()))
            assert np.array_equal(dev.fail_surf, orc.fail_surf)
        n_edge += int((orc.status == abi.BLOCKED).sum())
        assert 0 < int((orc.status == abi.BLOCKED).sum()) < R       # the points really straddle the edges
    eng.close()
    assert n_edge > 0


@pytest.mark.parametrize('name', ['dblgauss_c2', 'cell_phone', 'rc_telescope_c4'])
def test_host_pointer_small_launches_equal_the_device_results(name):
    """TraceEngine.trace_pupil_np / trace_rays_np (NumPy buffers through ROX_HOST_POINTERS: what
    the drop-ins use for launches of <= 1 MiB) against the DeviceResult path of the same
    launch: FULL / LAST / HITS, a vignetted grid with failing rays, a fan, a row block, a pupil
    list, explicit rays with per-ray wavelengths -- every array equal, NaN where nothing is
    produced"""
    from rayoptics_amd import workloads
    from rayoptics_amd.engine import TraceEngine, make_opts, make_grid
    wl = workloads.load(name)
    N = wl.n_ifcs
    eng = TraceEngine(wl.table)
    fi = len(wl.fields) - 1
    fld = wl.fields[fi]
    wi = wl.ref_wvl_idx

    def same(h, d):
        assert np.array_equal(h.status, d.status) and np.array_equal(h.fail_surf, d.fail_surf)
        assert np.array_equal(h.op, d.op, equal_nan=True)
        ok = d.status == abi.OK
        hs, ds = (h.seg, d.seg) if h.seg.ndim == 3 else (h.seg[None], d.seg[None])
        assert np.array_equal(hs[:, :, ok], ds[:, :, ok])
        assert np.isnan(hs[-1][:, ~ok]).all()            # nothing produced there: NaN
        if getattr(h, 'pupil', None) is not None and d.pupil is not None:
            assert np.array_equal(h.pupil, d.pupil)
    rng = np.random.default_rng(11)
    for mode in (abi.OUT_FULL, abi.OUT_LAST, abi.OUT_HITS):
        o = make_opts(flags=SPOT, out_mode=mode, first_surf=1, last_surf=N - 2, foc=wl.foc,
                      image_pt=wl.image_pts[fi])
        for g in (make_grid((-1.2, -1.2), (1.2, 1.2), 23), make_grid((0., -1.), (0., 1.), 21, abi.GRID_FAN),
                  make_grid((-1., -1.), (1., 1.), 40, row_begin=5, row_count=7)):
            same(eng.trace_pupil_np(fld, wi, o, grid=g),
                 eng.trace_pupil_grid(fld, g, wi, o, nan_fill=True).to_host())
        px, py = rng.uniform(-1.3, 1.3, 57), rng.uniform(-1.3, 1.3, 57)
        same(eng.trace_pupil_np(fld, wi, o, px=px, py=py),
             eng.trace_pupil_list(fld, px, py, wi, o, nan_fill=True).to_host())
        R = 41
        pt0 = np.stack([rng.uniform(-1, 1, R), rng.uniform(-1, 1, R), np.full(R, -30.0)])
        d = np.stack([rng.uniform(-.05, .05, R), rng.uniform(-.05, .05, R), np.ones(R)])
        d /= np.sqrt((d * d).sum(0))
        o2 = make_opts(flags=abi.INTERSECT_OBJ | abi.CHECK_APERTURES, out_mode=mode, first_surf=1, last_surf=N - 2)
        for w in (wi, rng.integers(0, len(wl.table.wvls), R).astype(np.int32)):
            same(eng.trace_rays_np(pt0, d, w, o2), eng.trace_rays(pt0, d, w, o2, nan_fill=True).to_host())
    eng.close()
