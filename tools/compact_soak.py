#!/usr/bin/env python3
"""Device vs oracle soak of the packed-hits output (ROX_OUT_HITS_COMPACT: stable compaction with
decoupled look-back) and of the ROX_HOST_POINTERS staging, over random grid / list sizes.

    python tools/compact_soak.py [trials]"""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))


def main():
    import rayoptics_amd  # noqa: F401
    from rayoptics_amd import abi, workloads
    from rayoptics_amd.engine import TraceEngine
    from oracle import oracle
    trials = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    rng = np.random.default_rng(21)
    t0 = time.time()
    flags = abi.INTERSECT_OBJ | abi.CHECK_APERTURES | abi.APPLY_VIGNETTING
    n = bad = n_hp = bad_hp = n_b = bad_b = 0
    for name in ('dblgauss_c2', 'rc_telescope_c4', 'nikkor_c3'):
        wl = workloads.load(name)
        eng = TraceEngine(wl.table)
        N = wl.n_ifcs
        for trial in range(trials // 3):
            fi = int(rng.integers(0, len(wl.fields)))
            wi = int(rng.integers(0, len(wl.table.wvls)))
            fld = wl.fields[fi]
            opts = oracle.make_opts(flags=flags, out_mode=abi.OUT_HITS_COMPACT, first_surf=1, last_surf=N - 2,
                                    foc=float(rng.uniform(-0.05, 0.05)), image_pt=wl.image_pts[fi])
            if trial % 2 == 0:
                num = int(rng.integers(1, 500))
                lo, hi = rng.uniform(-1.3, -0.5, 2), rng.uniform(0.5, 1.3, 2)
                grid = oracle.make_grid(lo, hi, num)
                want = oracle.trace_pupil_grid(wl.table, fld, grid, wi, opts).hits
                got = eng.trace_pupil_grid_hits(fld, grid, wi, opts)
            else:
                R = int(rng.integers(0, 200000))
                px, py = rng.uniform(-1.2, 1.2, R), rng.uniform(-1.2, 1.2, R)
                want = oracle.trace_pupil_list(wl.table, fld, px, py, wi, opts).hits
                got = eng.trace_pupil_list_hits(fld, px, py, wi, opts)
            n += 1
            bad += not (got.shape == want.shape and np.array_equal(got, want))
            # host-pointer staging around the pinned-block / arena switch
            num = int(rng.integers(1, 260))
            grid = oracle.make_grid((-1., -1.), (1., 1.), num)
            mode = [abi.OUT_FULL, abi.OUT_LAST, abi.OUT_HITS][trial % 3]
            o = oracle.make_opts(flags=flags, out_mode=mode, first_surf=1, last_surf=N - 2, foc=0.01,
                                 image_pt=wl.image_pts[fi])
            orc = oracle.trace_pupil_grid(wl.table, fld, grid, wi, o)
            o.flags |= abi.HOST_POINTERS
            res = oracle.HostResult(N, num * num, mode, want_pupil=True)
            res.seg[:] = 1.0
            out = res.out_struct()
            rc = eng.lib.rox_trace_pupil_grid(eng._handle, C.byref(fld), C.byref(grid), wi, C.byref(o),
                                              C.byref(out), None)
            n_hp += 1
            bad_hp += not (rc == 0 and np.array_equal(res.status, orc.status)
                           and np.array_equal(res.seg, orc.seg, equal_nan=True)
                           and np.array_equal(res.op, orc.op, equal_nan=True))
        # rox_trace_pupil_grids: random item counts and grid sizes, packed hits and FULL packets,
        # every item against the single-grid entry on the device (itself soaked against the
        # oracle above) -- the per-item tickets / look-back states and the item slots re-armed
        # between back-to-back batches of different shapes
        for trial in range(trials // 6):
            k = int(rng.integers(1, 24))
            num = int(rng.integers(1, 200)) if trial % 4 else int(rng.integers(200, 700))
            pairs = [(int(rng.integers(0, len(wl.fields))), int(rng.integers(0, len(wl.table.wvls))))
                     for _ in range(k)]
            grid = oracle.make_grid(rng.uniform(-1.3, -0.5, 2), rng.uniform(0.5, 1.3, 2), num)
            mode = abi.OUT_HITS_COMPACT if trial % 3 else abi.OUT_FULL
            optl = [oracle.make_opts(flags=flags, out_mode=mode, first_surf=1, last_surf=N - 2,
                                     foc=float(rng.uniform(-0.05, 0.05)), image_pt=wl.image_pts[fi])
                    for fi, _wi in pairs]
            flds = [wl.fields[fi] for fi, _wi in pairs]
            wis = [wi_ for _fi, wi_ in pairs]
            if mode == abi.OUT_HITS_COMPACT:
                got = eng.trace_pupil_grids_hits(flds, wis, grid, optl)
                got = [g.copy() for g in got]
                for f, w_, o, g in zip(flds, wis, optl, got):
                    one = eng.trace_pupil_grid_hits(f, grid, w_, o)
                    n_b += 1
                    bad_b += not (one.shape == g.shape and np.array_equal(one, g))
            else:
                res = eng.trace_pupil_grids(flds, wis, grid, optl, nan_fill=True)
                for f, w_, o, r in zip(flds, wis, optl, res):
                    a = r.to_host()
                    b = eng.trace_pupil_grid(f, grid, w_, o, nan_fill=True).to_host()
                    n_b += 1
                    bad_b += not (np.array_equal(a.status, b.status) and np.array_equal(a.seg, b.seg, equal_nan=True)
                                  and np.array_equal(a.op, b.op, equal_nan=True))
        eng.close()
    print(json.dumps({'compact_launches': n, 'compact_mismatching': bad, 'host_pointer_calls': n_hp,
                      'host_pointer_mismatching': bad_hp, 'batched_items': n_b, 'batched_mismatching': bad_b,
                      'seconds': round(time.time() - t0, 1)}))


if __name__ == '__main__':
    main()
