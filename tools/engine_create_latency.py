#!/usr/bin/env python3
"""What a model edit costs before the first launch: a new table means a new handle
(rox_system_create -- the handle's table is immutable, like path_sequence's cache), so an
interactive caller pays this once per edit.  ms: TraceEngine(table) / first 64 x 64 spot / close.

    python tools/engine_create_latency.py"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import rayoptics_amd  # noqa: F401
    from rayoptics_amd import abi, workloads
    from rayoptics_amd.engine import TraceEngine, make_opts, make_grid
    for name in ('singlet_c1', 'dblgauss_c2', 'nikkor_c3', 'litho_c5'):
        wl = workloads.load(name)
        N = wl.n_ifcs
        opts = make_opts(flags=abi.INTERSECT_OBJ | abi.CHECK_APERTURES | abi.APPLY_VIGNETTING,
                         out_mode=abi.OUT_HITS_COMPACT, first_surf=1, last_surf=N - 2, foc=wl.foc,
                         image_pt=wl.image_pts[0])
        grid = make_grid((-1., -1.), (1., 1.), 64)
        TraceEngine(wl.table).close()
        tc, tf, tn, td = [], [], [], []
        for _ in range(30):
            t0 = time.perf_counter()
            eng = TraceEngine(wl.table)
            t1 = time.perf_counter()
            eng.trace_pupil_grid_hits(wl.fields[0], grid, 0, opts)
            t2 = time.perf_counter()
            eng.trace_pupil_grid_hits(wl.fields[0], grid, 0, opts)
            t3 = time.perf_counter()
            eng.close()
            t4 = time.perf_counter()
            tc.append(t1 - t0); tf.append(t2 - t1); tn.append(t3 - t2); td.append(t4 - t3)
        med = lambda v: float(np.median(v) * 1e3)       # noqa: E731
        print(json.dumps({'workload': name, 'interfaces': N, 'create_ms': med(tc),
                          'first_64x64_spot_ms': med(tf), 'next_64x64_spot_ms': med(tn),
                          'close_ms': med(td)}), flush=True)


if __name__ == '__main__':
    main()
