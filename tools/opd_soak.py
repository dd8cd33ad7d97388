#!/usr/bin/env python3
"""Device vs oracle soak of the OPD and FAN epilogues (finite and infinite reference spheres,
both infinite forms) on the golden OPD cases with perturbed field constants, focus shifts,
image points and pupil windows -- bit for bit.

    python tools/opd_soak.py [trials]"""
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
    from rayoptics_amd import abi
    from rayoptics_amd.engine import TraceEngine
    from oracle import oracle
    import helpers as H
    from test_oracle_golden import OPD_CASES, opd_opts
    trials = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    rng = np.random.default_rng(11)
    t0 = time.time()
    n = bad = 0
    engines = {}
    for name, case in OPD_CASES:
        fx = H.fixture(name)
        c = fx[case]
        eng = engines.setdefault(name, TraceEngine(fx.table))
        for trial in range(trials):
            fld = H.field_from_arr(c['field'])
            fld.pt0[1] *= rng.uniform(0.8, 1.1)
            fld.aim[1] += rng.uniform(-0.05, 0.05)
            o = opd_opts(c)
            o.out_mode = abi.OUT_OPD if trial % 2 == 0 else abi.OUT_FAN
            o.foc = float(rng.uniform(-0.05, 0.05))
            o.image_pt[0], o.image_pt[1] = rng.uniform(-0.1, 0.1), rng.uniform(-0.1, 0.1)
            if o.wf.kind != abi.WF_FINITE and trial % 3 == 0:
                o.wf.kind = abi.WF_INF_SPLIT if o.wf.kind == abi.WF_INF_FULL else abi.WF_INF_FULL
            a, b = np.array(c['start']) * rng.uniform(0.7, 1.2), np.array(c['stop']) * rng.uniform(0.7, 1.2)
            grid = oracle.make_grid(a, b, int(rng.integers(5, 40)))
            wi = int(rng.integers(0, len(fx.table.wvls)))
            with np.errstate(all='ignore'):
                orc = oracle.trace_pupil_grid(fx.table, fld, grid, wi, o)
            dev = eng.trace_pupil_grid(fld, grid, wi, o, nan_fill=True).to_host()
            same = (np.array_equal(dev.status, orc.status)
                    and np.array_equal(np.asarray(dev.seg).reshape(np.asarray(orc.seg).shape), orc.seg,
                                       equal_nan=True))
            n += 1
            bad += not same
    for e in engines.values():
        e.close()
    print(json.dumps({'cases': len(OPD_CASES), 'launches': n, 'mismatching': bad,
                      'seconds': round(time.time() - t0, 1)}))


if __name__ == '__main__':
    main()
