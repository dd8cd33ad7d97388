#!/usr/bin/env python3
"""Latency of ONE ray through the product's single-ray entry
(rayoptics_amd.trace.raytrace_trace, the function rayoptics.raytr.raytrace.trace
is rebound to): Python call -> list-of-segments result.  The reference's own
rt.trace takes ~250 us per ray on the double Gauss (profiles/reference_cpu.json).

    python tools/single_ray_latency.py [workload] [n]  > profiles/r02_single_ray.json"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import rayoptics_amd  # noqa: F401
    from rayoptics_amd import abi, session, trace, workloads
    from rayoptics_amd.engine import make_opts
    name = sys.argv[1] if len(sys.argv) > 1 else 'dblgauss_c2'
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
    m = workloads.TableModel(name)
    sm = m['seq_model']
    tbl = sm.surface_table
    wvl = sm.central_wavelength()
    f = m.fields[min(1, len(m.fields) - 1)].rox_field
    # a fan of rays through the pupil of that field (host restatement of the epd ray start
    # is not needed: take pt0 of the field and aim at pupil points)
    rng = np.random.default_rng(0)
    pt0 = np.array([f.pt0[0], f.pt0[1], f.pt0[2]])
    rays = []
    for _ in range(64):
        px, py = rng.uniform(-0.6, 0.6, 2)
        p1 = np.array([f.eprad * px + f.aim[0], f.eprad * py + f.aim[1], f.z_enp])
        d = p1 - pt0
        rays.append((pt0.copy(), d / np.linalg.norm(d)))
    for k in range(200):                                    # warm: engine, pool, clocks
        trace.raytrace_trace(sm, *rays[k % 64], wvl)
    t = np.empty(n)
    for k in range(n):
        p, d = rays[k % 64]
        t0 = time.perf_counter()
        trace.raytrace_trace(sm, p, d, wvl)
        t[k] = time.perf_counter() - t0
    eng = session.engine_for(m)
    opts = make_opts(flags=abi.INTERSECT_OBJ, first_surf=1, last_surf=tbl.n_ifcs - 2)
    e = np.empty(n)
    for k in range(n):
        p, d = rays[k % 64]
        t0 = time.perf_counter()
        eng.trace_one(p, d, 0, opts)
        e[k] = time.perf_counter() - t0
    out = {'workload': name, 'n_ifcs': tbl.n_ifcs, 'calls': n,
           'what': 'rayoptics_amd.trace.raytrace_trace(seq_model, pt0, dir0, wvl): Python call -> '
                   '(list of [p, d, dst, nrml], op, wvl); launch + synchronise + result objects',
           'us_median': float(np.median(t) * 1e6), 'us_p10': float(np.percentile(t, 10) * 1e6),
           'us_p90': float(np.percentile(t, 90) * 1e6),
           'engine_trace_one_us_median': float(np.median(e) * 1e6),
           'rays_per_s': float(1.0 / np.median(t))}
    print(json.dumps(out))


if __name__ == '__main__':
    main()
