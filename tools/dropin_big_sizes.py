#!/usr/bin/env python3
"""The reference's consumers through the drop-ins at sizes far beyond what its own per-ray loop
is used with: a sanity sweep for host-side pathologies (pools, per-ray Python, copies) that the
kernel benchmarks cannot see.  GPU box, staged reference.  ms per call, second call timed."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))

import numpy as np  # noqa: E402


def ms_of(fn, reps=3):
    fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        out = fn()
        ts.append((time.perf_counter() - t0) * 1e3)
    return float(np.median(ts)), out


class _Box:
    __slots__ = ('p', 'pkg')

    def __init__(self, p, pkg):
        self.p, self.pkg = p, pkg


def main():
    import refmodels as ref
    import rayoptics_amd  # noqa: F401
    from rayoptics_amd import install
    import rayoptics.raytr.trace as trace
    from rayoptics.raytr import analyses
    opm = ref.dblgauss()
    osp = opm['osp']
    fld, wvl, foc = osp.lookup_fld_wvl_focus(1)
    install.install()
    try:
        for num in (64, 256, 512):
            rng = [np.array([-1., -1.]), np.array([1., 1.]), num]
            # (no img_filter -- entries [x, y, RayPkg] -- ends in ValueError at np.array(grid) under
            # NumPy >= 1.24, in the reference and in the drop-in alike: an object-valued filter)
            ms, g = ms_of(lambda: trace.trace_grid(opm, rng, fld, wvl, foc, form='grid', append_if_none=True,
                                                   img_filter=lambda p, pkg: _Box(p, pkg)))
            print(json.dumps({'what': "trace.trace_grid(form='grid'), object-valued filter: a [num, num] object array of (pupil, RayPkg) boxes",
                              'num': num, 'rays': num * num, 'ms': ms, 'us_per_ray': ms * 1e3 / (num * num)}), flush=True)
        for n in (1000, 100000, 1000000):
            pts = np.random.default_rng(1).uniform(-1, 1, (n, 2))
            ms, rl = ms_of(lambda: analyses.trace_ray_list(opm, pts, fld, wvl, foc, append_if_none=True))
            print(json.dumps({'what': 'analyses.trace_ray_list (explicit pupil coordinates)', 'rays': n, 'ms': ms,
                              'us_per_ray': ms * 1e3 / n}), flush=True)
        for num in (21, 2001, 200001):
            ms, f = ms_of(lambda: analyses.RayFan(opm, f=1, xyfan='y', num_rays=num))
            print(json.dumps({'what': "analyses.RayFan(xyfan='y')", 'num_rays': num, 'ms': ms}), flush=True)
        for num in (1000, 100000):
            rays = [(np.array([0., 0.01 * k / num, 0.]), np.array([0., 0., 1.]), wvl) for k in range(num)]
            ms, out = ms_of(lambda: analyses.trace_list_of_rays(opm, rays))
            print(json.dumps({'what': 'analyses.trace_list_of_rays (explicit rays)', 'rays': num, 'ms': ms,
                              'us_per_ray': ms * 1e3 / num}), flush=True)
    finally:
        install.uninstall()


if __name__ == '__main__':
    main()
