#!/usr/bin/env python3
"""rox_calc_psf (csrc/psf.hip: pruned DFT as two complex fp64 GEMMs on the matrix
cores) -- time per call with the OPD grid and the PSF resident in HBM, and through
plain host arrays (ROX_HOST_POINTERS).  flop = 8 M n (n + M) (4 real multiply-adds per
complex one, both GEMMs); peak = 78.6 TFLOP/s fp64 (matrix = vector rate on MI355X).
The reference's calc_psf (Python loop over maxdim^2 + numpy fft2) takes 28 / 143 / 621 ms
for (64, 256) / (128, 512) / (256, 1024) in the build container.

    python tools/psf_bench.py > profiles/r02_psf.jsonl"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import rayoptics_amd  # noqa: F401
    from rayoptics_amd.engine import calc_psf
    sizes = ((64, 256), (128, 512), (256, 1024), (512, 2048), (1024, 4096))
    if os.environ.get('PSF_ONLY_LARGE'):           # PMC passes: one size, few launches
        sizes = ((1024, 4096),)
    if os.environ.get('PSF_SIZES'):                # e.g. PSF_SIZES=64x256,128x512
        sizes = tuple(tuple(int(v) for v in t.split('x')) for t in os.environ['PSF_SIZES'].split(','))
    for ndim, maxdim in sizes:
        y, x = np.mgrid[-1:1:ndim * 1j, -1:1:ndim * 1j]
        opd = 1.5 * (x * x + y * y) + 0.4 * x * y * y
        opd[x * x + y * y > 1.0] = np.nan
        d = torch.from_numpy(opd).cuda()
        for _ in range(5):
            calc_psf(d, ndim, maxdim)
        torch.cuda.synchronize()
        reps = 200 if maxdim <= 1024 else 30
        if os.environ.get('PSF_ONLY_LARGE'):
            reps = 3
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            calc_psf(d, ndim, maxdim)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        th = []
        for _ in range(12):
            t0 = time.perf_counter()
            calc_psf(opd, ndim, maxdim)
            th.append(time.perf_counter() - t0)
        flop = 8.0 * maxdim * ndim * (ndim + maxdim)
        print(json.dumps({'ndim': ndim, 'maxdim': maxdim, 'device_ms': ms,
                          'host_arrays_ms': float(np.median(th[2:]) * 1e3),
                          'flop': flop, 'tflops': flop / (ms * 1e-3) / 1e12,
                          'frac_of_78.6': flop / (ms * 1e-3) / 78.6e12}))


if __name__ == '__main__':
    main()
