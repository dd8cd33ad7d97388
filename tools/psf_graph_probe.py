#!/usr/bin/env python3
"""Would replaying rox_calc_psf's five stream operations (a memset and four launches) as one
hipGraph help at figure sizes?  Captured through torch.cuda.CUDAGraph around the device-pointer
form of the call.  Measured (profiles/r03_psf_graph_probe.jsonl): no -- the replay is slower than
the plain launches on this stack, (64, 256): 0.042 vs 0.035 ms.

    python tools/psf_graph_probe.py"""
import sys, time, json
sys.path.insert(0, '/root/repo')
import numpy as np, torch
import rayoptics_amd
from rayoptics_amd.engine import calc_psf
for ndim, maxdim in ((64, 256), (128, 512), (32, 128)):
    y, x = np.mgrid[-1:1:ndim * 1j, -1:1:ndim * 1j]
    opd = 1.5 * (x * x + y * y) + 0.4 * x * y * y
    opd[x * x + y * y > 1.0] = np.nan
    d = torch.from_numpy(opd).cuda()
    for _ in range(5):
        ref = calc_psf(d, ndim, maxdim)
    torch.cuda.synchronize()
    def bench(fn, reps=300):
        for _ in range(20): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps
    plain = bench(lambda: calc_psf(d, ndim, maxdim))
    s = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    try:
        with torch.cuda.stream(s):
            for _ in range(3): out = calc_psf(d, ndim, maxdim)
            torch.cuda.synchronize()
            with torch.cuda.graph(g, stream=s):
                out = calc_psf(d, ndim, maxdim)
        torch.cuda.synchronize()
        graph = bench(lambda: g.replay())
        same = bool(torch.equal(out, ref))
    except Exception as e:
        graph, same = None, repr(e)
    print(json.dumps({'ndim': ndim, 'maxdim': maxdim, 'plain_ms': plain, 'graph_replay_ms': graph, 'same': same}))
