#!/usr/bin/env python3
"""Wall-clock of ROX_HOST_POINTERS calls (plain NumPy buffers through the C ABI,
as the maintainer stub of INTEGRATION.md makes them): the PCIe- and
host-memory-inclusive rate of the boundary when the caller hands over host memory.

    python tools/host_pointer_latency.py > profiles/r02_host_pointers.jsonl"""
import ctypes as C
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
    wl = workloads.load('dblgauss_c2')
    N = wl.n_ifcs
    eng = TraceEngine(wl.table)
    lib = eng.lib
    fld = wl.fields[1]
    flags = abi.INTERSECT_OBJ | abi.CHECK_APERTURES | abi.APPLY_VIGNETTING | abi.HOST_POINTERS
    for mode, name in ((abi.OUT_FULL, 'FULL'), (abi.OUT_HITS, 'HITS')):
        for num in (1, 8, 32, 128, 256, 1024):
            R = num * num
            rows = N * 10 if mode == abi.OUT_FULL else 2
            if rows * R * 8 > (3 << 30):
                continue
            seg = np.empty((rows, R))
            op = np.empty(R)
            status = np.empty(R, np.uint8)
            fail = np.empty(R, np.int16)
            o = abi.Out()
            o.seg, o.op, o.status, o.fail_surf, o.ld = (seg.ctypes.data, op.ctypes.data,
                                                       status.ctypes.data, fail.ctypes.data, R)
            opts = make_opts(flags=flags, out_mode=mode, first_surf=1, last_surf=N - 2,
                             foc=wl.foc, image_pt=wl.image_pts[1])
            grid = make_grid((-1., -1.), (1., 1.), num)
            reps = 200 if R <= 16384 else (20 if R <= 65536 else 5)
            ts = []
            for k in range(reps + 3):
                t0 = time.perf_counter()
                rc = lib.rox_trace_pupil_grid(eng._handle, C.byref(fld), C.byref(grid),
                                              wl.ref_wvl_idx, C.byref(opts), C.byref(o), None)
                ts.append(time.perf_counter() - t0)
                assert rc == 0, lib.rox_last_error()
            t = float(np.median(ts[3:]))
            nbytes = rows * R * 8 + R * 11
            print(json.dumps({'mode': name, 'rays': R, 'bytes_out': nbytes, 'ms': t * 1e3,
                              'rays_per_s': R / t, 'GBps_to_host': nbytes / t / 1e9}))
    eng.close()


if __name__ == '__main__':
    main()
