#!/usr/bin/env python3
"""How many Spencer-Murty steps (rayoptics/elem/profiles.py:155-186) the asphere
intersections of the asphere workloads take -- the source of lane divergence in
the polynomial-profile kernel instances.  Counted by the CPU oracle (the device
executes the same iteration, bit for bit) over a 256 x 256 pupil grid.

    python tools/newton_histogram.py > profiles/r02_newton_histogram.json"""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import rayoptics_amd  # noqa: F401
    from rayoptics_amd import abi, workloads
    from oracle import oracle
    lib = oracle.lib()
    lib.rox_oracle_newton_histogram.argtypes = [C.POINTER(C.c_longlong), C.c_int]
    out = {}
    for name in ('cell_phone', 'nikkor_c3'):
        wl = workloads.load(name)
        N = wl.n_ifcs
        h = (C.c_longlong * 16)()
        lib.rox_oracle_newton_histogram(h, 1)
        opts = oracle.make_opts(flags=abi.INTERSECT_OBJ | abi.CHECK_APERTURES | abi.APPLY_VIGNETTING,
                                out_mode=abi.OUT_HITS, first_surf=1, last_surf=N - 2)
        rec = {}
        for fi in range(len(wl.fields)):
            oracle.trace_pupil_grid(wl.table, wl.fields[fi], oracle.make_grid((-1., -1.), (1., 1.), 256),
                                    wl.ref_wvl_idx, opts)
        lib.rox_oracle_newton_histogram(h, 1)
        tot = sum(h)
        rec['asphere_hits'] = tot
        rec['steps_after_the_first_evaluation'] = {str(i): h[i] for i in range(16) if h[i]}
        rec['fraction'] = {str(i): round(h[i] / tot, 4) for i in range(16) if h[i]}
        rec['mean_steps'] = sum(i * h[i] for i in range(16)) / tot
        out[name] = rec
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
