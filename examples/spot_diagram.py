#!/usr/bin/env python3
"""Two ways to use the engine.

(1) stand-alone (no ray-optics installed): a stored workload -> spot diagram
    python examples/spot_diagram.py

(2) behind ray-optics (needs `rayoptics` importable *and* a GPU):
    import rayoptics_amd.install as roxi
    roxi.install()                       # rebinds trace_grid / trace_fan / ...
    from rayoptics.environment import *
    opm = open_model('my_lens.roa')
    fig = plt.figure(FigureClass=SpotDiagramFigure, opt_model=opm, num_rays=1024).plot()
    # SpotDiagramFigure, RayFan, RayList, RayGrid and analyses.eval_wavefront now run on the
    # device; roxi.uninstall() restores the reference's own functions.
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np  # noqa: E402


def main(num=1024):
    import rayoptics_amd  # noqa: F401
    from rayoptics_amd import abi, workloads
    from rayoptics_amd.engine import TraceEngine, make_opts, make_grid

    wl = workloads.load('dblgauss_c2')          # 13-interface double Gauss, 3 fields, 3 wavelengths
    eng = TraceEngine(wl.table)
    N = wl.n_ifcs
    flags = abi.INTERSECT_OBJ | abi.CHECK_APERTURES | abi.APPLY_VIGNETTING
    grid = make_grid((-1., -1.), (1., 1.), num)
    for fi, fld in enumerate(wl.fields):
        for wi, wvl in enumerate(wl.table.wvls):
            opts = make_opts(flags=flags, out_mode=abi.OUT_HITS, first_surf=1, last_surf=N - 2,
                             foc=wl.foc, image_pt=wl.image_pts[fi])
            res = eng.trace_pupil_grid(fld, grid, wi, opts, want_pupil=False)
            ok = res.status == 0
            xy = res.seg[:, ok]                  # transverse aberrations of the rays that get through
            rms = float(xy.std(dim=1).norm())
            print(f'field {fi}  {wvl:6.1f} nm  {int(ok.sum()):8d} of {num * num} rays   '
                  f'rms spot radius {rms * 1e3:8.3f} um')
    eng.close()


if __name__ == '__main__':
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 1024)
