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
    # every (field, wavelength) spot of the lens in ONE launch (rox_trace_pupil_grids): the rays
    # that get through come back packed in ray order, one (R_ok, 2) host array per spot --
    # what SequentialModel.trace_grid(spot, fi, wl, num_rays, form='list') returns
    pairs = [(fi, wi) for fi in range(len(wl.fields)) for wi in range(len(wl.table.wvls))]
    opts = [make_opts(flags=flags, out_mode=abi.OUT_HITS_COMPACT, first_surf=1, last_surf=N - 2,
                      foc=wl.foc, image_pt=wl.image_pts[fi]) for fi, _wi in pairs]
    spots = eng.trace_pupil_grids_hits([wl.fields[fi] for fi, _wi in pairs], [wi for _fi, wi in pairs],
                                       grid, opts)
    for (fi, wi), xy in zip(pairs, spots):
        rms = float(np.linalg.norm(xy.std(axis=0)))
        print(f'field {fi}  {wl.table.wvls[wi]:6.1f} nm  {len(xy):8d} of {num * num} rays   '
              f'rms spot radius {rms * 1e3:8.3f} um')
    eng.close()


if __name__ == '__main__':
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 1024)
