#!/usr/bin/env python3
"""What a ray-optics user sees: the reference's own figure classes and model update, timed in
one process with and without `rayoptics_amd.install()` -- the reference's per-ray Python loop
vs the same consumers served by libroxtrace.so -- and the data compared.  Needs the live
reference next to the GPU (oracle/_ref, oracle/stage_reference.py).

    python tools/figure_latency.py > profiles/r04_figure_latency.jsonl"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))

import numpy as np  # noqa: E402


def timed(fn, reps):
    ts, out = [], None
    for _ in range(reps):
        t0 = time.perf_counter()
        out = fn()
        ts.append((time.perf_counter() - t0) * 1e3)
    return float(np.median(ts)), out


def flat(x):
    """every number of a nested figure-data structure, in order"""
    if x is None:
        return [np.nan]
    if isinstance(x, str):          # (line colours ride along in the fan data)
        return []
    if isinstance(x, (list, tuple)):
        return [v for e in x for v in flat(e)]
    return np.asarray(x, dtype=float).ravel().tolist()


def main():
    import matplotlib
    matplotlib.use('Agg')
    import matplotlib.pyplot as plt
    import refmodels as ref
    import rayoptics_amd  # noqa: F401
    from rayoptics_amd import install
    from rayoptics.mpl.axisarrayfigure import SpotDiagramFigure, RayFanFigure
    from rayoptics.raytr import analyses

    # a figure is built once (with the drop-ins active, so that the constructor's own first
    # update_data is quick) and its refresh -- update_data() on the existing figure, what a
    # model edit triggers -- is what is timed, without and with the drop-ins
    figs = []

    def figure(cls, opm, **kw):
        install.install()
        try:
            fig = plt.figure(FigureClass=cls, opt_model=opm, **kw)
        finally:
            install.uninstall()
        figs.append(fig)
        return fig

    def spot(opm, num):
        fig = figure(SpotDiagramFigure, opm, num_rays=num)

        def run():
            fig.update_data()
            return [[np.array(g) for g in row[0][0]] for row in fig.axis_data_array]
        return run

    def fan(opm, data_type, num):
        fig = figure(RayFanFigure, opm, data_type=data_type, num_rays=num)

        def run():
            fig.update_data()
            return flat(fig.axis_data_array)        # (numbers only: the figure rebuilds its lists)
        return run

    def wavefront(opm, num):
        # (WavefrontFigure itself raises in the reference -- it hands the wavelength index
        # where the wavelength is wanted -- so: the RayGrid its panels are made of)
        def run():
            nf = len(opm['osp']['fov'].fields)
            return [analyses.RayGrid(opm, f=f, num_rays=num).grid for f in range(nf)]
        return run

    def update(opm):
        # (OpticalModel.update_model minus the element tree, which needs packages the shim
        # stubs: what tests/golden/refmodels.finish does -- chief-ray aiming of every field,
        # the vignetting search, the boundary rays of set_clear_apertures)
        def run():
            ref.finish(opm)
            osp = opm['osp']
            return [[f.aim_info if f.aim_info is not None else [np.nan, np.nan], f.vux, f.vlx, f.vuy, f.vly]
                    for f in osp['fov'].fields]
        return run

    cases = []
    for model in ('dblgauss', 'nikkor', 'cell_phone'):
        opm = getattr(ref, model)()
        n_fw = len(opm['osp']['fov'].fields) * len(opm['osp']['wvls'].wavelengths)
        cases += [(model, 'SpotDiagramFigure.update_data', {'num_rays': 21, 'rays': 21 * 21 * n_fw}, spot(opm, 21), 3),
                  (model, 'SpotDiagramFigure.update_data', {'num_rays': 64, 'rays': 64 * 64 * n_fw}, spot(opm, 64), 1),
                  (model, 'RayFanFigure(Ray).update_data', {'num_rays': 21}, fan(opm, 'Ray', 21), 3),
                  (model, 'RayFanFigure(OPD).update_data', {'num_rays': 21}, fan(opm, 'OPD', 21), 3),
                  (model, 'model update: sm.update_model, osp.update_model, update_optical_properties (aiming, vignetting, clear apertures)', {}, update(opm), 3)]
        if model != 'cell_phone':
            cases.append((model, 'analyses.RayGrid of every field (WavefrontFigure\'s data)', {'num_rays': 32}, wavefront(opm, 32), 1))
    for model, what, extra, run, reps in cases:
        rec = {'model': model, 'what': what, **extra}
        try:
            ms_ref, theirs = timed(run, reps)
            err_ref = None
        except Exception as e:
            ms_ref, theirs, err_ref = None, None, repr(e)
        install.install()
        try:
            run()                   # first call: engine + table
            ms_dev, ours = timed(run, max(5, 3 * reps))
            err_dev = None
        except Exception as e:
            ms_dev, ours, err_dev = None, None, repr(e)
        finally:
            install.uninstall()
        rec['reference_ms'], rec['drop_in_ms'] = ms_ref, ms_dev
        if err_ref or err_dev:
            rec['reference_raises'], rec['drop_in_raises'] = err_ref, err_dev
        else:
            a, b = np.array(flat(ours)), np.array(flat(theirs))
            rec['numbers_compared'] = int(a.size)
            rec['bit_identical'] = bool(a.shape == b.shape and np.array_equal(a, b, equal_nan=True))
            if not rec['bit_identical'] and a.shape == b.shape:
                ok = np.isfinite(a) & np.isfinite(b)
                rec['max_abs_diff'] = float(np.abs(a[ok] - b[ok]).max()) if ok.any() else None
            rec['speedup'] = ms_ref / ms_dev if ms_dev else None
        print(json.dumps(rec), flush=True)


def big():
    """figure sizes the reference cannot be waited for: the drop-ins alone, the reference's time
    extrapolated from its measured rays per second at num_rays = 64 (it is a per-ray loop)"""
    import matplotlib
    matplotlib.use('Agg')
    import matplotlib.pyplot as plt
    import refmodels as ref
    import rayoptics_amd  # noqa: F401
    from rayoptics_amd import install
    from rayoptics.mpl.axisarrayfigure import SpotDiagramFigure
    for model in ('dblgauss', 'nikkor'):
        opm = getattr(ref, model)()
        n_fw = len(opm['osp']['fov'].fields) * len(opm['osp']['wvls'].wavelengths)

        def refresh(fig):
            fig.update_data()
            return sum(len(g) for row in fig.axis_data_array for g in row[0][0])
        install.install()
        fig64 = plt.figure(FigureClass=SpotDiagramFigure, opt_model=opm, num_rays=64)
        install.uninstall()
        t0 = time.perf_counter()
        refresh(fig64)
        ref_rays_per_s = 64 * 64 * n_fw / (time.perf_counter() - t0)
        plt.close(fig64)
        install.install()
        try:
            for num in (256, 512, 1024):
                fig = plt.figure(FigureClass=SpotDiagramFigure, opt_model=opm, num_rays=num)
                refresh(fig)
                ms, n = timed(lambda: refresh(fig), 7)
                plt.close(fig)
                print(json.dumps({'model': model, 'what': 'SpotDiagramFigure.update_data (refresh of an existing figure)', 'num_rays': num,
                                  'rays': num * num * n_fw, 'points_in_the_figure': int(n), 'drop_in_ms': ms,
                                  'reference_rays_per_s_measured_at_64': ref_rays_per_s,
                                  'reference_extrapolated_s': num * num * n_fw / ref_rays_per_s}), flush=True)
            from rayoptics.raytr import analyses
            for num in (256, 512, 1024):
                def grids():
                    return [analyses.RayGrid(opm, f=f, num_rays=num) for f in range(len(opm['osp']['fov'].fields))]
                grids()
                ms, gs = timed(grids, 5)
                rec = {'model': model, 'what': 'analyses.RayGrid of every field', 'num_rays': num,
                       'rays': num * num * len(gs), 'drop_in_ms': ms,
                       'reference_extrapolated_s': num * num * len(gs) / ref_rays_per_s}
                g0 = gs[0]
                ms_psf, psf = timed(lambda: analyses.calc_psf(g0.grid[2], num, 2 * num), 5)
                rec['calc_psf_ndim_maxdim'] = [num, 2 * num]
                rec['calc_psf_ms'] = ms_psf
                rec['psf_peak'] = float(np.nanmax(psf))
                print(json.dumps(rec), flush=True)
        finally:
            install.uninstall()


if __name__ == '__main__':
    if '--big' in sys.argv:
        big()
    else:
        main()
