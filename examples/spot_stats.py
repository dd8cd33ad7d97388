#!/usr/bin/env python3
"""Spot statistics without the spot leaving the GPU, and the opt-in tolerance mode.

    python examples/spot_stats.py [num]

For every field of a stored workload: the rays of a num x num pupil grid traced to the image plane
(`trace.trace_grid_spot_stats`: one ROX_OUT_HITS launch + `rox_spot_stats`), reduced on the device
to count / centroid / RMS radius / extent and a 64 x 64 histogram over RayGeoPSF's 'fit' edges --
72 bytes and the histogram cross PCIe instead of 16 bytes per ray.  Then the same call after
`session.set_tolerance_mode(True)`: kernels that stay within 1e-10 of the reference instead of
reproducing it bit for bit, about twice as fast (include/roxtrace.h ROX_FAST_FP64)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np  # noqa: E402


def main(num=512):
    import rayoptics_amd  # noqa: F401
    from rayoptics_amd import session, trace, workloads

    wl = workloads.load('dblgauss_c2')          # 13-interface double Gauss, 3 fields, 3 wavelengths
    model = workloads.TableModel(wl)            # (behind ray-optics: the OpticalModel itself)
    grid_rng = [np.array([-1., -1.]), np.array([1., 1.]), num]
    wvl = wl.table.wvls[wl.ref_wvl_idx]

    def call(fi):
        return trace.trace_grid_spot_stats(model, grid_rng, model.fields[fi], wvl, wl.foc,
                                           wl.image_pts[fi], bins=65)

    for mode in ('bit-exact', 'tolerance mode'):
        was = session.set_tolerance_mode(mode != 'bit-exact')
        try:
            for fi in range(len(model.fields)):
                call(fi)                        # (first call: buffers)
                t0 = time.perf_counter()
                summ, hist, x_edges, y_edges = call(fi)
                ms = (time.perf_counter() - t0) * 1e3
                peak = np.unravel_index(int(hist.argmax()), hist.shape)
                print(f'{mode:15s} field {fi}  {summ["n"]:8d} of {num * num} rays   '
                      f'centroid ({summ["centroid"][0] * 1e3:+8.3f}, {summ["centroid"][1] * 1e3:+8.3f}) um   '
                      f'rms spot radius {summ["rms_radius"] * 1e3:8.3f} um   '
                      f'fullest bin {int(hist.max()):6d} at ({int(peak[0])}, {int(peak[1])})   {ms:6.3f} ms')
        finally:
            session.set_tolerance_mode(was)


if __name__ == '__main__':
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 512)
