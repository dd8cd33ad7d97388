#!/usr/bin/env python3
"""Wavefront map and point spread function of one field, entirely on the device:
pupil-grid trace with the OPD epilogue (ROX_OUT_OPD) -> rox_calc_psf (pruned DFT on the
fp64 matrix cores).  Stand-alone: the double Gauss table and the per-field constants come
from a stored fixture (tests/golden/psf.npz); behind ray-optics the same two calls are
analyses.eval_wavefront and analyses.calc_psf after `rayoptics_amd.install.install()`.

    python examples/wavefront_psf.py [ndim] [maxdim]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402


def main(ndim=64, maxdim=256):
    import rayoptics_amd  # noqa: F401
    from rayoptics_amd import SurfaceTable, analyses, workloads
    from rayoptics_amd.table import field_struct, wavefront_from_array
    z = np.load(os.path.join(ROOT, 'tests', 'golden', 'psf.npz'))
    tbl = SurfaceTable.from_dict(json.loads(str(z['dblgauss_table_json'])))
    for tag in ('dblgauss_f0', 'dblgauss_f2'):
        a = z[f'{tag}/field']
        fld = field_struct(a[0:3], a[3:5], a[5], a[6], a[7:11], a[11])
        wi = int(z[f'{tag}/wvl_idx'])
        m = workloads.TableModel(workloads.SimpleWorkload(tbl, [fld], [(0., 0.)]))
        m.fields[0].rox_wavefront = wavefront_from_array(z[f'{tag}/wavefront'])
        m.fields[0]._vig_bbox = (z[f'{tag}/bbox'][0], z[f'{tag}/bbox'][1])
        m._units_per_nm = 1.0 / (float(z[f'{tag}/convert_to_opd']) * tbl.wvls[wi])
        grid = analyses.eval_wavefront(m, m.fields[0], tbl.wvls[wi], 0.0, num_rays=ndim)
        opd = np.rollaxis(grid, 2)[2]                       # waves; NaN outside the pupil
        ok = ~np.isnan(opd)
        rms = float(np.sqrt(np.mean((opd[ok] - opd[ok].mean()) ** 2)))
        psf = analyses.calc_psf(opd, ndim, maxdim)
        ideal = analyses.calc_psf(np.where(ok, 1e-300, np.nan), ndim, maxdim)   # same pupil, flat phase
        # both are normalised to their own peak: compare energy-normalised peaks for the Strehl ratio
        strehl = float((1.0 / psf.sum()) / (1.0 / ideal.sum()))
        print(f'{tag}: {tbl.wvls[wi]:6.1f} nm  {int(ok.sum())} of {ndim * ndim} rays  '
              f'rms OPD {rms:6.3f} waves  Strehl ~ {strehl:5.3f}  (exp(-(2 pi rms)^2) = '
              f'{np.exp(-(2 * np.pi * rms) ** 2):5.3f})')


if __name__ == '__main__':
    main(*(int(a) for a in sys.argv[1:3]))
