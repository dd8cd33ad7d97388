#!/usr/bin/env python3
"""Soak of the device vs oracle comparison for phase elements (gratings / DOEs agree to one ulp:
the kernel squares exactly where the reference and the oracle call libm pow; holograms and thin
lenses bit for bit) over many random systems.

    python tools/phase_soak.py [seeds]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))


def main():
    import rayoptics_amd  # noqa: F401
    from rayoptics_amd import abi
    from rayoptics_amd.engine import TraceEngine
    from oracle import oracle
    import helpers as H
    n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    out = {}
    t0 = time.time()
    for kind in ('grating', 'doe', 'hologram', 'thinlens'):
        worst_status, worst_rel, frac_exact, n_rays = 1.0, 0.0, 1.0, 0
        for seed in range(n_seeds):
            rng = np.random.default_rng(4200 + 10 * seed + len(kind))
            tbl, _k = H.phase_table(rng, kind)
            N = tbl.n_ifcs
            R = 4096
            pt0, d = H.random_rays(rng, R, tbl.rows[0].t[2])
            wi = (np.arange(R) % len(tbl.wvls)).astype(np.int32)
            eng = TraceEngine(tbl)
            opts = oracle.make_opts(flags=abi.INTERSECT_OBJ | abi.CHECK_APERTURES, first_surf=1,
                                    last_surf=N - 2)
            with np.errstate(all='ignore'):
                orc = oracle.trace_rays(tbl, pt0, d, wi, opts)
            dev = eng.trace_rays(pt0, d, wi, opts, nan_fill=True).to_host()
            eng.close()
            agree = dev.status == orc.status
            worst_status = min(worst_status, float(agree.mean()))
            a, b = orc.seg[..., agree], dev.seg[..., agree]
            fin = np.isfinite(a) & np.isfinite(b)
            assert np.array_equal(np.isnan(a), np.isnan(b))
            rel = np.abs(a[fin] - b[fin]) / np.maximum(1.0, np.abs(a[fin]))
            worst_rel = max(worst_rel, float(rel.max()) if rel.size else 0.0)
            frac_exact = min(frac_exact, float((a[fin] == b[fin]).mean()) if rel.size else 1.0)
            n_rays += R
        out[kind] = {'systems': n_seeds, 'rays': n_rays, 'min_status_agreement': worst_status,
                     'max_rel_diff': worst_rel, 'min_fraction_bit_identical': frac_exact}
    out['seconds'] = round(time.time() - t0, 1)
    print(json.dumps(out))


if __name__ == '__main__':
    main()
