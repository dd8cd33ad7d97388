#!/usr/bin/env python3
"""Soak of the tolerance-mode kernels (ROX_FAST_FP64) on tests/test_gpu_fuzz.py's random
prescriptions (every profile kind, mirrors, phantoms, aperture lists, tilts, phase elements,
per-ray wavelengths; rays incl. steep and degenerate ones): HITS and LAST in tolerance mode
against the oracle.

    python tools/fast_soak.py [first_seed] [count]

Per system: rays whose status / failing surface agree must agree in value to 1e-10 * max(1, |ref|)
unless the ray is ILL-CONDITIONED -- shown by tracing it again in the ORACLE with its direction
nudged by 4 ulp: where the reference's own answer moves by more than the tolerance under a
rounding-sized change of its input, no other arithmetic can be expected to stay within it
(grazing exits, rays a hair inside a TIR limit, 1e10-long lever arms).  Rays whose status differs
are counted; so are the well-conditioned rays beyond the tolerance (expected: none)."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

TOL = 1e-10


def scaled(ref, got):
    with np.errstate(all='ignore'):
        e = np.abs(ref - got) / np.maximum(1.0, np.abs(ref))
    e[np.isnan(ref) & np.isnan(got)] = 0.0
    e[np.isnan(e)] = np.inf
    return e


def main():
    import rayoptics_amd  # noqa: F401
    from rayoptics_amd import abi
    from rayoptics_amd.engine import TraceEngine
    from oracle import oracle
    import test_gpu_fuzz as t
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 500
    t0 = time.time()
    tot = dict(systems=0, rays=0, same_status=0, flips=0, beyond_tol=0, beyond_tol_ill_conditioned=0,
               beyond_tol_well_conditioned=0, worst_well_conditioned=0.0)
    worst_cases = []
    for seed in range(first, first + count):
        rng = np.random.default_rng(1000 + seed)
        tbl = t.random_table(rng)
        N = tbl.n_ifcs
        R = 3000 + int(rng.integers(0, 200))
        pt0, d = t.random_rays(rng, tbl, R)
        W = len(tbl.wvls)
        wi = rng.integers(0, W, R).astype(np.int32) if seed % 2 else int(rng.integers(0, W))
        eng = TraceEngine(tbl)
        for mode in (abi.OUT_LAST, abi.OUT_HITS):
            flags = (abi.INTERSECT_OBJ if seed % 5 else 0) | (abi.CHECK_APERTURES if seed % 3 else 0)
            kw = dict(out_mode=mode, first_surf=int(seed % 2), last_surf=(N - 2) if seed % 7 else -1,
                      foc=0.01 * (seed % 50), image_pt=(0.1, -0.2))
            o_ref = oracle.make_opts(flags=flags, **kw)
            o_fast = oracle.make_opts(flags=flags | abi.FAST_FP64, **kw)
            with np.errstate(all='ignore'):
                orc = oracle.trace_rays(tbl, pt0, d, wi, o_ref)
            dev = eng.trace_rays(pt0, d, wi, o_fast, nan_fill=True).to_host()
            same = (orc.status == dev.status) & (orc.fail_surf == dev.fail_surf)
            ok = same & (orc.status == abi.OK)
            err = np.zeros(R)
            if ok.any():
                err[ok] = scaled(orc.seg[:, ok], dev.seg[:, ok]).max(axis=0)
                err[ok] = np.maximum(err[ok], scaled(orc.op[ok], dev.op[ok]))
            bad = np.flatnonzero(err > TOL)
            tot['rays'] += R
            tot['same_status'] += int(same.sum())
            tot['flips'] += int((~same).sum())
            tot['beyond_tol'] += len(bad)
            if len(bad):
                # conditioning of those rays in the reference's own arithmetic
                d2 = d[:, bad].copy()
                for k in range(4):
                    d2[0] = np.nextafter(d2[0], np.inf)
                w2 = wi[bad] if isinstance(wi, np.ndarray) else wi
                with np.errstate(all='ignore'):
                    pert = oracle.trace_rays(tbl, pt0[:, bad].copy(), d2, w2, o_ref)
                moved = scaled(orc.seg[:, bad], pert.seg).max(axis=0)
                moved = np.maximum(moved, scaled(orc.op[bad], pert.op))
                moved[pert.status != abi.OK] = np.inf
                ill = moved > TOL / 16        # a 4-ulp nudge of ONE input already moves the answer by tol/16
                tot['beyond_tol_ill_conditioned'] += int(ill.sum())
                tot['beyond_tol_well_conditioned'] += int((~ill).sum())
                for j in np.flatnonzero(~ill)[:3]:
                    worst_cases.append({'seed': seed, 'mode': int(mode), 'ray': int(bad[j]),
                                        'err': float(err[bad[j]]), 'reference_moves_by': float(moved[j])})
            good = ok & (err <= TOL)
            if good.any():
                tot['worst_well_conditioned'] = max(tot['worst_well_conditioned'], float(err[good].max()))
        eng.close()
        tot['systems'] += 1
    tot.update(first_seed=first, modes=2, seconds=round(time.time() - t0, 1), tolerance=TOL,
               first_well_conditioned_outliers=worst_cases[:10])
    print(json.dumps(tot))


if __name__ == '__main__':
    main()
