#!/usr/bin/env python3
"""Soak of the tolerance-mode kernels (ROX_FAST_FP64) on tests/test_gpu_fuzz.py's random
prescriptions (every profile kind, mirrors, phantoms, aperture lists, tilts, per-ray wavelengths;
rays incl. steep, backward and degenerate ones, objects 20 ... 1e10 away): LAST and HITS in
tolerance mode against the oracle.

    python tools/fast_soak.py [first_seed] [count]
    ROX_FAST_FP64_FULL=1 python tools/fast_soak.py [first_seed] [count] full     (FULL packets too)

`full`: ROX_OUT_FULL as a third mode -- every segment of every ray, partial records of failed rays
included.  The library sends FULL launches to the tolerance-mode kernels only for systems made
mostly of aspheres; ROX_FAST_FP64_FULL=1 (read once by the library) sends all of them there.

A ray DEVIATES when its status / failing surface differs from the oracle's, or a value differs by
more than 1e-10 * max(1, magnitude of the vector it belongs to).  Random prescriptions traced by
steep rays are full of rays whose answer the reference itself does not pin down: a Spencer-Murty
iteration started far from an asphere lands on another root after a rounding-sized change, a ray
leaves a hair inside a TIR limit, an intercept is 1e13 away.  Every deviating ray is therefore
traced again in the ORACLE -- FULL packets -- with its start point and direction changed by 1e-15
(relative; six random draws): if the reference's own packet changes status or moves anywhere by
more than a sixteenth of the tolerance under such a change, the ray is ILL-CONDITIONED -- no arithmetic other than the
reference's own, rounding for rounding, can be expected to reproduce it.  Rays that
pass this are traced once more, one by one, with the oracle logging its Spencer-Murty step counts
(rox_oracle_newton_log): an asphere intersection that takes more than eight steps (a regular one
takes 1-5), or that ends in a miss raised INSIDE the iteration, is an iteration that
WANDERS over the asphere before it lands -- which root it lands on (or whether an iterate
leaves the profile's domain first) is decided by roundings many steps upstream, far below any
input change that a 1e-15 perturbation test can resolve.  What is left after both -- deviating
although the reference's answer is stable and its iterations are short -- is counted as
`unexplained_deviations` (expected: 0) and listed."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

TOL = 1e-10
# rows of a LAST segment: p[3], d[3], dst, nrml[3]; of a HITS row pair: (x, y)
GROUPS = {1: (slice(0, 3), slice(3, 6), slice(6, 7), slice(7, 10)), 2: (slice(0, 2),)}


def scaled(ref, got, groups=None):
    """|ref - got| scaled by max(1, magnitude of the vector the component belongs to); inf where
    exactly one of the two is NaN"""
    with np.errstate(all='ignore'):
        mag = np.abs(ref)
        if groups is not None and ref.ndim == 2:
            mag = mag.copy()
            for sl in groups:
                blk = np.abs(ref[sl])
                blk = np.where(np.isnan(blk), 0.0, blk)
                mag[sl] = blk.max(axis=0, keepdims=True)
        e = np.abs(ref - got) / np.maximum(1.0, mag)
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
    with_full = len(sys.argv) > 3 and sys.argv[3] == 'full'
    modes = (abi.OUT_LAST, abi.OUT_HITS) + ((abi.OUT_FULL,) if with_full else ())
    t0 = time.time()
    tot = dict(systems=0, rays=0, agree=0, deviating=0, deviating_status=0, deviating_value=0,
               ill_conditioned=0, wandering_iteration=0, unexplained_deviations=0, worst_agreeing=0.0)
    import ctypes as C
    olib = oracle.lib()
    olib.rox_oracle_newton_log.restype = C.c_longlong
    olib.rox_oracle_newton_log.argtypes = [C.POINTER(C.c_byte), C.c_longlong]
    logbuf = np.zeros(4096, dtype=np.int8)
    listed = []
    prng = np.random.default_rng(first)
    for seed in range(first, first + count):
        rng = np.random.default_rng(1000 + seed)
        tbl = t.random_table(rng)
        N = tbl.n_ifcs
        R = 3000 + int(rng.integers(0, 200))
        pt0, d = t.random_rays(rng, tbl, R)
        W = len(tbl.wvls)
        wi = rng.integers(0, W, R).astype(np.int32) if seed % 2 else int(rng.integers(0, W))
        eng = TraceEngine(tbl)
        for mode in modes:
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
            if mode == abi.OUT_FULL:
                # every ray both end alike: the segments written (same NaN pattern), value by value
                ns = orc.seg.shape[0]
                g_full = tuple(slice(10 * k + a0, 10 * k + a1) for k in range(ns)
                               for a0, a1 in ((0, 3), (3, 6), (6, 7), (7, 10)))
                if same.any():
                    err[same] = scaled(orc.seg.reshape(ns * 10, -1)[:, same],
                                       dev.seg.reshape(ns * 10, -1)[:, same], g_full).max(axis=0)
                    err[ok] = np.maximum(err[ok], scaled(orc.op[ok], dev.op[ok]))
            elif ok.any():
                err[ok] = scaled(orc.seg[:, ok], dev.seg[:, ok], GROUPS[int(mode)]).max(axis=0)
                err[ok] = np.maximum(err[ok], scaled(orc.op[ok], dev.op[ok]))
            dev_mask = ~same | (err > TOL)
            bad = np.flatnonzero(dev_mask)
            tot['rays'] += R
            tot['agree'] += int(R - len(bad))
            tot['deviating'] += len(bad)
            tot['deviating_status'] += int((~same).sum())
            tot['deviating_value'] += int((same & (err > TOL)).sum())
            good = ok & ~dev_mask
            if good.any():
                tot['worst_agreeing'] = max(tot['worst_agreeing'], float(err[good].max()))
            if not len(bad):
                continue
            # the reference's own answer under 1e-15 changes of the ray: its whole packet (FULL), so
            # that an iteration that lands elsewhere is seen also where the ray fails later anyway
            p_b, d_b = pt0[:, bad].copy(), d[:, bad].copy()
            w_b = wi[bad] if isinstance(wi, np.ndarray) else wi
            o_full = oracle.make_opts(flags=flags, **dict(kw, out_mode=abi.OUT_FULL))
            with np.errstate(all='ignore'):
                base = oracle.trace_rays(tbl, p_b, d_b, w_b, o_full)
            nseg = base.seg.shape[0]
            groups = tuple(slice(10 * k + a0, 10 * k + a1) for k in range(nseg)
                           for a0, a1 in ((0, 3), (3, 6), (6, 7), (7, 10)))
            base_flat = base.seg.reshape(nseg * 10, -1)
            ill = np.zeros(len(bad), dtype=bool)
            for _ in range(6):
                ep = prng.uniform(-1e-15, 1e-15, size=p_b.shape)
                ed = prng.uniform(-1e-15, 1e-15, size=d_b.shape)
                p2 = p_b * (1.0 + ep) + ep * np.abs(p_b).max(axis=0, keepdims=True)
                d2 = d_b * (1.0 + ed) + ed      # (|d| = 1: an absolute 1e-15 for the zero components)
                with np.errstate(all='ignore'):
                    pert = oracle.trace_rays(tbl, p2, d2, w_b, o_full)
                moved = scaled(base_flat, pert.seg.reshape(nseg * 10, -1), groups).max(axis=0)
                moved = np.maximum(moved, scaled(base.op, pert.op))
                ill |= (pert.status != base.status) | (pert.fail_surf != base.fail_surf) | (moved > TOL / 16)
                # ... and what the output mode makes of it (HITS: inc + (foc / ad.z) ad)
                if mode != abi.OUT_FULL:
                    with np.errstate(all='ignore'):
                        pm = oracle.trace_rays(tbl, p2, d2, w_b, o_ref)
                    mm = scaled(orc.seg[:, bad], pm.seg, GROUPS[int(mode)]).max(axis=0)
                    ill |= (orc.status[bad] == abi.OK) & (pm.status == abi.OK) & (mm > TOL / 16)
            tot['ill_conditioned'] += int(ill.sum())
            for j in np.flatnonzero(~ill):
                # the reference's own Spencer-Murty step counts along this ray
                wj = int(w_b[j]) if isinstance(w_b, np.ndarray) else w_b
                logbuf[:] = 0
                olib.rox_oracle_newton_log(logbuf.ctypes.data_as(C.POINTER(C.c_byte)), len(logbuf))
                with np.errstate(all='ignore'):
                    one = oracle.trace_rays(tbl, p_b[:, j:j + 1].copy(), d_b[:, j:j + 1].copy(), wj, o_full)
                n_log = int(olib.rox_oracle_newton_log(None, 0))
                steps = logbuf[:min(n_log, len(logbuf))]
                k_fail = int(one.fail_surf[0])
                miss_inside = (int(one.status[0]) == abi.MISSED_SURFACE and k_fail >= 0 and
                               tbl.rows[k_fail].profile >= abi.EVENPOLY and len(steps) and int(steps[-1]) >= 1)
                if (len(steps) and int(steps.max()) > 8) or miss_inside:
                    tot['wandering_iteration'] += 1
                    continue
                tot['unexplained_deviations'] += 1
                if len(listed) < 16:
                    r = int(bad[j])
                    listed.append({'seed': seed, 'mode': int(mode), 'ray': r,
                                   'ref': [int(orc.status[r]), int(orc.fail_surf[r])],
                                   'fast': [int(dev.status[r]), int(dev.fail_surf[r])], 'err': float(err[r]),
                                   'newton_steps': [int(x) for x in steps[:8]]})
        eng.close()
        tot['systems'] += 1
    tot.update(first_seed=first, modes=len(modes), full_packets=with_full,
               full_tolerance_kernels_forced=os.environ.get('ROX_FAST_FP64_FULL') == '1', seconds=round(time.time() - t0, 1), tolerance=TOL,
               unexplained_deviations_listed=listed)
    print(json.dumps(tot))


if __name__ == '__main__':
    main()
