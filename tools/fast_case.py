#!/usr/bin/env python3
"""One ray of a tools/fast_soak.py system under the microscope: the oracle's FULL packet beside the
tolerance-mode FULL packet, segment by segment, and the HITS / LAST values of both.

    python tools/fast_case.py <seed> <ray> [<seed> <ray> ...]"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))


def main():
    import rayoptics_amd  # noqa: F401
    from rayoptics_amd import abi
    from rayoptics_amd.engine import TraceEngine
    from oracle import oracle
    import test_gpu_fuzz as t
    args = [int(a) for a in sys.argv[1:]]
    for seed, ray in zip(args[0::2], args[1::2]):
        rng = np.random.default_rng(1000 + seed)
        tbl = t.random_table(rng)
        N = tbl.n_ifcs
        R = 3000 + int(rng.integers(0, 200))
        pt0, d = t.random_rays(rng, tbl, R)
        W = len(tbl.wvls)
        wi = rng.integers(0, W, R).astype(np.int32) if seed % 2 else int(rng.integers(0, W))
        eng = TraceEngine(tbl)
        flags = (abi.INTERSECT_OBJ if seed % 5 else 0) | (abi.CHECK_APERTURES if seed % 3 else 0)
        kw = dict(first_surf=int(seed % 2), last_surf=(N - 2) if seed % 7 else -1,
                  foc=0.01 * (seed % 50), image_pt=(0.1, -0.2))
        sl = slice(ray, ray + 1)
        w1 = wi[sl] if isinstance(wi, np.ndarray) else wi
        rec = {'seed': seed, 'ray': ray, 'interfaces': N, 'pt0': pt0[:, ray].tolist(), 'dir0': d[:, ray].tolist(),
               'profiles': [int(r.profile) for r in tbl.rows], 'modes': [int(r.mode) for r in tbl.rows],
               'thi': [float(r.t[2]) for r in tbl.rows], 'foc': kw['foc']}
        for mode, label in ((abi.OUT_FULL, 'full'), (abi.OUT_HITS, 'hits'), (abi.OUT_LAST, 'last')):
            o_ref = oracle.make_opts(flags=flags, out_mode=mode, **kw)
            o_fast = oracle.make_opts(flags=flags | abi.FAST_FP64, out_mode=mode, **kw)
            with np.errstate(all='ignore'):
                orc = oracle.trace_rays(tbl, pt0[:, sl], d[:, sl], w1, o_ref)
            dev = eng.trace_rays(pt0[:, sl], d[:, sl], w1, o_fast, nan_fill=True).to_host()
            ex = eng.trace_rays(pt0[:, sl], d[:, sl], w1, o_ref, nan_fill=True).to_host()
            rec[label] = {'status': [int(orc.status[0]), int(dev.status[0])],
                          'fail_surf': [int(orc.fail_surf[0]), int(dev.fail_surf[0])],
                          'exact_device_equals_oracle': bool(np.array_equal(orc.seg, ex.seg, equal_nan=True))}
            if mode == abi.OUT_FULL:
                with np.errstate(all='ignore'):
                    diff = np.abs(orc.seg[:, :, 0] - dev.seg[:, :, 0])
                    mag = np.maximum(1.0, np.abs(orc.seg[:, :, 0]))
                rec[label]['per_segment_max_scaled_dev'] = [float(np.nanmax(diff[k] / mag[k])) if np.isfinite(diff[k]).any() else None
                                                            for k in range(diff.shape[0])]
                rec[label]['oracle_p_d_dst'] = [[float(v) for v in orc.seg[k, :7, 0]] for k in range(diff.shape[0])]
                rec[label]['fast_p_d_dst'] = [[float(v) for v in dev.seg[k, :7, 0]] for k in range(diff.shape[0])]
            else:
                rec[label]['oracle'] = [float(v) for v in orc.seg[:, 0]]
                rec[label]['fast'] = [float(v) for v in dev.seg[:, 0]]
        print(json.dumps(rec), flush=True)
        eng.close()


if __name__ == '__main__':
    main()
