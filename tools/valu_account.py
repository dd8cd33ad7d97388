#!/usr/bin/env python3
"""Where the VALU instructions of the Newton instances go (no GPU needed): the wave-level work of
the PMC launches -- a 1024 x 1024 on-axis grid, 64 consecutive rays per wave -- counted by the
oracle, held against the measured SQ_INSTS_VALU of the same launches (profiles/*pmc_summary*).

    python tools/valu_account.py [--pmc r04] > profiles/r05_valu_account.json

A wave executes a surface while ANY of its lanes is alive there, and a Spencer-Murty
evaluation at an asphere while any lane still iterates: what the kernel pays is
  wave-surfaces  = sum over waves of the deepest surface one of its rays reaches
  wave-evals     = sum over waves and aspheres of 1 + max over lanes of the steps there.
`SQ_INSTS_VALU / intersections` (profiles/valu_per_intersection.json) divides by the rays' own
intersections instead: a wave with one live lane weighs 64 x there, and a model that vignettes
more looks more expensive per intersection without executing one instruction more per
wave-surface.  With the lean instance's cost per wave-surface (double Gauss) taken for the
spherical surfaces, the remainder over the wave-evals is the cost of ONE asphere evaluation per
wave -- comparable between models, and with the static count of one Spencer-Murty step."""
import argparse
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))


def wave_work(name, num=1024, fi=0):
    import rayoptics_amd  # noqa: F401
    from rayoptics_amd import abi, workloads
    from oracle import oracle
    import newton_histogram as nh
    lib = oracle.lib()
    lib.rox_oracle_newton_log.restype = C.c_longlong
    lib.rox_oracle_newton_log.argtypes = [C.POINTER(C.c_byte), C.c_longlong]
    wl = workloads.load(name)
    N = wl.n_ifcs
    asph = [i for i, r in enumerate(wl.table.rows)
            if r.profile not in (abi.SPHERICAL, abi.CONIC, abi.THINLENS)]
    flags = abi.CHECK_APERTURES | abi.APPLY_VIGNETTING
    if wl.fields[fi].kind != abi.FLD_EPD_WIDE and wl.fields[fi].z_dir0 != 0.0:
        flags |= abi.INTERSECT_OBJ
    opts = oracle.make_opts(flags=flags, out_mode=abi.OUT_HITS, first_surf=1, last_surf=N - 2,
                            foc=wl.foc, image_pt=wl.image_pts[fi])
    res = oracle.trace_pupil_grid(wl.table, wl.fields[fi], oracle.make_grid((-1., -1.), (1., 1.), num),
                                  wl.ref_wvl_idx, opts)
    ok = res.status == abi.OK
    # deepest surface a ray executes: all N - 1 when it gets through, else the one it fails at
    depth = np.where(ok, N - 1, res.fail_surf.astype(np.int64)).astype(np.int64)
    depth = np.maximum(depth, 0)
    W = num * num // 64
    wdepth = depth.reshape(W, 64).max(axis=1)
    rec = {'workload': name, 'interfaces': N, 'aspheres': asph, 'grid': f'{num} x {num} field {fi}',
           'rays': num * num, 'rays_through': int(ok.sum()), 'intersections': int(depth.sum()),
           'waves': W, 'wave_surfaces': int(wdepth.sum()),
           'lane_utilisation': float(depth.sum()) / (64.0 * float(wdepth.sum()))}
    if asph:
        st = nh.steps_per_ray(lib, wl, fi, wl.ref_wvl_idx, num, len(asph))
        evals, hits, per = 0, 0, []
        for k, s in enumerate(asph):
            alive = (wdepth >= s)                           # the wave executes this asphere
            mx = st[k].astype(np.int64).reshape(W, 64).max(axis=1)
            e = int((1 + mx[alive]).sum())
            evals += e
            lane_hits = int((depth >= s).sum())
            hits += lane_hits
            per.append({'ifc': s, 'ncoef': int(wl.table.rows[s].ncoef), 'waves_there': int(alive.sum()),
                        'wave_evals': e, 'evals_per_wave': e / max(int(alive.sum()), 1),
                        'lane_hits': lane_hits})
        rec.update(wave_evals=evals, asphere_lane_hits=hits, per_asphere=per)
    return rec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--pmc', default='r04', help='profiles/<tag>_pmc_summary_<workload>.json')
    args = ap.parse_args()
    out = {'what': __doc__.split('\n\n')[2].replace('\n', ' ')}
    recs = {n: wave_work(n) for n in ('dblgauss_c2', 'zmx_evenasph_c3', 'nikkor_c3', 'cell_phone')}
    for n, r in recs.items():
        p = os.path.join(ROOT, 'profiles', f'{args.pmc}_pmc_summary_{n}.json')
        if os.path.exists(p):
            h = json.load(open(p))['HITS']
            r['pmc'] = {'source': os.path.basename(p), 'SQ_INSTS_VALU': h['SQ_INSTS_VALU'],
                        'SQ_INSTS_SALU': h['SQ_INSTS_SALU'], 'SQ_INSTS_LDS': h['SQ_INSTS_LDS']}
            r['valu_per_intersection'] = h['SQ_INSTS_VALU'] / r['intersections']
            r['valu_per_wave_surface'] = h['SQ_INSTS_VALU'] / r['wave_surfaces']
    lean = recs['dblgauss_c2'].get('valu_per_wave_surface')
    if lean:
        for n in ('zmx_evenasph_c3', 'nikkor_c3', 'cell_phone'):
            r = recs[n]
            if 'pmc' not in r:
                continue
            rest = r['pmc']['SQ_INSTS_VALU'] - lean * r['wave_surfaces']
            r['asphere_model'] = {
                'spherical_cost_per_wave_surface_taken_from': 'dblgauss_c2 (lean instance): %.1f' % lean,
                'valu_left_for_the_asphere_evaluations': rest,
                'valu_per_wave_eval': rest / r['wave_evals'],
                'valu_per_asphere_wave_hit': rest / sum(a['waves_there'] for a in r['per_asphere']),
                'evals_per_asphere_wave_hit': r['wave_evals'] / sum(a['waves_there'] for a in r['per_asphere']),
                'note': 'per wave-hit = what one asphere adds on top of a sphere, per wave that reaches it'}
    out['workloads'] = recs
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
