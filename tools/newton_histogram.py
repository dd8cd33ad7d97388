#!/usr/bin/env python3
"""How many Spencer-Murty steps (rayoptics/elem/profiles.py:155-186) the asphere
intersections of the asphere workloads take -- the source of lane divergence in
the polynomial-profile kernel instances.  Counted by the CPU oracle (the device
executes the same iteration, bit for bit).

    python tools/newton_histogram.py            > profiles/r02_newton_histogram.json   (per-hit histogram)
    python tools/newton_histogram.py --waves    > profiles/r04_newton_wave_steps.json

--waves: a wave executes a step when ANY of its 64 lanes needs it (the four straight-line
steps are skipped wave-uniformly, the residual loop runs to the slowest lane), so what the
kernel pays at an asphere is the MAXIMUM step count over the wave's lanes.  For every asphere
interface of every (field) grid the tool sums that maximum over the waves of two lane
mappings of a num x num pupil grid (ray r = i * num + j, i = x index):
  rows     a wave = 64 consecutive rays of one pupil row (the shipped mapping)
  patch8   a wave = an 8 x 8 pupil patch (a narrower range of radii on every asphere)
and, as the floor, the lane-mean (what a divergence-free machine would pay)."""
import argparse
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def histogram():
    import rayoptics_amd  # noqa: F401
    from rayoptics_amd import abi, workloads
    from oracle import oracle
    lib = oracle.lib()
    lib.rox_oracle_newton_histogram.argtypes = [C.POINTER(C.c_longlong), C.c_int]
    out = {}
    for name in ('cell_phone', 'nikkor_c3'):
        wl = workloads.load(name)
        N = wl.n_ifcs
        h = (C.c_longlong * 16)()
        lib.rox_oracle_newton_histogram(h, 1)
        opts = oracle.make_opts(flags=abi.INTERSECT_OBJ | abi.CHECK_APERTURES | abi.APPLY_VIGNETTING,
                                out_mode=abi.OUT_HITS, first_surf=1, last_surf=N - 2)
        rec = {}
        for fi in range(len(wl.fields)):
            oracle.trace_pupil_grid(wl.table, wl.fields[fi], oracle.make_grid((-1., -1.), (1., 1.), 256),
                                    wl.ref_wvl_idx, opts)
        lib.rox_oracle_newton_histogram(h, 1)
        tot = sum(h)
        rec['asphere_hits'] = tot
        rec['steps_after_the_first_evaluation'] = {str(i): h[i] for i in range(16) if h[i]}
        rec['fraction'] = {str(i): round(h[i] / tot, 4) for i in range(16) if h[i]}
        rec['mean_steps'] = sum(i * h[i] for i in range(16)) / tot
        out[name] = rec
    print(json.dumps(out, indent=1))


def steps_per_ray(lib, wl, fi, wi, num, n_asph):
    """[n_asph][num*num] int8: steps of ray r at its k-th asphere interface (0 where the ray
    did not get there)"""
    from rayoptics_amd import abi
    from oracle import oracle
    N = wl.n_ifcs
    R = num * num
    cap = R * (n_asph + 1) + 16
    buf = np.zeros(cap, dtype=np.int8)
    lib.rox_oracle_newton_log(buf.ctypes.data_as(C.POINTER(C.c_byte)), cap)
    flags = abi.CHECK_APERTURES | abi.APPLY_VIGNETTING
    if wl.fields[fi].kind != abi.FLD_EPD_WIDE and wl.fields[fi].z_dir0 != 0.0:
        flags |= abi.INTERSECT_OBJ
    opts = oracle.make_opts(flags=flags, out_mode=abi.OUT_HITS, first_surf=1, last_surf=N - 2)
    oracle.trace_pupil_grid(wl.table, wl.fields[fi], oracle.make_grid((-1., -1.), (1., 1.), num), wi, opts)
    n = lib.rox_oracle_newton_log(None, 0)
    assert n <= cap
    log = buf[:n]
    ends = np.flatnonzero(log == -1)
    assert len(ends) == R, (len(ends), R)
    starts = np.concatenate([[0], ends[:-1] + 1])
    lens = ends - starts
    out = np.zeros((n_asph, R), dtype=np.int8)
    for k in range(n_asph):
        has = lens > k
        out[k, has] = log[starts[has] + k]
    return out


def wave_steps(st, num, mapping):
    """sum over waves of the maximum over the wave's lanes, st = [R] steps of one asphere"""
    g = st.reshape(num, num)
    if mapping == 'rows':
        return int(g.reshape(num, num // 64, 64).max(axis=2).sum())
    assert mapping == 'patch8'
    return int(g.reshape(num // 8, 8, num // 8, 8).max(axis=(1, 3)).sum())


def waves(num):
    import rayoptics_amd  # noqa: F401
    from rayoptics_amd import abi, workloads
    from oracle import oracle
    lib = oracle.lib()
    lib.rox_oracle_newton_log.restype = C.c_longlong
    lib.rox_oracle_newton_log.argtypes = [C.POINTER(C.c_byte), C.c_longlong]
    out = {'grid': f'{num} x {num}', 'waves_per_grid': num * num // 64,
           'what': 'wave-steps = sum over the waves of a grid of max-over-lanes Spencer-Murty steps, per '
                   'asphere interface, summed over them; ratio to lane_mean_x64 = how much of the Newton '
                   'work a wave executes is divergence'}
    for name in ('cell_phone', 'nikkor_c3', 'zmx_evenasph_c3'):
        wl = workloads.load(name)
        asph = [i for i, r in enumerate(wl.table.rows)
                if r.profile not in (abi.SPHERICAL, abi.CONIC, abi.THINLENS)]
        rec = {'asphere_interfaces': asph, 'fields': []}
        tot = {'rows': 0, 'patch8': 0, 'lane_mean_x64': 0.0}
        for fi in range(len(wl.fields)):
            st = steps_per_ray(lib, wl, fi, wl.ref_wvl_idx, num, len(asph))
            f = {'field': fi, 'rows': 0, 'patch8': 0, 'lane_mean_x64': 0.0, 'per_asphere': []}
            for k in range(len(asph)):
                a, b = wave_steps(st[k], num, 'rows'), wave_steps(st[k], num, 'patch8')
                m = float(st[k].astype(np.int64).sum()) / 64.0
                f['per_asphere'].append({'ifc': asph[k], 'rows': a, 'patch8': b, 'lane_mean_x64': round(m, 1),
                                         'max_steps': int(st[k].max())})
                f['rows'] += a
                f['patch8'] += b
                f['lane_mean_x64'] += m
            f['patch8_over_rows'] = round(f['patch8'] / f['rows'], 4)
            f['lane_mean_x64'] = round(f['lane_mean_x64'], 1)
            for k in tot:
                tot[k] += f[k]
            rec['fields'].append(f)
        rec['all_fields'] = {'rows': tot['rows'], 'patch8': tot['patch8'],
                             'lane_mean_x64': round(tot['lane_mean_x64'], 1),
                             'patch8_over_rows': round(tot['patch8'] / tot['rows'], 4),
                             'rows_over_lane_mean': round(tot['rows'] / tot['lane_mean_x64'], 4),
                             'patch8_over_lane_mean': round(tot['patch8'] / tot['lane_mean_x64'], 4)}
        out[name] = rec
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--waves', action='store_true')
    ap.add_argument('--num', type=int, default=512)
    a = ap.parse_args()
    waves(a.num) if a.waves else histogram()
