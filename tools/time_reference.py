#!/usr/bin/env python3
"""Times the REFERENCE ITSELF (mjhoptics/ray-optics, pure Python/NumPy) on the
hot path, on this host, and records the result in profiles/reference_cpu.json.

    python tools/time_reference.py [--num 256] [--procs N]

Build container only: the reference tree (/root/reference) does not exist on
the GPU box, so bench.py folds this file into ``cpu_baseline.reference_python``
as a number measured on *this* host (stated), next to the same-host C port.
The measurement mirrors the reference's own benchmark of this path,
rayoptics/raytr/tests/time_trace.py:21-133 (repeated ``rt.trace`` calls on the
model files), on BASELINE.json configs[1]:

  driver   ``trace.trace_grid`` (rayoptics/raytr/trace.py:563-605) over a
           num x num pupil grid of the double Gauss, on-axis field, central
           wavelength (trace_grid sets check_apertures=True), SpotDiagramFigure's ``spot`` filter:
           what a user of the reference runs to get a spot diagram
  raw      the same rays (pt0, dir0 precomputed with ray_start_from_osp) through
           ``rt.trace`` (rayoptics/raytr/raytrace.py:51-80) alone, as
           time_trace.py does
  fanned   ``raw`` over ``--procs`` processes (the reference has no parallelism
           of its own; rays are independent)
  port     oracle/rox_oracle.c (the C restatement bench.py times on the GPU box
           as ``cpu_baseline``), one thread, on the SAME num x num grid on THIS
           host -> ``port_same_host`` with the port : reference ratio, which is
           the bridge between the two hosts (bench.py divides the GPU host's
           port rate by it)
  config1  BASELINE.json configs[0]: singlet, 1 field, 1 wvl, 64 x 64 grid through
           the reference's ``trace.trace_grid`` (plumbing case, CPU only)
"""
import argparse
import json
import multiprocessing as mp
import os
import platform
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))

import numpy as np  # noqa: E402


def cpu_model():
    try:
        with open('/proc/cpuinfo') as f:
            for line in f:
                if line.startswith('model name'):
                    return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return platform.processor()


def blas_info():
    try:
        cfg = np.show_config(mode='dicts')
        b = cfg.get('Build Dependencies', {}).get('blas', {})
        return f"{b.get('name')} {b.get('version')}"
    except Exception:
        return 'unknown'


def build():
    import refmodels as rm
    opm = rm.dblgauss()
    return opm


def rays_of_grid(opm, num):
    """(pt0, dir0) of the num x num grid, the way trace_grid/trace_base make them"""
    osp = opm['optical_spec']
    sm = opm['seq_model']
    fld = osp['fov'].fields[0]
    start = np.array([-1., -1.])
    stop = np.array([1., 1.])
    step = (stop - start) / (num - 1)
    rays = []
    s = start.copy()
    for _i in range(num):
        for _j in range(num):
            pupil = fld.apply_vignetting(np.array(s))
            pt0, dir0 = osp.ray_start_from_osp(pupil, fld, 'rel pupil')
            if dir0[2] * sm.z_dir[0] < 0:
                dir0 = -dir0
            rays.append((pt0, dir0))
            s[1] += step[1]
        s[0] += step[0]
        s[1] = start[1]
    return rays


def raw_loop(args):
    """worker: rt.trace over a slice of the rays; returns (seconds, rays, intersections)"""
    lo, hi, num = args
    import rayoptics.raytr.raytrace as rt
    from rayoptics.raytr.traceerror import TraceError
    opm = build()
    sm = opm['seq_model']
    wvl = sm.central_wavelength()
    rays = rays_of_grid(opm, num)[lo:hi]
    K = len(sm.ifcs) - 1
    inters = 0
    t0 = time.perf_counter()
    for pt0, dir0 in rays:
        try:
            rt.trace(sm, pt0, dir0, wvl, check_apertures=True)
            inters += K
        except TraceError as e:
            inters += e.surf
    return time.perf_counter() - t0, len(rays), inters


def port_same_host(num, ref_inters_per_s, ref_rays_per_s, ref_inters):
    """oracle/rox_oracle.c, single thread, FULL packets, on the same grid of the
    same model (the stored dblgauss_c2 table is the table of rm.dblgauss())."""
    import rayoptics_amd  # noqa: F401
    from rayoptics_amd import abi, workloads
    from rayoptics_amd.engine import make_opts
    from oracle import oracle
    oracle.build()
    wl = workloads.load('dblgauss_c2')
    N = wl.n_ifcs
    fld = wl.fields[0]
    opts = make_opts(flags=abi.INTERSECT_OBJ | abi.CHECK_APERTURES | abi.APPLY_VIGNETTING,
                     out_mode=abi.OUT_FULL, first_surf=1, last_surf=N - 2)
    xs = np.empty(num)
    v, step = -1.0, 2.0 / (num - 1)
    for k in range(num):
        xs[k] = v
        v += step
    px = np.repeat(xs, num)
    py = np.tile(xs, num)
    res = oracle.HostResult(N, num * num, opts.out_mode, want_pupil=True)
    oracle.trace_pupil_list(wl.table, fld, px, py, wl.ref_wvl_idx, opts, res=res)   # warm
    passes, dt = 0, 0.0
    while dt < 5.0:
        t0 = time.perf_counter()
        oracle.trace_pupil_list(wl.table, fld, px, py, wl.ref_wvl_idx, opts, res=res)
        dt += time.perf_counter() - t0
        passes += 1
    ok = res.status == abi.OK
    inters = int(ok.sum()) * (N - 1) + int(res.fail_surf[~ok].astype(np.int64).sum())
    rate = inters * passes / dt
    return {'what': 'oracle/rox_oracle.c -O2, 1 thread, FULL packets, the same grid on the same host',
            'passes': passes, 'seconds': dt, 'rays': num * num, 'rays_through': int(ok.sum()),
            'intersections_per_pass': inters, 'reference_intersections': ref_inters,
            'rays_per_s': num * num * passes / dt, 'intersections_per_s': rate,
            'port_over_reference': rate / ref_inters_per_s,
            'port_over_reference_by_rays': (num * num * passes / dt) / ref_rays_per_s,
            'ratio_of': 'port intersections/s : reference raw rt.trace intersections/s, one core each'}


def config1_singlet(num=64):
    """BASELINE.json configs[0] on the reference itself"""
    import refmodels as rm
    import rayoptics.raytr.trace as trace
    opm = rm.singlet()
    sm, osp = opm['seq_model'], opm['optical_spec']
    fld = osp['fov'].fields[0]
    wvl = sm.central_wavelength()
    foc = osp['focus'].focus_shift
    rs, cr = trace.setup_pupil_coords(opm, fld, wvl, foc)
    fld.chief_ray, fld.ref_sphere = cr, rs
    image_pt = rs[0]

    def spot(p, ray_pkg):                      # axisarrayfigure.py:229-238
        if ray_pkg is not None:
            seg = ray_pkg[0][-1]
            dist = foc / seg[1][2]
            t_abr = (seg[0] + dist * seg[1]) - image_pt
            return np.array([t_abr[0], t_abr[1]])
        return None

    best = None
    for _ in range(3):
        t0 = time.perf_counter()
        grid = trace.trace_grid(opm, [np.array([-1., -1.]), np.array([1., 1.]), num], fld, wvl, foc,
                                img_filter=spot, form='list', append_if_none=False)
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    K = len(sm.ifcs) - 1
    return {'workload': f'BASELINE.json configs[0]: singlet, {len(sm.ifcs)} interfaces, 1 field, 1 wvl, '
                        f'{num}x{num} grid, reference trace.trace_grid',
            'seconds_best_of_3': best, 'rays': num * num, 'rays_through': len(grid),
            'rays_per_s': num * num / best, 'intersections_per_s_nominal': num * num * K / best}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--num', type=int, default=256, help='pupil grid is num x num (>= 65536 rays)')
    ap.add_argument('--procs', type=int, default=os.cpu_count() or 1)
    ap.add_argument('--out', default=os.path.join(ROOT, 'profiles', 'reference_cpu.json'))
    args = ap.parse_args()
    num = args.num
    R = num * num
    opm = build()
    import rayoptics
    import rayoptics.raytr.trace as trace
    sm, osp = opm['seq_model'], opm['optical_spec']
    fld = osp['fov'].fields[0]
    wvl = sm.central_wavelength()
    foc = osp['focus'].focus_shift
    rs, cr = trace.setup_pupil_coords(opm, fld, wvl, foc)
    fld.chief_ray, fld.ref_sphere = cr, rs
    image_pt = rs[0]

    def spot(p, ray_pkg):                      # axisarrayfigure.py:229-238
        if ray_pkg is not None:
            seg = ray_pkg[0][-1]
            dist = foc / seg[1][2]
            t_abr = (seg[0] + dist * seg[1]) - image_pt
            return np.array([t_abr[0], t_abr[1]])
        return None

    # driver: what sequential.py:1058-1085 calls per wavelength
    t0 = time.perf_counter()
    grid = trace.trace_grid(opm, [np.array([-1., -1.]), np.array([1., 1.]), num], fld, wvl, foc,
                            img_filter=spot, form='list', append_if_none=False)
    t_driver = time.perf_counter() - t0
    n_through = len(grid)

    t_raw, n_raw, inters = raw_loop((0, R, num))
    with mp.get_context('fork').Pool(args.procs) as pool:
        bounds = [(R * k) // args.procs for k in range(args.procs + 1)]
        t0 = time.perf_counter()
        parts = pool.map(raw_loop, [(bounds[k], bounds[k + 1], num) for k in range(args.procs)])
        t_fan = time.perf_counter() - t0
    # (each worker rebuilds the model and the rays outside its timed loop; the
    # fanned figure uses the slowest worker's loop time, not the pool wall-clock)
    t_fan_loop = max(p[0] for p in parts)

    N = len(sm.ifcs)
    rec = {
        'what': 'reference (mjhoptics/ray-optics) timed on the build container, not on the GPU box',
        'workload': f'BASELINE.json configs[1]: double Gauss, {N} interfaces (K={N - 1}), on-axis field, '
                    f'{wvl} nm, {num}x{num} pupil grid = {R} rays, check_apertures=True',
        'host': {'cpu': cpu_model(), 'nproc': os.cpu_count(), 'python': platform.python_version(),
                 'numpy': np.__version__, 'blas': blas_info(),
                 'rayoptics': getattr(rayoptics, '__version__', 'source tree /root/reference')},
        'driver_trace_grid': {'seconds': t_driver, 'rays': R, 'rays_through': n_through,
                              'rays_per_s': R / t_driver,
                              'intersections_per_s': inters / t_driver,
                              'extrapolated_1M_ray_spot_s': 1048576 / (R / t_driver)},
        'raw_rt_trace': {'seconds': t_raw, 'rays': n_raw, 'rays_per_s': n_raw / t_raw,
                         'intersections': inters, 'intersections_per_s': inters / t_raw, 'cores': 1},
        'raw_rt_trace_fanned': {'processes': args.procs, 'slowest_worker_loop_s': t_fan_loop,
                                'pool_wallclock_s': t_fan,
                                'rays_per_s': R / t_fan_loop,
                                'intersections_per_s': sum(p[2] for p in parts) / t_fan_loop},
    }
    rec['port_same_host'] = port_same_host(num, inters / t_raw, n_raw / t_raw, inters)
    rec['config1_singlet_64'] = config1_singlet(64)
    with open(args.out, 'w') as f:
        json.dump(rec, f, indent=1)
    print(json.dumps(rec, indent=1))


if __name__ == '__main__':
    main()
