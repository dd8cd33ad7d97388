#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on BASELINE.json's configuration.

    python bench.py [--gpus N] [--steps K] [--warmup W]

Workload (configs[1]): the 13-interface double Gauss (K = 12 intersections per
ray), 1 field, 1 wavelength, 1024 x 1024 pupil grid = 1,048,576 rays per step,
FULL ray packets (the RayPkg return shape kept intact: 13 segments x 10 f64 per
ray, written SoA to HBM), rays generated on the device -- the trace_grid-shaped
entry of the C ABI.  One "step" = one such grid.  Inputs are resident in HBM
(the surface table; there are no per-ray inputs).

metric      ray-surface intersections per second, counting the intersections
            actually performed (a ray blocked at surface s contributes s, not K)
roofline    HBM-bound kernel: algorithmic bytes = what one launch must write
            (80 B per appended segment + 8 B op + 3 B status, +16 B pupil)
            divided by the trace kernel's mean duration from HIP events on the
            launch stream
cpu_baseline the plain-C oracle (oracle/rox_oracle.c, "port"), 1 thread, on a
            bounded sample of the same grid, timed on this host

N > 1: launched by torch.distributed.run, one rank per GPU; each rank traces
its own (field, wavelength) grid of the same size (weak scaling, no data-path
collective in the timed region); max-over-ranks time.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--num', type=int, default=1024, help='pupil grid is num x num')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--force-dist', action='store_true',
                    help='take the multi-rank code path (process group, collectives, sharded '
                         'spot) even with one rank: a 1-GPU rehearsal of the N>1 run')
    ap.add_argument('--cpu-sample-rows', type=int, default=0,
                    help='pupil rows traced by the CPU baseline (0 = auto, ~10 s)')
    return ap.parse_args()


def main():
    args = parse()
    import torch
    import torch.distributed as dist
    import rayoptics_amd  # noqa: F401
    from rayoptics_amd import abi, workloads
    from rayoptics_amd.engine import TraceEngine, make_opts, make_grid, DeviceResult

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f'--gpus {args.gpus} needs torch.distributed.run with '
                         f'{args.gpus} ranks (WORLD_SIZE={world})')
    torch.cuda.set_device(local_rank)
    multi = world > 1 or args.force_dist
    if multi:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29533')
        dist.init_process_group('nccl', rank=rank, world_size=world,
                                device_id=torch.device('cuda', local_rank))

    wl = workloads.load('dblgauss_c2')
    N = wl.n_ifcs
    K = N - 1
    eng = TraceEngine(wl.table)
    num = args.num
    R = num * num
    # weak scaling: rank r owns (field, wavelength) block r
    nf, nw = len(wl.fields), len(wl.table.wvls)
    fi = rank % nf
    wi = (wl.ref_wvl_idx + rank // nf) % nw
    fld = wl.fields[fi]
    grid = make_grid((-1., -1.), (1., 1.), num)
    flags = abi.INTERSECT_OBJ | abi.CHECK_APERTURES | abi.APPLY_VIGNETTING
    opts = make_opts(flags=flags, out_mode=abi.OUT_FULL, first_surf=1, last_surf=N - 2)
    out = DeviceResult(torch, eng.device, eng.num_segments(flags), R, abi.OUT_FULL,
                       want_pupil=True, nan_fill=False)

    def step():
        eng.trace_pupil_grid(fld, grid, wi, opts, out=out)

    def fence():
        if multi:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    if multi:
        t = torch.tensor([dt], dtype=torch.float64, device=eng.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()

    # work actually done by this rank's grid
    status = out.status.cpu().numpy()
    fail = out.fail_surf.cpu().numpy().astype(np.int64)
    ok = status == abi.OK
    inters = int(ok.sum()) * K + int(fail[~ok].sum())
    nseg = np.where(ok, N, np.where(status == abi.MISSED_SURFACE, fail, fail + 1))
    alg_bytes = int(nseg.sum()) * 80 + R * (8 + 1 + 2 + 16)
    tot = torch.tensor([inters, R], dtype=torch.float64, device=eng.device)
    if multi:
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
    inters_all, rays_all = tot[0].item(), tot[1].item()

    # dominant kernel: mean launch duration from HIP events on the launch stream
    kern_ms = eng.time_pupil_grid(fld, grid, wi, opts, out, max(args.steps, 10))
    # spot-diagram wall-clock (HITS mode: Python call -> host (R_ok, 2) array)
    o_hits = make_opts(flags=flags, out_mode=abi.OUT_HITS, first_surf=1, last_surf=N - 2,
                       foc=wl.foc, image_pt=wl.image_pts[fi])
    hits = DeviceResult(torch, eng.device, 0, R, abi.OUT_HITS, want_pupil=False, nan_fill=False)
    xy_pinned = torch.empty((R, 2), dtype=torch.float64).pin_memory()
    spot_ms = []
    for _ in range(5):
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        eng.trace_pupil_grid(fld, grid, wi, o_hits, want_pupil=False, out=hits)
        idx = torch.nonzero(hits.status == 0).squeeze(1)
        xy_dev = torch.stack((hits.seg[0].index_select(0, idx),
                              hits.seg[1].index_select(0, idx)), dim=1)
        xy_pinned[:xy_dev.shape[0]].copy_(xy_dev, non_blocking=True)
        torch.cuda.synchronize()
        xy = xy_pinned[:xy_dev.shape[0]].numpy()
        spot_ms.append((time.perf_counter() - t1) * 1e3)
    hits_kern_ms = eng.time_pupil_grid(fld, grid, wi, o_hits, hits, 10)

    # N > 1: the path's one exchange step -- every (field, wavelength) spot
    # diagram sharded by pupil-row blocks, hits gathered to rank 0 over RCCL
    sharded = None
    if multi:
        try:
            from rayoptics_amd.dist import trace_spot_sharded
            trace_spot_sharded(eng, wl.fields, wl.image_pts, nw, num, wl.foc)      # warm-up
            fence()
            t1 = time.perf_counter()
            res = trace_spot_sharded(eng, wl.fields, wl.image_pts, nw, num, wl.foc)
            fence()
            sharded = {'wallclock_ms': (time.perf_counter() - t1) * 1e3,
                       'grids': nf * nw, 'rays': nf * nw * R,
                       'what': 'all (field,wvl) spot diagrams, pupil-row blocks over ranks, '
                               'HITS trace + gather to rank 0 + host reassembly'}
            if rank == 0:
                sharded['grids_returned'] = len(res)
        except Exception as e:      # never lose the main line to the extra leg
            sharded = {'error': repr(e)}

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(wl, fld, wi, opts, num, args.cpu_sample_rows)

    if rank == 0:
        traffic = None
        tpath = os.path.join(ROOT, 'profiles', 'traffic.json')
        if os.path.exists(tpath):
            with open(tpath) as f:
                tj = json.load(f)
            if tj.get('num') == num and tj.get('workload') == wl.name:
                traffic = tj.get('hbm_bytes_per_launch')
        achieved = alg_bytes / (kern_ms * 1e-3) / 1e9
        line = {
            'metric': 'ray-surface intersections/sec',
            'value': inters_all / dt * args.steps,
            'unit': 'ray-surface intersections/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': dt / args.steps * 1e3,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f64', 'data': 'synthetic',
            'config': {'workload': 'double-Gauss 13 interfaces (K=12), 1 field, 1 wvl, '
                                   f'{num}x{num} pupil grid per GPU, FULL ray packets, '
                                   'device-generated rays (BASELINE.json configs[1])',
                       'rays_per_step': int(rays_all), 'interfaces': N,
                       'intersections_per_step': int(inters_all),
                       'nominal_R_times_K': int(rays_all) * K,
                       'out_mode': 'FULL', 'field_index': fi, 'wvl_nm': wl.table.wvls[wi],
                       'sharding': 'one (field,wvl) grid per rank' if world > 1 else 'single GPU'},
            'rays_per_s': rays_all / dt * args.steps,
            'roofline': {'bound': 'hbm', 'achieved': achieved, 'peak': 8000.0, 'unit': 'GB/s',
                         'frac': achieved / 8000.0, 'traffic': traffic,
                         'kernel': 'trace_kernel<FULL,PUPIL>', 'kernel_ms': kern_ms,
                         'algorithmic_bytes_per_launch': alg_bytes,
                         'frac_of_measured_copy_peak_6290': achieved / 6290.0},
            'spot_diagram': {'wallclock_ms': float(np.median(spot_ms)), 'rays': R,
                             'rays_through': int(xy.shape[0]), 'kernel_ms': hits_kern_ms,
                             'what': 'Python call -> host (R_ok,2) array, HITS mode'},
            'cpu_baseline': cpu,
            'sharded_spot': sharded,
        }
        print(json.dumps(line))
    if multi:
        dist.destroy_process_group()


def cpu_baseline(wl, fld, wi, opts, num, rows):
    """the oracle (plain-C port of the reference's algorithm), one thread, on a
    bounded sample of the same workload: the first `rows` pupil rows of the
    num x num grid (explicit pupil coordinates, same accumulate-by-step)."""
    from oracle import oracle
    from rayoptics_amd import abi
    oracle.build()
    N = wl.n_ifcs
    xs = np.empty(num)
    ys = np.empty(num)
    step = 2.0 / (num - 1)
    v = -1.0
    for k in range(num):
        xs[k] = v
        ys[k] = v
        v += step
    if rows <= 0:
        rows = num
    i0 = (num - rows) // 2
    px = np.repeat(xs[i0:i0 + rows], num)
    py = np.tile(ys, rows)
    # bounded sample: repeat the block of rows until ~10 s of CPU work is done
    res = oracle.HostResult(N, rows * num, opts.out_mode, want_pupil=True)   # untimed
    passes, dt = 0, 0.0
    while dt < 10.0 and passes < 64:
        t0 = time.perf_counter()
        oracle.trace_pupil_list(wl.table, fld, px, py, wi, opts, res=res)
        dt += time.perf_counter() - t0
        passes += 1
    ok = res.status == abi.OK
    inters = int(ok.sum()) * (N - 1) + int(res.fail_surf[~ok].astype(np.int64).sum())
    inters *= passes
    allc = cpu_all_cores(wl, fld, wi, opts, num, xs, ys)
    return {'value': inters / dt, 'unit': 'ray-surface intersections/s', 'cores': 1,
            'kind': 'port',
            'sample': f'{passes} passes over {rows} pupil rows x {num} = {rows * num} rays of the '
                      f'same grid, FULL packets, oracle/rox_oracle.c -O2 single thread, {dt:.1f} s',
            'rays_per_s': passes * rows * num / dt,
            'host_cpu_count': os.cpu_count(), 'all_cores': allc}


def cpu_all_cores(wl, fld, wi, opts, num, xs, ys):
    """context only: the same oracle fanned over threads (ctypes releases the
    GIL), one pass over the whole grid split into row blocks"""
    import threading
    from oracle import oracle
    from rayoptics_amd import abi
    nthr = max(1, min(os.cpu_count() or 1, 64, num))
    N = wl.n_ifcs
    bounds = [(num * k) // nthr for k in range(nthr + 1)]
    jobs = []
    for k in range(nthr):
        rows = bounds[k + 1] - bounds[k]
        if rows == 0:
            continue
        px = np.repeat(xs[bounds[k]:bounds[k + 1]], num)
        py = np.tile(ys, rows)
        jobs.append((px, py, oracle.HostResult(N, rows * num, opts.out_mode, want_pupil=True)))
    oracle.lib()
    thr = [threading.Thread(target=oracle.trace_pupil_list,
                            args=(wl.table, fld, px, py, wi, opts), kwargs={'res': res})
           for px, py, res in jobs]
    t0 = time.perf_counter()
    for t in thr:
        t.start()
    for t in thr:
        t.join()
    dt = time.perf_counter() - t0
    inters = 0
    for _px, _py, res in jobs:
        ok = res.status == abi.OK
        inters += int(ok.sum()) * (N - 1) + int(res.fail_surf[~ok].astype(np.int64).sum())
    return {'value': inters / dt, 'unit': 'ray-surface intersections/s', 'threads': len(jobs),
            'sample': f'one pass over the {num}x{num} grid, {dt:.2f} s'}


if __name__ == '__main__':
    main()
