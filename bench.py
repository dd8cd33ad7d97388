#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on BASELINE.json's configuration.

    python bench.py [--gpus N] [--steps K] [--warmup W]

Workload (configs[1]): the 13-interface double Gauss (K = 12 intersections per
ray), 1 field, 1 wavelength, 1024 x 1024 pupil grid = 1,048,576 rays per step,
FULL ray packets (the RayPkg return shape kept intact: 13 segments x 10 f64 per
ray, written SoA to HBM), rays generated on the device -- the trace_grid-shaped
entry of the C ABI.  One "step" = one such grid.  Inputs are resident in HBM
(the surface table; there are no per-ray inputs).

metric        ray-surface intersections per second, counting the intersections
              actually performed (a ray blocked at surface s contributes s, not K)
roofline      HBM-bound kernel: algorithmic bytes = what one launch must write
              (80 B per appended segment + 8 B op + 3 B status, +16 B pupil)
              divided by the trace kernel's mean duration from HIP events on the
              launch stream
roofline_hits the HITS kernel behind spot diagrams / OPD / refocus is fp64-VALU
              bound: TFLOP/s by SURVEY 8(d)'s 130 flop per intersection against the
              78.6 TFLOP/s fp64 vector peak, and the VALU issue fraction from the
              committed PMC summary
spot_diagram  BASELINE's second metric at the PRODUCT boundary: wall-clock of
              rayoptics_amd.trace.trace_grid_spot (the function SequentialModel.trace_grid
              is rebound to for SpotDiagramFigure) from the Python call to the host
              (R_ok, 2) array, on a table-backed model
cpu_baseline  the plain-C oracle (oracle/rox_oracle.c, "port"), 1 thread, on a
              bounded sample of the same grid, timed on this host; next to it the
              reference's own Python path as timed by tools/time_reference.py in the
              build container (the reference is not installed on the GPU box)
strong_scaling  every run, any N: BASELINE configs[4]'s shape -- 9 fields x 5
              wavelengths x 2048^2 pupil grids of the 44-interface lithography
              lens (188.7 M rays, HITS) cut into pupil-row blocks over the ranks
              (dist.partition), hits gathered to rank 0 over RCCL: kernel ms (max
              over ranks), gather ms and end-to-end ms, separately

N > 1: launched by torch.distributed.run, one rank per GPU.  The main line is
weak scaling (each rank traces its own (field, wavelength) grid of the same
size, no data-path collective in the timed region; max-over-ranks time); the
strong_scaling object carries the fixed-size problem with its one exchange step.
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
    ap.add_argument('--steps', type=int, default=400)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--num', type=int, default=1024, help='pupil grid is num x num')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--force-dist', action='store_true',
                    help='take the multi-rank code path (process group, collectives, sharded '
                         'spot) even with one rank: a 1-GPU rehearsal of the N>1 run')
    ap.add_argument('--cpu-sample-rows', type=int, default=0,
                    help='pupil rows traced by the CPU baseline (0 = auto, ~10 s)')
    ap.add_argument('--no-strong', action='store_true', help='skip the strong-scaling leg')
    ap.add_argument('--strong-num', type=int, default=2048,
                    help='pupil grid of the strong-scaling leg is num x num per (field, wvl)')
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
    # rehearsal switches (a 1-GPU box cannot run RCCL with two ranks): ROX_BENCH_SHARE_GPU=1
    # puts every rank on device 0 and ROX_BENCH_BACKEND=gloo carries the collectives --
    # the N > 1 control flow, partitioning and gathers run as they will over RCCL
    backend = os.environ.get('ROX_BENCH_BACKEND', 'nccl')
    if os.environ.get('ROX_BENCH_SHARE_GPU') == '1':
        local_rank = 0
    torch.cuda.set_device(local_rank)
    multi = world > 1 or args.force_dist
    saved_stdout = None
    if multi:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29533')
        # RCCL prints a version banner on stdout when its communicator is built; stdout
        # carries exactly one JSON line, so fd 1 points at stderr until that is over
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
        if backend == 'nccl':
            dist.init_process_group('nccl', rank=rank, world_size=world,
                                    device_id=torch.device('cuda', local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

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

    _tok = torch.zeros(1, device=eng.device) if multi else None

    def fence():
        # barrier + torch.cuda.synchronize(): the barrier is a one-element all-reduce on a
        # preallocated tensor (every rank must enter it; dist.barrier() is the same
        # collective behind more bookkeeping -- 2 ms per call on this stack)
        if multi:
            torch.cuda.synchronize()
            dist.all_reduce(_tok)
        torch.cuda.synchronize()

    # the GPU's clocks need tens of milliseconds of continuous work to settle (a cold
    # start reads 10-30 % slow, tools/sustained_probe.py): untimed launches first, at
    # least the W the caller asked for
    t_w = time.perf_counter()
    n_w = 0
    while n_w < args.warmup or (time.perf_counter() - t_w) < 0.15:
        step()
        n_w += 1
        if n_w % 64 == 0:
            torch.cuda.synchronize()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    t_f = time.perf_counter()
    fence()
    fence_ms = (time.perf_counter() - t_f) * 1e3      # cost of one (idle) fence, for the record
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
    kern_ms = eng.time_pupil_grid_sustained(fld, grid, wi, opts, out, max(min(args.steps, 50), 10))
    # HITS kernel (spot diagrams, OPD, refocus): mean launch duration
    o_hits = make_opts(flags=flags, out_mode=abi.OUT_HITS, first_surf=1, last_surf=N - 2,
                       foc=wl.foc, image_pt=wl.image_pts[fi])
    hits = DeviceResult(torch, eng.device, 0, R, abi.OUT_HITS, want_pupil=False, nan_fill=False)
    hits_kern_ms = eng.time_pupil_grid_sustained(fld, grid, wi, o_hits, hits,
                                                 max(min(args.steps, 50), 10))
    del hits

    # spot-diagram wall-clock at the product boundary: the function the reference's
    # SpotDiagramFigure reaches through SequentialModel.trace_grid, on a table-backed
    # model (the extraction of table / field constants from a live reference model is
    # what the stand-in skips; the reference is not installed on the GPU box)
    from rayoptics_amd import trace as rox_trace
    model = workloads.TableModel(wl)
    mfld = model.fields[fi]
    grid_rng = [np.array([-1., -1.]), np.array([1., 1.]), num]
    wvl_nm = wl.table.wvls[wi]
    for _ in range(60):         # settle clocks / pinned pool
        xy = rox_trace.trace_grid_spot(model, grid_rng, mfld, wvl_nm, wl.foc, wl.image_pts[fi])
    spot_ms = []
    for _ in range(41):
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        xy = rox_trace.trace_grid_spot(model, grid_rng, mfld, wvl_nm, wl.foc, wl.image_pts[fi])
        spot_ms.append((time.perf_counter() - t1) * 1e3)

    # the PSF of an OPD grid (analyses.calc_psf): the GEMM-shaped neighbour of the path,
    # on the fp64 matrix cores; device-resident, mean of back-to-back calls
    psf = None
    if rank == 0:
        try:
            psf = psf_leg(torch)
        except Exception as e:
            psf = {'error': repr(e)}

    # every run: the fixed-size problem with the path's one exchange step
    strong = None
    if not args.no_strong:
        try:
            strong = strong_scaling(args, torch, dist, multi, world, rank, fence)
        except Exception as e:      # never lose the main line to the extra leg
            strong = {'error': repr(e)}

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
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'warmup_steps_run': n_w,
            'ms_per_step': dt / args.steps * 1e3, 'fence_ms': fence_ms,
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
            'roofline_hits': roofline_hits(inters, R, hits_kern_ms),
            'spot_diagram': {'wallclock_ms': float(np.median(spot_ms)),
                             'wallclock_min_ms': float(np.min(spot_ms)), 'rays': R,
                             'rays_through': int(xy.shape[0]), 'kernel_hits_ms': hits_kern_ms,
                             'pcie_floor_ms': xy.shape[0] * 16 / 54.7e9 * 1e3,
                             'what': 'rayoptics_amd.trace.trace_grid_spot(model, grid_rng, fld, wvl, '
                                     'foc, image_pt): Python call -> host (R_ok, 2) float64 array '
                                     '(survivors packed in ray order by the trace launch, written '
                                     'straight into pinned host memory; 13 MB over PCIe at ~55 GB/s '
                                     'is the floor)'},
            'psf': psf,
            'cpu_baseline': cpu,
            'strong_scaling': strong,
        }
        # the one JSON line goes to the real stdout (fd 1 was pointed at stderr while the
        # communicator was being built and the collectives ran)
        sys.stdout.flush()
        if saved_stdout is not None:
            os.dup2(saved_stdout, 1)
            os.close(saved_stdout)
            saved_stdout = None
        print(json.dumps(line), flush=True)
    if multi:
        if saved_stdout is not None:            # other ranks: keep their fd 1 on stderr
            os.close(saved_stdout)
        sys.stdout.flush()
        os.dup2(2, 1)                           # teardown chatter does not belong on stdout
        dist.destroy_process_group()


def psf_leg(torch):
    """rox_calc_psf at two sizes: ms per call and fp64 TFLOP/s by 8 M n (n + M) flop
    (two complex GEMMs) against the 78.6 TFLOP/s fp64 peak (matrix = vector rate)"""
    from rayoptics_amd.engine import calc_psf
    out = {'what': 'rayoptics_amd.engine.calc_psf (rox_calc_psf: analyses.calc_psf as a pruned DFT, '
                   'v_mfma_f64_16x16x4), OPD grid and PSF resident in HBM',
           'bound': 'mfma fp64', 'peak_tflops': 78.6, 'sizes': []}
    for ndim, maxdim, reps in ((64, 256, 200), (256, 1024, 100), (1024, 4096, 20)):
        y, x = np.mgrid[-1:1:ndim * 1j, -1:1:ndim * 1j]
        opd = 1.5 * (x * x + y * y) + 0.4 * x * y * y
        opd[x * x + y * y > 1.0] = np.nan
        d = torch.from_numpy(opd).cuda()
        for _ in range(3):
            calc_psf(d, ndim, maxdim)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(reps):
            calc_psf(d, ndim, maxdim)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        flop = 8.0 * maxdim * ndim * (ndim + maxdim)
        out['sizes'].append({'ndim': ndim, 'maxdim': maxdim, 'ms': ms,
                             'tflops': flop / (ms * 1e-3) / 1e12,
                             'frac': flop / (ms * 1e-3) / 78.6e12})
    return out


def roofline_hits(inters, R, kern_ms):
    """the VALU-bound HITS kernel: achieved fp64 TFLOP/s by SURVEY 8(d)'s count of
    130 flop per spherical refracting intersection (5 sqrt + 8 div counted as one
    each) against the 78.6 TFLOP/s fp64 vector peak; VALU issue fraction from the
    committed PMC summary of the same kernel when present"""
    flops = 130.0 * inters
    achieved = flops / (kern_ms * 1e-3) / 1e12
    out = {'bound': 'fp64 valu', 'achieved': achieved, 'peak': 78.6, 'unit': 'TFLOP/s',
           'frac': achieved / 78.6, 'kernel': 'trace_kernel<HITS,PUPIL>', 'kernel_ms': kern_ms,
           'flop_per_intersection': 130, 'algorithmic_bytes_per_launch': R * 19,
           'hbm_GBps': R * 19 / (kern_ms * 1e-3) / 1e9}
    ppath = os.path.join(ROOT, 'profiles', 'r02_pmc_summary.json')
    if os.path.exists(ppath):
        try:
            with open(ppath) as f:
                pj = json.load(f)
            h = pj.get('HITS', {})
            if 'SQ_INSTS_VALU' in h:
                # wave-level VALU instructions x 4 cycles (fp64: 16 lanes/clk/SIMD) over
                # the SIMD-cycles of the launch (256 CUs x 4 SIMDs x 2.4 GHz)
                out['valu_insts_per_launch'] = h['SQ_INSTS_VALU']
                out['valu_issue_frac'] = h['SQ_INSTS_VALU'] * 4 / (256 * 4 * 2.4e9 * kern_ms * 1e-3)
        except Exception:
            pass
    return out


def strong_scaling(args, torch, dist, multi, world, rank, fence):
    """BASELINE configs[4]'s shape on the 44-interface lithography lens: 9 fields
    x 5 wavelengths x num^2 pupil grids, HITS, pupil-row blocks over the ranks,
    hits gathered to rank 0.  Kernel, gather and end-to-end times separately."""
    from rayoptics_amd import workloads
    from rayoptics_amd import dist as rdist
    from rayoptics_amd.engine import TraceEngine
    wl = workloads.load('litho_c5')
    eng = TraceEngine(wl.table)
    nf, nw, num = len(wl.fields), len(wl.table.wvls), args.strong_num
    plan = rdist.partition(nf, nw, num, world)
    sizes = [sum(b.row_count for b in blocks) * num for blocks in plan]
    cap = max(max(sizes), 1)
    for _rep in range(2):               # the first pass warms allocations and RCCL
        fence()
        t0 = time.perf_counter()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        xy, st, n_local = rdist.trace_blocks(eng, plan[rank], cap, wl.fields, wl.image_pts, num, wl.foc)
        e1.record()
        torch.cuda.synchronize()
        t_trace = time.perf_counter() - t0
        kern_ms = e0.elapsed_time(e1)
        fence()
        t1 = time.perf_counter()
        parts = rdist.gather_hits(xy, st)
        fence()
        t_gather = time.perf_counter() - t1
        t_all = time.perf_counter() - t0
        ok_local = int((st[:n_local] == 0).sum().item())
        del parts, xy, st
    vals = torch.tensor([kern_ms, t_trace * 1e3, t_gather * 1e3, t_all * 1e3],
                        dtype=torch.float64, device=eng.device)
    cnt = torch.tensor([n_local, ok_local], dtype=torch.float64, device=eng.device)
    if multi:
        dist.all_reduce(vals, op=dist.ReduceOp.MAX)
        dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
    K = wl.n_ifcs - 1
    rays = int(cnt[0].item())
    res = {'workload': f'litho_c5 ({wl.n_ifcs} interfaces, K={K}), {nf} fields x {nw} wvls x '
                       f'{num}x{num} pupil grids, HITS, pupil-row blocks over ranks (dist.partition)',
           'scaling': 'strong', 'ranks': world,
           'backend': (dist.get_backend() if multi else 'none (single process)'),
           'rays': rays, 'rays_through': int(cnt[1].item()),
           'rays_per_rank_max': cap,
           'kernel_ms_max_over_ranks': vals[0].item(),
           'trace_wallclock_ms_max': vals[1].item(),
           'gather_ms': vals[2].item(),
           'end_to_end_ms': vals[3].item(),
           'gather_bytes_to_root': int(17 * cap * (world - 1)),
           'rays_per_s_end_to_end': rays / (vals[3].item() * 1e-3),
           'ray_surface_per_s_kernel': rays * K / (vals[0].item() * 1e-3),
           'note': 'nominal R*K intersections (blocked rays stop early); host reassembly of the '
                   'gathered hits is not part of these times'}
    eng.close()
    return res


def reference_python():
    """the reference's own Python path on BASELINE configs[1], as timed by
    tools/time_reference.py in the build container (it cannot run on the GPU box)"""
    path = os.path.join(ROOT, 'profiles', 'reference_cpu.json')
    if not os.path.exists(path):
        return None
    with open(path) as f:
        r = json.load(f)
    return {'measured_on': r['host'], 'workload': r['workload'],
            'driver_trace_grid_rays_per_s': r['driver_trace_grid']['rays_per_s'],
            'driver_trace_grid_intersections_per_s': r['driver_trace_grid']['intersections_per_s'],
            'extrapolated_1M_ray_spot_s': r['driver_trace_grid']['extrapolated_1M_ray_spot_s'],
            'raw_rt_trace_rays_per_s': r['raw_rt_trace']['rays_per_s'],
            'raw_rt_trace_intersections_per_s': r['raw_rt_trace']['intersections_per_s'],
            'fanned': r['raw_rt_trace_fanned'],
            'source': 'profiles/reference_cpu.json (tools/time_reference.py; mirrors the '
                      "reference's own rayoptics/raytr/tests/time_trace.py)"}


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
            'host_cpu_count': os.cpu_count(), 'all_cores': allc,
            'reference_python': reference_python()}


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
