#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on BASELINE.json's configuration.

    python bench.py [--gpus N] [--steps K] [--warmup W]

``--gpus N`` with N > 1 needs one process per GPU: when this script is started
plainly (no WORLD_SIZE in the environment) it re-executes itself under
``python -m torch.distributed.run --nnodes=1 --nproc-per-node N`` on 127.0.0.1;
rank 0 prints the one JSON line on the original stdout.

Workload (configs[1]): the 13-interface double Gauss (K = 12 intersections per
ray), 1 field, 1 wavelength, 1024 x 1024 pupil grid = 1,048,576 rays per step,
FULL ray packets (the RayPkg return shape kept intact: 13 segments x 10 f64 per
ray, written SoA to HBM), rays generated on the device -- the trace_grid-shaped
entry of the C ABI.  One "step" = one such grid.  Inputs are resident in HBM
(the surface table; there are no per-ray inputs).

metric        ray-surface intersections per second, counting the intersections
              actually performed (a ray blocked at surface s contributes s, not K)
cold_ms_per_step  the same K launches issued right after an idle second, before any
              warm-up (the GPU's clocks ramp over the first ~100 launches)
roofline      HBM-bound kernel: algorithmic bytes = what one launch must write
              (80 B per appended segment + 8 B op + 3 B status, +16 B pupil)
              divided by the trace kernel's mean duration from HIP events on the
              launch stream; `traffic` is a committed PMC figure (traffic_committed_from says
              from which library build) -- null when the library has changed since
roofline_hits the HITS kernel behind spot diagrams / OPD / refocus is fp64-VALU
              bound: TFLOP/s by SURVEY 8(d)'s 130 flop per intersection against the
              78.6 TFLOP/s fp64 vector peak, and the VALU issue fraction from the
              committed PMC summary
configs       every BASELINE.json configuration at its own shape: kernel ms, rays/s,
              intersections/s, roofline fraction (HBM for FULL packets, VALU issue /
              flop rate for HITS)
spot_diagram  BASELINE's second metric at the PRODUCT boundary: wall-clock of
              rayoptics_amd.trace.trace_grid_spot from the Python call to the host
              (R_ok, 2) array, on a table-backed model
cpu_baseline  kind "reference": the reference ITSELF (staged as sourceless byte code in
              oracle/_ref by oracle/stage_reference.py), one core of this host: rt.trace over
              64 rows x 1024 rays of the timed grid (the shape of its own time_trace.py) and
              trace.trace_grid 256x256; every packet it returns is compared with the timed HIP
              launch's (parity_vs_timed_launch).  Beside it the plain-C port
              (oracle/rox_oracle.c), one thread and all cores.  Where no reference is staged:
              kind "port" with the build container's reference figure bridged by a ratio
strong_scaling  every run, any N: BASELINE configs[4]'s shape -- 9 fields x 5
              wavelengths x 2048^2 pupil grids of the 44-interface lithography
              lens (188.7 M rays) cut into pupil-row blocks over the ranks, packed
              hits (no padding for blocked rays), brought to rank 0's host memory
              (a) by the pipelined RCCL gather + copy-engine D2H per piece, (b) by every
              rank's copy engine into a shared pinned host segment over its own PCIe
              link, (c) left in rank 0's HBM, (d) round 3's un-pipelined form; each timed
              as passes between fences with the exchange inside; plus configs[3]
              sharded by field and configs[1]'s grid sharded by rows

N > 1: one rank per GPU.  The headline (`value`, `ms_per_step`, "scaling": "strong") is the
path north_star describes: BASELINE configs[4]'s spot problem (188.7 M rays) cut by pupil rows
over the ranks, packed hits gathered to rank 0 over RCCL and delivered to its host memory,
pipelined per piece, K passes between two fences -- the exchange is INSIDE the timed region;
`predicted_ms` holds DESIGN section 7's arithmetic per N.  The collective-free figure (one
FULL grid per rank, weak scaling) is kept under `weak_full`.  `strong_scaling` carries every
exchange variant incl. the N = 1 line's own grid cut by rows (`c2_sharded`).  A rank hanging
in an exchange ends the run with "ok": false and exit status 3 (the line is still printed).
"""
import argparse
import atexit
import json
import os
import socket
import sys
import tempfile
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

SPOT_FLAGS = None       # set in main (abi constants)


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
    ap.add_argument('--ref-sample-rows', type=int, default=64,
                    help='pupil rows of the timed grid the reference itself (oracle/_ref) re-traces: '
                         '64 x 1024 = 65 536 rays, ~15 s of its per-ray Python loop')
    ap.add_argument('--no-strong', action='store_true', help='skip the strong-scaling leg')
    ap.add_argument('--strong-timeout', type=float, default=300.0,
                    help='multi-rank runs: seconds after which the strong-scaling leg is given up')
    ap.add_argument('--preflight-timeout', type=float, default=20.0,
                    help='N > 1: seconds each collective of the exchange pre-flight may take')
    ap.add_argument('--no-configs', action='store_true', help='skip the per-config leg')
    ap.add_argument('--ref-fan-seconds', type=float, default=20.0,
                    help='cpu_baseline: seconds of reference tracing per process of the all-core leg')
    ap.add_argument('--ref-worker', default=None, help=argparse.SUPPRESS)
    ap.add_argument('--strong-num', type=int, default=2048,
                    help='pupil grid of the strong-scaling leg is num x num per (field, wvl)')
    return ap.parse_args()


def free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def self_launch(args):
    """--gpus N > 1 without a launcher: become `torch.distributed.run` with N local
    ranks (exec: same stdout, same exit status)"""
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1',
           f'--nproc-per-node={args.gpus}', '--master-addr', '127.0.0.1',
           '--master-port', str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    env.setdefault('OMP_NUM_THREADS', '1')
    sys.stdout.flush()
    sys.stderr.flush()
    os.execve(sys.executable, cmd, env)


def main():
    global SPOT_FLAGS
    t_process = time.perf_counter()
    args = parse()
    if args.ref_worker:                 # a process of cpu_reference_fanned(): no torch, no GPU
        return ref_worker(args.ref_worker)
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        self_launch(args)
    # the BASELINE configurations are timed as a call whose fields or outputs changed: the library's
    # short cut for a batch that repeats the stream's previous one byte for byte (no item upload,
    # ~6 us of a ~24 us configs[3] pass) is switched off for this process (read once by the library)
    os.environ.setdefault('ROX_BATCH_ALWAYS_UPLOAD', '1')
    import torch
    import torch.distributed as dist
    import rayoptics_amd  # noqa: F401
    from rayoptics_amd import abi, workloads
    from rayoptics_amd.engine import TraceEngine, make_opts, make_grid, DeviceResult
    SPOT_FLAGS = abi.INTERSECT_OBJ | abi.CHECK_APERTURES | abi.APPLY_VIGNETTING

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f'--gpus {args.gpus} but the launcher started {world} ranks')
    # rehearsal switches (a 1-GPU box cannot run RCCL with two ranks): ROX_BENCH_SHARE_GPU=1
    # puts every rank on device 0 and ROX_BENCH_BACKEND=gloo carries the collectives --
    # the N > 1 control flow, partitioning and exchanges run as they will over RCCL
    backend = os.environ.get('ROX_BENCH_BACKEND', 'nccl')
    if os.environ.get('ROX_BENCH_SHARE_GPU') == '1':
        local_rank = 0
    torch.cuda.set_device(local_rank)
    multi = world > 1 or args.force_dist
    saved_stdout = None
    if multi:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29533')
        # what RCCL has to say about a failure goes to a file per rank (NCCL_DEBUG_FILE: %h host,
        # %p pid); a failing leg's record carries its tail (nccl_debug_tail())
        os.environ.setdefault('NCCL_DEBUG', 'WARN')
        os.environ.setdefault('NCCL_DEBUG_FILE', os.path.join(
            tempfile.gettempdir(), f"rox_nccl_{os.environ['MASTER_PORT']}_%h_%p.log"))
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
    flags = SPOT_FLAGS
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

    # how many ranks the backend really connects (an all-reduce of ones)
    ranks_seen = 1
    if multi:
        ones = torch.ones(1, device=eng.device)
        dist.all_reduce(ones)
        ranks_seen = int(ones.item())

    # N > 1: the exchange's call pattern once, tiny, under a short watchdog, BEFORE anything is
    # timed: if grouped isend / irecv or an all-gather on the side stream cannot complete on this
    # node, the line says which one within seconds instead of a timed leg dying in its timeout
    preflight = None
    if multi:
        preflight = exchange_preflight(torch, dist, world, rank, eng.device, args.preflight_timeout)
        if not preflight.get('ok'):
            if rank == 0:
                print(json.dumps({'metric': 'ray-surface intersections/sec', 'value': None, 'ok': False,
                                  'n_gpus': world, 'hung_in': 'preflight: ' + str(preflight.get('hung_in')),
                                  'preflight': preflight, 'errors': {'nccl_debug': nccl_debug_tail()}}),
                      file=os.fdopen(saved_stdout, 'w'), flush=True)
            else:
                time.sleep(2.0)
            os._exit(3)

    # cold: the K launches as a caller meets them after an idle second (one launch first, so
    # that module load and allocation are not in it)
    step()
    torch.cuda.synchronize()
    time.sleep(1.0)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    cold_dt = time.perf_counter() - t0

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
        t = torch.tensor([dt, cold_dt], dtype=torch.float64, device=eng.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt, cold_dt = t[0].item(), t[1].item()

    # work actually done by this rank's grid
    inters, alg_bytes = work_of(out.status, out.fail_surf, N, abi, full=True)
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
    # ... and its tolerance-mode twin (ROX_FAST_FP64: <= 1e-10 from the reference instead of its
    # bits, tests/test_gpu_fast.py), with the deviation of this very launch from the exact one
    o_fast = make_opts(flags=flags | abi.FAST_FP64, out_mode=abi.OUT_HITS, first_surf=1, last_surf=N - 2,
                       foc=wl.foc, image_pt=wl.image_pts[fi])
    hits_f = DeviceResult(torch, eng.device, 0, R, abi.OUT_HITS, want_pupil=False, nan_fill=False)
    hits_fast_ms = eng.time_pupil_grid_sustained(fld, grid, wi, o_fast, hits_f,
                                                 max(min(args.steps, 50), 10))
    fast_check = fast_vs_exact(torch, abi, hits, hits_f)
    del hits, hits_f

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
    n_through = int(xy.shape[0])
    # ... and behind the opt-in tolerance-mode kernels (session.set_tolerance_mode: ROX_FAST_FP64 on
    # every reduced-output launch of the drop-ins; <= 1e-10 from the reference, not its bits)
    from rayoptics_amd import session as rox_session
    spot_tol_ms, n_through_tol = [], None
    rox_session.set_tolerance_mode(True)
    try:
        for _ in range(20):
            xy_t = rox_trace.trace_grid_spot(model, grid_rng, mfld, wvl_nm, wl.foc, wl.image_pts[fi])
        for _ in range(41):
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            xy_t = rox_trace.trace_grid_spot(model, grid_rng, mfld, wvl_nm, wl.foc, wl.image_pts[fi])
            spot_tol_ms.append((time.perf_counter() - t1) * 1e3)
        n_through_tol = int(xy_t.shape[0])
        spot_tol_dev = (float(np.max(np.abs(xy_t - xy))) if xy_t.shape == xy.shape else None)
    finally:
        rox_session.set_tolerance_mode(False)
    # what an interactive caller meets: the same call after the GPU has idled for a second (the
    # clocks have dropped), at the 1M-ray grid and at the 64 x 64 grid figures default to
    cold_spot = {}
    for cnum in (num, 64):
        cgrid = [np.array([-1., -1.]), np.array([1., 1.]), cnum]
        rox_trace.trace_grid_spot(model, cgrid, mfld, wvl_nm, wl.foc, wl.image_pts[fi])
        ts = []
        for _ in range(3):
            torch.cuda.synchronize()
            time.sleep(1.0)
            t1 = time.perf_counter()
            rox_trace.trace_grid_spot(model, cgrid, mfld, wvl_nm, wl.foc, wl.image_pts[fi])
            ts.append((time.perf_counter() - t1) * 1e3)
        warm = []
        for _ in range(21):
            t1 = time.perf_counter()
            rox_trace.trace_grid_spot(model, cgrid, mfld, wvl_nm, wl.foc, wl.image_pts[fi])
            warm.append((time.perf_counter() - t1) * 1e3)
        cold_spot[f'{cnum}x{cnum}'] = {'first_call_after_1s_idle_ms': float(np.median(ts)),
                                       'back_to_back_ms': float(np.median(warm))}
    # the rows of the timed launch's packets the CPU legs re-trace (the reference itself, when
    # it is staged on this host, is compared with them ray by ray)
    dev_sample = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        rr = min(args.ref_sample_rows, num)
        i0r = (num - rr) // 2
        sl = slice(i0r * num, (i0r + rr) * num)
        step()
        torch.cuda.synchronize()
        dev_sample = {'row0': i0r, 'rows': rr,
                      'seg': out.seg[:, :, sl].cpu().numpy(), 'op': out.op[sl].cpu().numpy(),
                      'status': out.status[sl].cpu().numpy(),
                      'fail_surf': out.fail_surf[sl].cpu().numpy(),
                      'pupil': out.pupil[:, sl].cpu().numpy()}
    del xy, out

    # the PSF of an OPD grid (analyses.calc_psf): the GEMM-shaped neighbour of the path,
    # on the fp64 matrix cores; device-resident, mean of back-to-back calls
    psf = None
    if rank == 0:
        try:
            psf = psf_leg(torch)
        except Exception as e:
            psf = {'error': repr(e)}

    # every BASELINE configuration at its own shape (rank 0's GPU; the others wait at the
    # next fence)
    configs = None
    if rank == 0 and not args.no_configs:
        try:
            configs = configs_leg(torch, abi, workloads)
        except Exception as e:
            configs = {'error': repr(e)}
    torch.cuda.empty_cache()

    line = None
    if rank == 0:
        traffic, traffic_source = committed_traffic(num, wl.name)
        achieved = alg_bytes / (kern_ms * 1e-3) / 1e9
        line = {
            'metric': 'ray-surface intersections/sec',
            'value': inters_all / dt * args.steps,
            'unit': 'ray-surface intersections/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'warmup_steps_run': n_w,
            'ms_per_step': dt / args.steps * 1e3,
            'cold_ms_per_step': cold_dt / args.steps * 1e3,
            'fence_ms': fence_ms,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f64', 'data': 'synthetic',
            'ranks_seen_by_backend': ranks_seen,
            'preflight': preflight,
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
                         'traffic_committed_from': traffic_source,
                         'kernel': 'trace_kernel<FULL,PUPIL>', 'kernel_ms': kern_ms,
                         'algorithmic_bytes_per_launch': alg_bytes,
                         'frac_of_measured_copy_peak_6290': achieved / 6290.0},
            'roofline_hits': roofline_hits(inters, R, hits_kern_ms),
            'roofline_hits_fast': dict(roofline_hits(inters, R, hits_fast_ms, workload='dblgauss_c2_fast'),
                                       kernel='trace_kernel<HITS,PUPIL,F_FAST> (ROX_FAST_FP64: <= 1e-10 '
                                              'from the reference, not bit-exact; opt-in)',
                                       vs_exact_same_launch=fast_check,
                                       speedup_over_exact=hits_kern_ms / hits_fast_ms),
            'spot_diagram': {'wallclock_ms': float(np.median(spot_ms)),
                             'wallclock_min_ms': float(np.min(spot_ms)), 'rays': R,
                             'rays_through': n_through, 'kernel_hits_ms': hits_kern_ms,
                             'tolerance_mode': {'wallclock_ms': float(np.median(spot_tol_ms)),
                                                'wallclock_min_ms': float(np.min(spot_tol_ms)),
                                                'rays_through': n_through_tol,
                                                'max_abs_deviation_from_exact': spot_tol_dev,
                                                'what': 'the same call after session.set_tolerance_mode(True): '
                                                        'opt-in, never the figure above'},
                             'pcie_floor_ms': n_through * 16 / 54.7e9 * 1e3,
                             'after_idle': cold_spot,
                             'what': 'rayoptics_amd.trace.trace_grid_spot(model, grid_rng, fld, wvl, '
                                     'foc, image_pt): Python call -> host (R_ok, 2) float64 array '
                                     '(survivors packed in ray order by the trace launch, written '
                                     'straight into pinned host memory; 13 MB over PCIe at ~55 GB/s '
                                     'is the floor)'},
            'configs': configs,
            'configs_batch_items_uploaded_every_pass': os.environ.get('ROX_BATCH_ALWAYS_UPLOAD') == '1',
            'configs_batch_items_in_kernel_argument_up_to': 16,
            'psf': psf,
            'cpu_baseline': None,
            'strong_scaling': None,
            'library': library_id(),
            'ok': True,
        }

    emitted = threading.Event()

    def emit():
        """the one JSON line goes to the real stdout (fd 1 was pointed at stderr while the
        communicator was being built and the collectives ran)"""
        nonlocal saved_stdout
        if rank != 0 or emitted.is_set():
            return
        emitted.set()
        # (this process, from its start to the line: imports, build check, every leg -- what a
        # driver's clock around the command sees, less the interpreter's own start-up)
        line['bench_wall_s'] = round(time.perf_counter() - t_process, 2)
        sys.stdout.flush()
        if saved_stdout is not None:
            os.dup2(saved_stdout, 1)
            os.close(saved_stdout)
            saved_stdout = None
        print(json.dumps(line), flush=True)

    # A rank that hangs in an exchange (a divergent failure leaves the others waiting in a
    # collective until the backend's own timeout kills the job) trips a watchdog: rank 0 prints
    # the line it has -- "ok": false, the leg that hung named -- and every rank ends with exit
    # status 3, so that neither the numbers already measured nor the failure are lost.
    hung_in = ['']

    def give_up():
        drop_segments()
        if rank == 0:
            line['ok'] = False
            line['hung_in'] = hung_in[0]
            line.setdefault('errors', {})['nccl_debug'] = nccl_debug_tail()
            if not line.get('strong_scaling'):
                line['strong_scaling'] = {'error': f'timed out after {args.strong_timeout} s in '
                                                   f'{hung_in[0]}: a rank hung in an exchange'}
            emit()
        else:
            # the launcher ends every rank as soon as one fails: rank 0 prints first
            time.sleep(3.0)
        os._exit(3)

    def guarded(what, fn):
        hung_in[0] = what
        dog = threading.Timer(args.strong_timeout, give_up) if multi else None
        if dog:
            dog.daemon = True
            dog.start()
        try:
            return fn()
        except Exception as e:
            import traceback
            traceback.print_exc(file=sys.stderr)
            return {'error': repr(e), 'nccl_debug': nccl_debug_tail() if multi else None}
        finally:
            if dog:
                dog.cancel()

    # N > 1: the headline is the path north_star describes -- the fixed-size spot problem cut
    # over the ranks with the exchange INSIDE the timed region (the faster of the pipelined
    # RCCL gather and the shared-host-segment delivery, both timed) -- and the collective-free
    # one-grid-per-rank figure above moves to `weak_full`.  (--force-dist rehearses this with
    # one rank.)  If the leg fails, the weak figure stays the headline and says so.
    if multi and not args.no_strong:
        head = guarded('strong_headline', lambda: strong_headline(args, torch, dist, multi, world, rank, fence))
        if rank == 0:
            line['weak_full'] = {k: line[k] for k in ('value', 'ms_per_step', 'cold_ms_per_step', 'rays_per_s',
                                                      'config', 'scaling')}
            if 'error' in head:
                line['headline'] = 'weak_full (the strong-scaled leg failed: ' + head['error'] + ')'
                line['ok'] = False
            else:
                line['headline'] = 'strong-scaled spot problem, exchange inside the timed region'
                line['value'] = head['intersections_per_step'] / (head['ms_per_step'] * 1e-3)
                line['ms_per_step'] = head['ms_per_step']
                line['rays_per_s'] = head['rays_per_step'] / (head['ms_per_step'] * 1e-3)
                line['scaling'] = 'strong'
                line['cold_ms_per_step'] = None
                how = {'rccl': 'grouped send/recv of the packed pairs to rank 0 over RCCL -> rank 0\'s '
                               'copy-engine D2H',
                       'host': 'every rank\'s copy-engine D2H over its own PCIe link into a shared pinned '
                               'host segment (RCCL carries the per-piece counts only)'}[head['exchange']]
                line['config'] = {
                    'workload': head['workload'] + ' (BASELINE.json configs[4]); each step = launches -> '
                                'per-piece counts -> ' + how + ' -> rank 0 holds every (field, wvl) '
                                'grid\'s (R_ok, 2) host array; pipelined per piece of <= 4 Mi rays',
                    'rays_per_step': head['rays_per_step'],
                    'intersections_per_step': head['intersections_per_step'],
                    'out_mode': 'HITS_COMPACT (two-pass: HITS + pack)',
                    'exchange': head['exchange'] + ', pipelined (the faster of rccl / host, both timed)',
                    'sharding': f'pupil-row blocks over {world} ranks', 'pairs_to_host': head['pairs'],
                    'pieces_per_rank': head['pieces_per_rank'], 'stages': head['stages']}
                line['predicted_ms'] = head['predicted_ms']
                line.update(scaling_keys(head.get('one_gpu_same_problem_ms'), head['ms_per_step'], world))
                line['one_gpu_same_problem'] = (
                    'rank 0 alone, same problem, same exchange code (' + head['exchange'] + '), same run'
                    + ('' if not head.get('one_gpu_error') else ' -- FAILED: ' + head['one_gpu_error']))
                if head.get('configs3_by_field'):
                    line['configs3_by_field'] = head['configs3_by_field']
                if head.get('configs3_by_rows'):
                    line['configs3_by_rows'] = head['configs3_by_rows']
                # north_star's own words -- "an RCCL gather over xGMI of the image-plane hits" -- are the
                # rccl_device figure: the gather alone, result left in rank 0's HBM
                line['rccl_gather_only_ms'] = head['ms_per_step_by_exchange'].get('rccl_device')
                line['strong_headline'] = {k: head[k] for k in ('exchange', 'ms_per_step_by_exchange', 'errors',
                                                                  'last_pass_phases_ms_rank0', 'grids_delivered')}

    # every run: the fixed-size problems in every exchange variant (extra; the line above is
    # complete without it)
    if not args.no_strong:
        strong = guarded('strong_scaling', lambda: strong_scaling(args, torch, dist, multi, world, rank,
                                                                   fence, ranks_seen))
        if rank == 0:
            line['strong_scaling'] = strong
            if world == 1 and isinstance(strong, dict) and isinstance(strong.get('c5'), dict):
                # the N = 1 point of the N > 1 headline's curve: the same strong problem, best
                # exchange, end to end -- same field names as a `--gpus N` line carries
                c5 = strong['c5']
                legs = {k: c5[k]['end_to_end_ms'] for k in ('rccl', 'host')
                        if isinstance(c5.get(k), dict) and 'end_to_end_ms' in c5[k]}
                if legs:
                    bestx = min(legs, key=legs.get)
                    line['strong_equiv'] = dict(
                        scaling_keys(legs[bestx], legs[bestx], 1), ms_per_step=legs[bestx], exchange=bestx,
                        ms_per_step_by_exchange=legs, workload=c5['workload'] + ' (BASELINE.json configs[4])',
                        intersections_per_step=c5['intersections'],
                        value=c5['intersections'] / (legs[bestx] * 1e-3),
                        what='what a `--gpus N` line reports under value / ms_per_step, at N = 1: draw the '
                             'scaling curve from this and the N > 1 lines, not from this line\'s `value`')
                    tm = c5.get('rccl_tolerance_mode')
                    if isinstance(tm, dict) and 'end_to_end_ms' in tm:
                        # the same problem behind the tolerance-mode kernels (ROX_FAST_FP64; opt-in, not the headline)
                        line['strong_equiv']['ms_per_step_tolerance_mode'] = tm['end_to_end_ms']

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        line['cpu_baseline'] = cpu_baseline(wl, fld, wi, opts, num, args.cpu_sample_rows,
                                            fi, dev_sample)
    emit()
    if multi:
        if saved_stdout is not None:            # other ranks: keep their fd 1 on stderr
            os.close(saved_stdout)
        sys.stdout.flush()
        os.dup2(2, 1)                           # teardown chatter does not belong on stdout
        dist.destroy_process_group()


# ----------------------------------------------------------------------------- helpers
def work_of(status, fail_surf, N, abi, full):
    """(intersections actually performed, algorithmic bytes) of one traced batch from its
    status / fail_surf arrays (torch, on the device): a ray that got through did N - 1
    intersections, a ray that failed at surface s did s; FULL packets hold N segments for
    a survivor, s (missed) or s + 1 (blocked / TIR) for a failed ray"""
    import torch
    ok = status == abi.OK
    fs = fail_surf.to(torch.int64)
    n_ok = int(ok.sum().item())
    R = status.numel()
    inters = n_ok * (N - 1) + int(fs[~ok].sum().item())
    if not full:
        return inters, R * 19
    missed = status == abi.MISSED_SURFACE
    nseg = n_ok * N + int(fs[missed].sum().item()) + int((fs[~ok & ~missed] + 1).sum().item())
    return inters, nseg * 80 + R * (8 + 1 + 2 + 16)


def library_id():
    """the digest build.py stamps next to libroxtrace.so (sources + headers + flags)"""
    try:
        with open(os.path.join(ROOT, 'ray-optics_amd', 'libroxtrace.so.srchash')) as f:
            return {'source_hash': f.read().strip()[:16]}
    except OSError:
        return {'source_hash': None}


def committed_traffic(num, workload):
    """roofline.traffic is not measured in this run (PMC passes need rocprofv3): it is the
    committed figure of tools/make_traffic.py, valid for the workload / grid AND the library
    build it was taken with -- says so, and is withheld when the kernels have changed since"""
    tpath = os.path.join(ROOT, 'profiles', 'traffic.json')
    if not os.path.exists(tpath):
        return None, 'no committed PMC figure'
    with open(tpath) as f:
        tj = json.load(f)
    if tj.get('num') != num or tj.get('workload') != workload:
        return None, 'committed PMC figure is for another workload / grid'
    have = library_id()['source_hash']
    src = f"profiles/traffic.json ({tj.get('source')}; separate rocprofv3 --pmc passes)"
    if tj.get('library_source_hash') and tj['library_source_hash'] != have:
        return None, src + f" -- taken with library {tj['library_source_hash']}, this is {have}: withheld"
    if not tj.get('library_source_hash'):
        src += ' -- library build of that run not recorded'
    return tj.get('hbm_bytes_per_launch'), src


def committed_pmc(workload):
    """VALU wave instructions per intersection of the HITS kernel from the committed PMC
    summaries (profiles/valu_per_intersection.json, tools/pmc_summary.py)"""
    p = os.path.join(ROOT, 'profiles', 'valu_per_intersection.json')
    if not os.path.exists(p):
        return None
    with open(p) as f:
        return json.load(f).get(workload)


def psf_leg(torch):
    """rox_calc_psf at three sizes: ms per call and fp64 TFLOP/s by 8 M n (n + M) flop
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


def fast_vs_exact(torch, abi, exact, fast):
    """the tolerance-mode HITS launch against the bit-exact one of the same grid: status flips
    and the worst scaled deviation max |a - b| / max(1, |a|) over the rays both carry through"""
    st_e, st_f = exact.status, fast.status
    flips = int((st_e != st_f).sum().item())
    ok = (st_e == abi.OK) & (st_f == abi.OK)
    a, b = exact.seg[:, ok], fast.seg[:, ok]
    dev = float(((a - b).abs() / a.abs().clamp(min=1.0)).max().item()) if int(ok.sum().item()) else 0.0
    return {'status_flips': flips, 'rays_through': int(ok.sum().item()), 'max_scaled_deviation': dev,
            'tolerance': 1e-10}


def roofline_hits(inters, R, kern_ms, workload='dblgauss_c2'):
    """the VALU-bound HITS kernel: achieved fp64 TFLOP/s by SURVEY 8(d)'s count of
    130 flop per spherical refracting intersection (5 sqrt + 8 div counted as one
    each) against the 78.6 TFLOP/s fp64 vector peak; VALU issue fraction from the
    committed PMC instruction count per intersection of the same kernel when present"""
    flops = 130.0 * inters
    achieved = flops / (kern_ms * 1e-3) / 1e12
    out = {'bound': 'fp64 valu', 'achieved': achieved, 'peak': 78.6, 'unit': 'TFLOP/s',
           'frac': achieved / 78.6, 'kernel': 'trace_kernel<HITS,PUPIL>', 'kernel_ms': kern_ms,
           'flop_per_intersection': 130, 'algorithmic_bytes_per_launch': R * 19,
           'hbm_GBps': R * 19 / (kern_ms * 1e-3) / 1e9}
    pmc = committed_pmc(workload)
    if pmc:
        # wave-level VALU instructions x 4 cycles (fp64: 16 lanes/clk/SIMD) over the
        # SIMD-cycles of the launch (256 CUs x 4 SIMDs x 2.4 GHz)
        insts = pmc['valu_wave_insts_per_intersection'] * inters
        out['valu_insts_per_launch'] = insts
        out['valu_issue_frac'] = insts * 4 / (256 * 4 * 2.4e9 * kern_ms * 1e-3)
        out['valu_source'] = pmc.get('source')
    return out


def configs_leg(torch, abi, workloads):
    """Every BASELINE.json configuration at its own shape on this GPU: one pass = every
    (field, wavelength) grid of the configuration, in ONE launch (rox_trace_pupil_grids; the
    figure for one launch per grid, back to back, is kept beside it); steady-state ms
    per pass from events on the launch stream (median of 5 timed batches after >= 100 ms of
    untimed passes).  FULL packets where they fit HBM (config 5's 666 GB do not)."""
    from rayoptics_amd.engine import TraceEngine, make_opts, make_grid, DeviceResult
    specs = [
        ('c1', 'singlet_c1', [0], [None], 64, True,
         'configs[0]: singlet (4 interfaces), 1 field, 1 wvl, 64x64'),
        ('c2', 'dblgauss_c2', [0], [None], 1024, True,
         'configs[1]: double Gauss (13 interfaces), 1 field, 1 wvl, 1024x1024 (the main line)'),
        ('c3', 'nikkor_c3', [0, 1, 2], [0, 1, 2], 512, True,
         'configs[2] stand-in: 29-interface zoom with 4 even aspheres (.roa), 3 fields x 3 wvls x 512x512'),
        ('c3_zmx', 'zmx_evenasph_c3', [0, 1, 2], [0, 1, 2], 512, True,
         'configs[2]: Zemax .zmx import US08427765-1.ZMX, 13 interfaces incl. an EVENASPH, '
         '3 fields x 3 wvls x 512x512'),
        ('c4', 'rc_telescope_c4', [0, 1, 2, 3, 4], [None], 256, True,
         'configs[3]: Ritchey-Chretien mirror pair + field stop (5 interfaces), 5 fields x 256x256'),
        ('c5', 'litho_c5', list(range(9)), list(range(5)), 2048, False,
         'configs[4]: 44-interface lithography lens, 9 fields x 5 wvls x 2048x2048 (HITS: FULL packets '
         'would be 666 GB)'),
    ]
    out = {}
    for key, name, fis, wis, num, do_full, what in specs:
        wl = workloads.load(name)
        N = wl.n_ifcs
        eng = TraceEngine(wl.table)
        wis = [wl.ref_wvl_idx if w is None else w for w in wis]
        R = num * num
        grid = make_grid((-1., -1.), (1., 1.), num)
        wide = [abi.INTERSECT_OBJ if (f.kind != abi.FLD_EPD_WIDE and f.z_dir0 != 0.0) else 0
                for f in wl.fields]
        pairs = [(f, w) for f in fis for w in wis]
        rec = {'what': what, 'workload': name, 'interfaces': N, 'grids': len(pairs),
               'rays': R * len(pairs)}
        for mode, label in ((abi.OUT_HITS, 'hits'), (abi.OUT_HITS, 'hits_fast'), (abi.OUT_FULL, 'full')):
            if mode == abi.OUT_FULL and not do_full:
                continue
            # (hits_fast: the opt-in tolerance-mode kernels, ROX_FAST_FP64 -- never a headline figure.
            # FULL launches of these configurations keep the exact kernels under the flag too: none
            # of them is made mostly of aspheres, roxtrace.hip use_fast())
            fast = abi.FAST_FP64 if label == 'hits_fast' else 0
            # every grid of the configuration has its own output buffers (a pass leaves the
            # whole configuration's result in HBM)
            ress = [DeviceResult(torch, eng.device, eng.num_segments(0), R, mode,
                                 want_pupil=(mode == abi.OUT_FULL), nan_fill=False) for _ in pairs]
            optl = [make_opts(flags=(SPOT_FLAGS & ~abi.INTERSECT_OBJ) | wide[f] | fast, out_mode=mode,
                              first_surf=1, last_surf=N - 2, foc=wl.foc, image_pt=wl.image_pts[f])
                    for f, _w in pairs]
            fldl = [wl.fields[f] for f, _w in pairs]
            wvll = [w for _f, w in pairs]

            def looped_pass(count=False):       # one launch per (field, wavelength)
                inters = nbytes = 0
                for fl, w, o, res in zip(fldl, wvll, optl, ress):
                    eng.trace_pupil_grid(fl, grid, w, o, out=res)
                    if count:
                        i, b = work_of(res.status, res.fail_surf, N, abi, full=(mode == abi.OUT_FULL))
                        inters += i
                        nbytes += b
                return inters, nbytes

            def batched_pass():                 # rox_trace_pupil_grids: the configuration in one launch
                eng.trace_pupil_grids(fldl, wvll, grid, optl, outs=ress)

            def steady_ms(one_pass):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                n_warm = 0
                while (time.perf_counter() - t0) < 0.1 or n_warm < 2:
                    one_pass()
                    n_warm += 1
                    if n_warm % 16 == 0:
                        torch.cuda.synchronize()
                torch.cuda.synchronize()
                per = max(1, min(50, int(0.02 / max((time.perf_counter() - t0) / n_warm, 1e-6))))
                ts = []
                for _ in range(5):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(per):
                        one_pass()
                    e1.record()
                    torch.cuda.synchronize()
                    ts.append(e0.elapsed_time(e1) / per)
                return sorted(ts)[len(ts) // 2]
            inters, nbytes = looped_pass(count=True)
            ms_loop = steady_ms(looped_pass)
            ms = ms_loop
            r = {}
            if len(pairs) > 1:
                ms = steady_ms(batched_pass)
                r['launches_per_pass'] = 1
                r['kernel_ms_per_pass_one_launch_per_grid'] = ms_loop
            r.update({'kernel_ms_per_pass': ms, 'rays_per_s': R * len(pairs) / (ms * 1e-3),
                      'intersections': inters, 'intersections_per_s': inters / (ms * 1e-3)})
            if mode == abi.OUT_FULL:
                gbps = nbytes / (ms * 1e-3) / 1e9
                r.update({'bound': 'hbm', 'algorithmic_bytes': nbytes, 'GBps': gbps,
                          'frac_of_8000': gbps / 8000.0, 'frac_of_measured_copy_peak_6290': gbps / 6290.0})
            else:
                r.update({'bound': 'fp64 valu', 'tflops_130_per_intersection': 130.0 * inters / (ms * 1e-3) / 1e12})
                pmc = committed_pmc(name + ('_fast' if fast else ''))
                if pmc:
                    r['valu_issue_frac'] = (pmc['valu_wave_insts_per_intersection'] * inters * 4 /
                                            (256 * 4 * 2.4e9 * ms * 1e-3))
                    r['valu_source'] = pmc.get('source')
            rec[label] = r
            del ress
        out[key] = rec
        eng.close()
        torch.cuda.empty_cache()
    return out


# measured on one MI355X (profiles/r04_pack_crossover.jsonl, profiles/r02_pcie_store.jsonl) and
# the link rates DESIGN section 7 prices the exchanges with
# shared host segments that exist right now (files under /dev/shm): a run that gives up on a
# hung exchange, or dies, must not leave gigabytes of tmpfs behind
LIVE_SEGMENTS = set()


def drop_segments():
    for path in list(LIVE_SEGMENTS):
        try:
            os.unlink(path)
        except OSError:
            pass
        LIVE_SEGMENTS.discard(path)


atexit.register(drop_segments)


PRED = {'c5_kernels_ms_one_gpu': 68.2, 'pcie_GBps': 55.0, 'xgmi_link_GBps': 153.0,
        'piece_tail_ms': 0.3, 'stage_sync_ms_per_stage': 0.06}


def predicted_ms(pairs_bytes, kernels_ms_one_gpu, stages_of):
    """DESIGN section 7's arithmetic for the pipelined exchanges, per N: the kernels split N
    ways; rccl = 1/N of the pairs per peer over its own xGMI link, all of them over rank 0's
    one PCIe link; host = every rank's share over its own PCIe link; a pipelined run ends
    max(kernels, transfers) + the last piece's copy + one host sync per stage"""
    out = {'assumptions': dict(PRED, pairs_bytes=pairs_bytes, kernels_ms_one_gpu=kernels_ms_one_gpu),
           'rccl_to_host': {}, 'host_segment': {}, 'rccl_device_resident': {}}
    for n in (1, 2, 4, 8):
        kern = kernels_ms_one_gpu / n
        xgmi = pairs_bytes / n / (PRED['xgmi_link_GBps'] * 1e6) if n > 1 else 0.0
        sync = PRED['stage_sync_ms_per_stage'] * stages_of(n)
        tail = PRED['piece_tail_ms']
        out['rccl_to_host'][str(n)] = max(kern, xgmi, pairs_bytes / (PRED['pcie_GBps'] * 1e6)) + tail + sync
        out['host_segment'][str(n)] = max(kern, pairs_bytes / n / (PRED['pcie_GBps'] * 1e6)) + tail + sync
        out['rccl_device_resident'][str(n)] = max(kern, xgmi) + sync + 0.3
    return out


class SpotProblem:
    """one fixed-size spot-diagram problem (every (field, wavelength) grid of a workload) cut
    over the ranks: the engine, the plan, and the intersections one pass actually performs"""

    def __init__(self, torch, dist, multi, world, rank, name, num, by, field_idx=None, n_wvls=None,
                 group=None):
        from rayoptics_amd import abi, workloads
        self.group = group      # None = the default group; a one-rank group = rank 0 on its own
        from rayoptics_amd import dist as rdist
        from rayoptics_amd.engine import TraceEngine, make_opts, make_grid, DeviceResult
        self.torch, self.dist, self.multi, self.world, self.rank = torch, dist, multi, world, rank
        self.name, self.num, self.by = name, num, by
        self.wl = wl = workloads.load(name)
        self.eng = eng = TraceEngine(wl.table)
        self.fields = list(wl.fields) if field_idx is None else [wl.fields[i] for i in field_idx]
        self.image_pts = list(wl.image_pts) if field_idx is None else [wl.image_pts[i] for i in field_idx]
        self.nf, self.nw = len(self.fields), (len(wl.table.wvls) if n_wvls is None else n_wvls)
        self.plan = rdist.partition(self.nf, self.nw, num, world, by)
        self.caps = [rdist.rays_of(b, num) for b in self.plan]
        self.K = wl.n_ifcs - 1
        self.pieces, self.order = rdist.schedule(self.plan, num, self.nw)
        # untimed: what this rank's blocks really do (a ray blocked at surface s did s
        # intersections) -- plain HITS launches with status / fail_surf, summed over the ranks
        N = wl.n_ifcs
        inters = 0
        for b in self.plan[rank]:
            R = b.row_count * num
            res = DeviceResult(torch, eng.device, 0, R, abi.OUT_HITS, want_pupil=False, nan_fill=False)
            o = make_opts(flags=SPOT_FLAGS, out_mode=abi.OUT_HITS, first_surf=1, last_surf=N - 2,
                          foc=wl.foc, image_pt=self.image_pts[b.fi])
            eng.trace_pupil_grid(self.fields[b.fi], make_grid((-1., -1.), (1., 1.), num, row_begin=b.row_begin,
                                                            row_count=b.row_count), b.wi, o, out=res)
            inters += work_of(res.status, res.fail_surf, N, abi, full=False)[0]
            del res
        tot = torch.tensor([float(inters)], dtype=torch.float64, device=eng.device)
        if multi:
            dist.all_reduce(tot)
        self.intersections = int(tot.item())
        self.what = (f'{name} ({wl.n_ifcs} interfaces, K={self.K}), {self.nf} fields x {self.nw} wvls x '
                     f'{num}x{num} pupil grids = {sum(self.caps)} rays, packed hits, partition by {by}')

    def segment(self, tag, fence):
        """the shared pinned host segment of exchange='host' (one region per grid); collective"""
        from rayoptics_amd import dist as rdist
        torch, dist, rank = self.torch, self.dist, self.rank
        name = f"rox_seg_{os.environ.get('MASTER_PORT', '0')}_{self.name}_{self.num}_{tag}"
        err = torch.zeros(1, device=self.eng.device)
        seg, first_err = None, None
        if rank == 0:
            try:
                seg = rdist.HostSegment.for_grids(self.eng, name, self.nf * self.nw, self.num, rank, create=True)
            except Exception as e:          # every rank must learn of it
                err += 1
                first_err = repr(e)
        if self.multi:
            dist.all_reduce(err)
        if err.item() > 0:
            raise RuntimeError(first_err if rank == 0 else 'rank 0 could not create the segment')
        if rank != 0:
            seg = rdist.HostSegment.for_grids(self.eng, name, self.nf * self.nw, self.num, rank, create=False)
        fence()
        LIVE_SEGMENTS.add(seg.path)
        return seg

    def run(self, exchange, segment=None, result_on='host', pipeline=True, timings=None, tolerance_mode=False):
        from rayoptics_amd import abi, dist as rdist
        wl = self.wl
        flags = (SPOT_FLAGS | abi.FAST_FP64) if tolerance_mode else None
        return rdist.trace_spot_sharded(self.eng, self.fields, self.image_pts, self.nw, self.num, wl.foc,
                                        flags=flags, by=self.by, exchange=exchange, segment=segment,
                                        timings=timings, pipeline=pipeline, result_on=result_on,
                                        group=self.group)

    def kernel_ms(self):
        """this rank's launches alone (packed hits appended into HBM), events on the launch stream"""
        from rayoptics_amd import dist as rdist
        torch, wl = self.torch, self.wl
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        probe = rdist.trace_blocks(self.eng, self.plan[self.rank], self.num, self.fields, self.image_pts, wl.foc)
        e1.record()
        probe.counts()
        return e0.elapsed_time(e1)

    def timed(self, fence, steps, warmup, **kw):
        """W untimed passes, then exactly K passes between two fences: each pass ends with rank 0
        holding every grid's (R_ok, 2) array -- the exchange is INSIDE the timed region"""
        tm = {}
        n_views = 0
        for _ in range(warmup):
            v = self.run(timings=tm, **kw)
            del v
        fence()
        t0 = time.perf_counter()
        for _ in range(steps):
            v = self.run(timings=tm, **kw)
            n_views = 0 if v is None else len(v)
            del v
        fence()
        dt = time.perf_counter() - t0
        t = self.torch.tensor([dt], dtype=self.torch.float64, device=self.eng.device)
        if self.multi:
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        tm = dict(tm)
        tm['grids_delivered'] = n_views if self.rank == 0 else None
        return t.item() / steps * 1e3, tm

    def stages_at(self, n):
        """pipeline stages (= pieces of the busiest rank) if this problem ran on n ranks"""
        from rayoptics_amd import dist as rdist
        return max(len(o) for o in rdist.schedule(rdist.partition(self.nf, self.nw, self.num, n, self.by),
                                                  self.num, self.nw)[1])

    def close(self):
        self.eng.close()
        self.torch.cuda.empty_cache()


def nccl_debug_tail(max_bytes=4000):
    """the tail of every NCCL_DEBUG_FILE this run's ranks wrote on this host (WARN level)"""
    import glob
    pat = os.environ.get('NCCL_DEBUG_FILE')
    if not pat:
        return None
    out = {}
    for path in sorted(glob.glob(pat.replace('%h', '*').replace('%p', '*'))):
        try:
            with open(path, 'rb') as f:
                f.seek(0, os.SEEK_END)
                n = f.tell()
                f.seek(max(0, n - max_bytes))
                txt = f.read().decode(errors='replace').strip()
            if txt:
                out[os.path.basename(path)] = txt
        except OSError:
            pass
    return out or None


def exchange_preflight(torch, dist, world, rank, device, timeout_s, skip=()):
    """The collectives the sharded spot exchange is made of (dist.trace_spot_sharded), once and
    tiny, each under its own watchdog: (1) an all-gather of one count word issued on a SIDE
    stream that waits for an event of the launch stream, copied to pinned memory behind it;
    (2) grouped isend / irecv of a row slice from every rank to rank 0 on that stream
    (batch_isend_irecv: 7 peers -> 7 xGMI links at N = 8); (3) an all-reduce (the fence).
    Returns {'ok', 'steps_ms': {...}} or {'ok': False, 'hung_in': step} -- a step that does not
    return within `timeout_s` ends the run (exit status 3) with that record in the line.
    device=None runs the same calls without streams (CPU, gloo: tests/test_dist_gloo.py, where
    `skip` lets one rank stay away from a step to show what a hang looks like)."""
    import contextlib
    cuda = device is not None
    done = {}
    state = {'step': None}
    failed = threading.Event()

    def watchdog():
        t_end = time.time() + timeout_s
        while time.time() < t_end:
            if state['step'] is None:
                return
            time.sleep(0.05)
        failed.set()

    def run(name, fn):
        state['step'] = name
        dog = threading.Thread(target=watchdog, daemon=True)
        dog.start()
        box = {}

        def body():
            try:
                t0 = time.perf_counter()
                if name not in skip:
                    fn()
                if cuda:
                    torch.cuda.synchronize()
                box['ms'] = (time.perf_counter() - t0) * 1e3
            except Exception as e:      # noqa: BLE001
                box['error'] = repr(e)
        th = threading.Thread(target=body, daemon=True)
        th.start()
        while th.is_alive() and not failed.is_set():
            th.join(0.05)
        state['step'] = None
        if failed.is_set() and th.is_alive():
            return False
        if 'error' in box:
            done[name] = {'error': box['error']}
            return False
        done[name] = round(box['ms'], 3)
        return True

    side = None
    if cuda:
        torch.cuda.set_device(device)
        side = torch.cuda.Stream(device=device)
    on_side = (lambda: torch.cuda.stream(side)) if cuda else contextlib.nullcontext
    # (gloo -- CPU tests, the one-GPU rehearsal -- carries host tensors, as dist._wire has it)
    wire = device if (cuda and dist.get_backend() == 'nccl') else 'cpu'
    word = torch.full((1,), float(rank + 1), dtype=torch.float64, device=wire)
    words = torch.zeros(world, dtype=torch.float64, device=wire)
    pinned = torch.zeros(world, dtype=torch.float64)
    if cuda:
        pinned = pinned.pin_memory()
    rows = torch.full((2, 256), float(rank), dtype=torch.float64, device=wire)
    inbox = torch.zeros(world, 2, 256, dtype=torch.float64, device=wire) if rank == 0 else None

    def all_gather_on_side_stream():
        if cuda:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(device))
            side.wait_event(ev)
        with on_side():
            dist.all_gather_into_tensor(words, word)
            pinned.copy_(words, non_blocking=True)
        if cuda:
            side.synchronize()
        if [float(x) for x in pinned.tolist()] != [float(r + 1) for r in range(world)]:
            raise RuntimeError(f'all-gather returned {pinned.tolist()}')

    def grouped_p2p_to_rank0():
        with on_side():
            ops = []
            if rank == 0:
                ops = [dist.P2POp(dist.irecv, inbox[r], r) for r in range(1, world)]
            else:
                ops = [dist.P2POp(dist.isend, rows, 0)]
            if ops:
                for w in dist.batch_isend_irecv(ops):
                    w.wait()
        if cuda:
            side.synchronize()
        if rank == 0 and world > 1:
            got = [float(inbox[r, 0, 0].item()) for r in range(1, world)]
            if got != [float(r) for r in range(1, world)]:
                raise RuntimeError(f'grouped recv delivered {got}')

    def all_reduce_fence():
        t = torch.ones(1, device=device if cuda else 'cpu')
        dist.all_reduce(t)
        if int(t.item()) != world:
            raise RuntimeError(f'all-reduce of ones gave {t.item()} on {world} ranks')

    for name, fn in (('all_gather_on_side_stream', all_gather_on_side_stream),
                     ('grouped_isend_irecv_to_rank0', grouped_p2p_to_rank0),
                     ('all_reduce_fence', all_reduce_fence)):
        if not run(name, fn):
            return {'ok': False, 'hung_in': name, 'steps_ms': done, 'timeout_s': timeout_s}
    return {'ok': True, 'steps_ms': done, 'timeout_s': timeout_s}


def scaling_keys(one_gpu_same_problem_ms, ms_per_step, world):
    """the keys that make a `--gpus N` line readable on its own: what ONE GPU needs for the same
    fixed-size problem through the same code (measured in the same run, on rank 0, while the
    other ranks wait), and what the N ranks made of it.  speedup = one_gpu_same_problem_ms /
    ms_per_step; efficiency = speedup / n_gpus.  The curve over N is drawn from `ms_per_step`
    (or `value`) of the strong-scaled lines and these keys -- never across the N = 1 line's own
    `value`, which is a different workload (README, "Reading the bench line")."""
    if not one_gpu_same_problem_ms or not ms_per_step:
        return {'one_gpu_same_problem_ms': one_gpu_same_problem_ms, 'speedup': None, 'efficiency': None}
    sp = one_gpu_same_problem_ms / ms_per_step
    return {'one_gpu_same_problem_ms': one_gpu_same_problem_ms, 'speedup': sp, 'efficiency': sp / world}


def solo_same_problem(args, torch, dist, multi, rank, fence, solo_group, name, num, by, exchange, steps):
    """rank 0 alone on the SAME strong problem (all pieces on its GPU, same exchange code, world
    = 1 through a one-rank group) while the other ranks wait at the fence: ms per pass"""
    ms, err = None, None
    if rank == 0:
        prob = None
        try:
            prob = SpotProblem(torch, dist, False, 1, 0, name, num, by, group=solo_group)
            lf = torch.cuda.synchronize
            seg = prob.segment('solo', lf) if exchange == 'host' else None
            try:
                ms = prob.timed(lf, steps, 1, exchange=exchange, segment=seg)[0]
            finally:
                if seg is not None:
                    LIVE_SEGMENTS.discard(seg.path)
                    seg.close(unlink=True)
        except Exception as e:      # noqa: BLE001
            import traceback
            traceback.print_exc(file=sys.stderr)
            err = repr(e)
        finally:
            if prob is not None:
                prob.close()
    fence()
    return ms, err


def strong_headline(args, torch, dist, multi, world, rank, fence):
    """N > 1: BASELINE configs[4]'s spot problem cut by pupil rows over the ranks and delivered to
    rank 0 as host arrays -- K passes between two fences, the exchange inside the timed region
    -- by both exchanges the path has: `rccl` (pipelined grouped send/recv of the packed pairs
    to rank 0 over xGMI, then rank 0's copy engine over ITS PCIe link) and `host` (every rank's
    copy engine over its OWN PCIe link into a shared pinned host segment; the only collective
    is the per-piece count exchange).  The headline is the faster one: rank 0's single PCIe
    link carries all 2.2 GB in the first, 1/N of them in the second."""
    prob = SpotProblem(torch, dist, multi, world, rank, 'litho_c5', args.strong_num, 'rows')
    try:
        legs, errors = {}, {}
        warm = max(args.warmup, 2)
        legs['rccl'] = prob.timed(fence, args.steps, warm, exchange='rccl')
        seg = None
        try:
            seg = prob.segment('headline', fence)
            legs['host'] = prob.timed(fence, args.steps, warm, exchange='host', segment=seg)
        except Exception as e:
            import traceback
            traceback.print_exc(file=sys.stderr)
            errors['host'] = repr(e)
        finally:
            if seg is not None:
                LIVE_SEGMENTS.discard(seg.path)
                seg.close(unlink=(rank == 0))
        best = min(legs, key=lambda k: legs[k][0])      # (max-over-ranks times: the same on every rank)
        ms, tm = legs[best]
        # the gather alone (pairs left in rank 0's HBM): beside the two, never the headline -- the
        # consumer of a spot diagram is host code
        by_exchange = {k: v[0] for k, v in legs.items()}
        try:
            by_exchange['rccl_device'] = prob.timed(fence, args.steps, 1, exchange='rccl', result_on='device')[0]
        except Exception as e:
            import traceback
            traceback.print_exc(file=sys.stderr)
            errors['rccl_device'] = repr(e)
        pairs = tm['pairs_total']
        phases = ('trace_ms', 'stage_sync_ms', 'gather_ms', 'd2h_ms', 'reassembly_ms')
        # the same problem on ONE GPU in the same run (collective: every rank makes the group)
        solo_group = dist.new_group(ranks=[0]) if multi else None
        one_ms, one_err = solo_same_problem(args, torch, dist, multi, rank, fence, solo_group, 'litho_c5',
                                            args.strong_num, 'rows', best, max(2, min(args.steps, 5)))
        c4, c4r = None, None
        if world == 4:
            # BASELINE configs[3] as north_star words it -- shard-by-field over 4 GPUs (5 fields:
            # 2/1/1/1, at most 62.5 % efficient by construction) -- and, SURVEY 8(e)'s "or split
            # rows for balance", the same five grids cut by pupil rows (320 rows per rank)
            def c4_leg(by, note):
                c4p = SpotProblem(torch, dist, multi, world, rank, 'rc_telescope_c4', 256, by)
                try:
                    c4_ms, c4_tm = c4p.timed(fence, max(args.steps, 5), max(args.warmup, 2), exchange='rccl')
                    c4_one, c4_err = solo_same_problem(args, torch, dist, multi, rank, fence, solo_group,
                                                       'rc_telescope_c4', 256, by, 'rccl', max(args.steps, 5))
                    return dict(scaling_keys(c4_one, c4_ms, world), ms_per_step=c4_ms, exchange='rccl',
                                workload=c4p.what + ' (BASELINE.json configs[3], ' + note + ')',
                                rays_per_rank=c4p.caps,
                                intersections_per_step=c4p.intersections, pairs=c4_tm['pairs_total'],
                                one_gpu_error=c4_err)
                finally:
                    c4p.close()
            c4 = c4_leg('field', 'one field per rank, rank 0 two')
            c4r = c4_leg('rows', 'the five grids cut by pupil rows, the same share for every rank')
        return {'ms_per_step': ms, 'exchange': best, 'intersections_per_step': prob.intersections,
                'one_gpu_same_problem_ms': one_ms, 'one_gpu_error': one_err, 'configs3_by_field': c4,
                'configs3_by_rows': c4r,
                'rays_per_step': sum(prob.caps), 'workload': prob.what, 'pairs': pairs,
                'stages': tm.get('stages'), 'pieces_per_rank': tm.get('pieces'),
                'ms_per_step_by_exchange': by_exchange,
                'errors': errors,
                'last_pass_phases_ms_rank0': {k: {q: v[1].get(q) for q in phases} for k, v in legs.items()},
                'grids_delivered': tm.get('grids_delivered'),
                'predicted_ms': predicted_ms(pairs * 16, PRED['c5_kernels_ms_one_gpu'] * (args.strong_num / 2048) ** 2,
                                             prob.stages_at)}
    finally:
        prob.close()


def strong_scaling(args, torch, dist, multi, world, rank, fence, ranks_seen):
    """The fixed-size problems with the path's one exchange step, every variant timed with the
    exchange inside the timed region (3 passes between fences after 1 warm-up): BASELINE
    configs[4] (44-interface lithography lens, 9 x 5 x num^2, by pupil rows), configs[3]
    (Ritchey-Chretien, 5 fields x 256^2, whole fields per rank) and configs[1] (the N = 1 main
    line's double Gauss grid, by pupil rows) -- delivered to rank 0's host memory by the
    pipelined RCCL gather (`rccl`), by every rank's copy engine into a shared pinned host
    segment (`host`), or left in rank 0's HBM (`rccl_device`); round 3's un-pipelined rccl
    form beside them."""

    def problem(name, num, by, variants, **sub):
        prob = SpotProblem(torch, dist, multi, world, rank, name, num, by, **sub)
        res = {'workload': prob.what, 'rays': sum(prob.caps), 'rays_per_rank_max': max(prob.caps),
               'pieces_per_rank': [len(p) for p in prob.pieces], 'intersections': prob.intersections}
        try:
            k = torch.tensor([prob.kernel_ms(), prob.kernel_ms()][1:], dtype=torch.float64, device=prob.eng.device)
            if multi:
                dist.all_reduce(k, op=dist.ReduceOp.MAX)
            res['kernel_ms_max_over_ranks'] = k.item()
            res['ray_surface_per_s_kernel'] = prob.intersections / (k.item() * 1e-3)
        except Exception as e:
            res['kernel_ms_max_over_ranks'] = {'error': repr(e)}
        if name == 'litho_c5' and isinstance(res.get('kernel_ms_max_over_ranks'), float):
            res['predicted_ms'] = predicted_ms(0, 0, prob.stages_at)     # (pairs filled in below)
        for key in variants:
            seg = None
            try:
                kw = {'rccl': dict(exchange='rccl'), 'host': dict(exchange='host'),
                      'rccl_device': dict(exchange='rccl', result_on='device'),
                      'rccl_unpipelined': dict(exchange='rccl', pipeline=False),
                      # the same exchange behind the tolerance-mode kernels (ROX_FAST_FP64: <= 1e-10
                      # from the reference, not bit-exact; never the headline)
                      'rccl_tolerance_mode': dict(exchange='rccl', tolerance_mode=True)}[key]
                setup_ms = 0.0
                if key == 'host':
                    t_s = time.perf_counter()
                    seg = prob.segment(key, fence)
                    setup_ms = (time.perf_counter() - t_s) * 1e3
                    kw['segment'] = seg
                ms, tm = prob.timed(fence, 3, 1, **kw)
                out = {'end_to_end_ms': ms, 'pairs': tm['pairs_total'], 'bytes_to_host': tm['pairs_total'] * 16,
                       'rays_per_s_end_to_end': sum(prob.caps) / (ms * 1e-3),
                       'intersections_per_s_end_to_end': prob.intersections / (ms * 1e-3),
                       'grids_delivered': tm.get('grids_delivered'),
                       'phases_ms_last_pass_this_rank': {k2: tm.get(k2) for k2 in
                                                         ('trace_ms', 'stage_sync_ms', 'counts_ms', 'gather_ms',
                                                          'd2h_ms', 'reassembly_ms')},
                       'stages': tm.get('stages')}
                if key == 'host':
                    out['segment_setup_ms'] = setup_ms
                    out['segment'] = {'path': seg.path, 'MiB': seg.nbytes >> 20}
                if 'predicted_ms' in res and key == 'rccl':
                    res['predicted_ms'] = predicted_ms(
                        out['bytes_to_host'], PRED['c5_kernels_ms_one_gpu'] * (num / 2048) ** 2, prob.stages_at)
                res[key] = out
            except Exception as e:
                import traceback
                traceback.print_exc(file=sys.stderr)
                res[key] = {'error': repr(e)}
            finally:
                if seg is not None:
                    LIVE_SEGMENTS.discard(seg.path)
                    seg.close(unlink=(rank == 0))
                torch.cuda.empty_cache()
        prob.close()
        return res

    return {'scaling': 'strong', 'ranks': world, 'ranks_seen_by_backend': ranks_seen,
            'backend': (dist.get_backend() if multi else 'none (single process)'),
            'what': 'end_to_end_ms = (fence, 3 passes, fence) / 3, each pass: launches -> counts -> exchange -> '
                    'rank 0 holds {(field, wvl): (R_ok, 2) array} (host memory; rccl_device: its HBM).  '
                    'Pipelined: a piece (<= 4 Mi rays) moves on while the next pieces are traced.  rccl = '
                    'grouped send/recv of the packed pairs to rank 0 + copy-engine D2H per piece; host = '
                    'copy-engine D2H of every rank over its own PCIe link into a shared pinned segment, no '
                    'xGMI step',
            'c5': problem('litho_c5', args.strong_num, 'rows', ('rccl', 'host', 'rccl_device', 'rccl_unpipelined',
                                                                   'rccl_tolerance_mode')),
            'c4': problem('rc_telescope_c4', 256, 'field', ('rccl', 'host')),
            # the N = 1 main line's lens and grid (one field, one wavelength), by pupil rows
            'c2_sharded': problem('dblgauss_c2', args.num, 'rows', ('rccl', 'rccl_device'),
                                  field_idx=[0], n_wvls=1)}


def reference_python():
    """the reference's own Python path on BASELINE configs[1], and the C port on the same
    rays on the same host, as timed by tools/time_reference.py in the build container (the
    reference cannot run on the GPU box)"""
    path = os.path.join(ROOT, 'profiles', 'reference_cpu.json')
    if not os.path.exists(path):
        return None
    with open(path) as f:
        r = json.load(f)
    out = {'measured_on': r['host'], 'workload': r['workload'],
           'driver_trace_grid_rays_per_s': r['driver_trace_grid']['rays_per_s'],
           'driver_trace_grid_intersections_per_s': r['driver_trace_grid']['intersections_per_s'],
           'extrapolated_1M_ray_spot_s': r['driver_trace_grid']['extrapolated_1M_ray_spot_s'],
           'raw_rt_trace_rays_per_s': r['raw_rt_trace']['rays_per_s'],
           'raw_rt_trace_intersections_per_s': r['raw_rt_trace']['intersections_per_s'],
           'fanned': r['raw_rt_trace_fanned'],
           'source': 'profiles/reference_cpu.json (tools/time_reference.py; mirrors the '
                     "reference's own rayoptics/raytr/tests/time_trace.py)"}
    if 'port_same_host' in r:
        out['port_same_host'] = r['port_same_host']
    if 'config1_singlet_64' in r:
        out['config1_singlet_64'] = r['config1_singlet_64']
    return out


def cpu_baseline(wl, fld, wi, opts, num, rows, fi=0, dev_sample=None):
    """the CPU beside the kernel, on this host.  When the reference itself is staged here
    (oracle/_ref, built by oracle/stage_reference.py) the headline is ITS rate -- kind
    "reference": rt.trace over a block of rows of the same grid, one core -- with the plain-C
    port (oracle/rox_oracle.c) kept beside it; without it the port is the headline (kind
    "port") and the reference's figure is the build container's, bridged by a same-host ratio."""
    port = cpu_port(wl, fld, wi, opts, num, rows)
    try:
        live = cpu_reference_live(wl, fi, wi, num, dev_sample)
    except Exception as e:                  # the bench line must not die with the baseline
        import traceback
        traceback.print_exc(file=sys.stderr)
        live = {'error': repr(e)}
    if not live or 'error' in live:
        port['reference_on_this_host'] = live
        return port
    out = {'value': live['raw_rt_trace']['intersections_per_s'],
           'unit': 'ray-surface intersections/s', 'cores': 1, 'kind': 'reference',
           'sample': live['raw_rt_trace']['sample'],
           'rays_per_s': live['raw_rt_trace']['rays_per_s'],
           'host_cpu_count': os.cpu_count(), 'host': live['host'],
           'driver_trace_grid': live['driver_trace_grid'],
           'parity_vs_timed_launch': live['parity_vs_timed_launch'],
           'reference_build': live['reference_build'],
           'port': {k: port[k] for k in ('value', 'unit', 'cores', 'kind', 'sample', 'rays_per_s',
                                         'all_cores')},
           'fanned': live.get('fanned'),
           'port_over_reference_this_host': port['value'] / live['raw_rt_trace']['intersections_per_s'],
           'reference_python_build_container': port.get('reference_python')}
    return out


def cpu_reference_live(wl, fi, wi, num, dev_sample):
    """the reference's own Python path (mjhoptics/ray-optics, sourceless byte code staged in
    oracle/_ref) timed on THIS host, one core, the shape of its own benchmark
    (rayoptics/raytr/tests/time_trace.py:37-45: repeated rt.trace) on BASELINE configs[1]:

      raw     rt.trace (raytrace.py:51-80) over `rows` x num rays of the timed num x num grid
              -- the same pupil coordinates (accumulate-by-step), ray starts made by the
              reference's own apply_vignetting / ray_start_from_osp, check_apertures=True
      driver  trace.trace_grid (trace.py:563-605) with SpotDiagramFigure's spot filter over a
              256 x 256 grid of the same field: what a user of the reference runs

    and every packet the raw loop returns is compared with the packets of the timed HIP launch
    (`dev_sample`: the same rows of the device result) -- segments, op_delta, failure kind and
    surface.  None when no reference is importable on this host."""
    from oracle import refshim
    if not refshim.available():
        return None
    import platform
    gold = os.path.join(ROOT, 'tests', 'golden')
    if gold not in sys.path:
        sys.path.insert(0, gold)
    import refmodels as rm                      # installs the import shim
    import rayoptics.raytr.raytrace as rt
    import rayoptics.raytr.trace as rtrace
    from rayoptics.raytr.traceerror import (TraceError, TraceMissedSurfaceError, TraceTIRError,
                                            TraceRayBlockedError)
    from rayoptics_amd import abi
    opm = rm.dblgauss()
    sm, osp = opm['seq_model'], opm['optical_spec']
    fld = osp['fov'].fields[fi]
    wvl = osp['wvls'].wavelengths[wi]
    N = len(sm.ifcs)
    rows = dev_sample['rows'] if dev_sample else 64
    row0 = dev_sample['row0'] if dev_sample else (num - rows) // 2
    # pupil coordinates of the grid by repeated += (trace.py:563-605)
    xs = np.empty(num)
    step = 2.0 / (num - 1)
    v = -1.0
    for k in range(num):
        xs[k] = v
        v += step
    starts = []
    for i in range(row0, row0 + rows):
        for j in range(num):
            pupil = fld.apply_vignetting(np.array([xs[i], xs[j]]))
            pt0, dir0 = osp.ray_start_from_osp(pupil, fld, 'rel pupil')
            if dir0[2] * sm.z_dir[0] < 0:
                dir0 = -dir0
            starts.append((pt0, dir0))
    R = len(starts)
    pkgs = [None] * R
    inters = 0
    t0 = time.perf_counter()
    for r, (pt0, dir0) in enumerate(starts):
        try:
            pkgs[r] = rt.trace(sm, pt0, dir0, wvl, check_apertures=True)
            inters += N - 1
        except TraceError as e:
            pkgs[r] = e
            inters += e.surf
    dt_raw = time.perf_counter() - t0

    # the driver a user calls: trace_grid with the spot filter, 256 x 256
    foc = 0.0
    rs_pkg, cr_pkg = rtrace.setup_pupil_coords(opm, fld, wvl, foc)
    fld.chief_ray, fld.ref_sphere = cr_pkg, rs_pkg
    img = rs_pkg[0]

    def spot(p, ray_pkg):
        if ray_pkg is None:
            return None
        pt = ray_pkg[0][-1][0]
        return np.array([pt[0] - img[0], pt[1] - img[1]])
    n_drv = 256
    t0 = time.perf_counter()
    g = rtrace.trace_grid(opm, [np.array([-1., -1.]), np.array([1., 1.]), n_drv], fld, wvl, foc,
                          img_filter=spot, form='list', append_if_none=False)
    dt_drv = time.perf_counter() - t0

    parity = None
    if dev_sample is not None:
        kinds = {TraceMissedSurfaceError: abi.MISSED_SURFACE, TraceTIRError: abi.TIR,
                 TraceRayBlockedError: abi.BLOCKED}
        seg, op = dev_sample['seg'], dev_sample['op']
        st, fs = dev_sample['status'], dev_sample['fail_surf']
        worst, n_bits, n_vals, bad_status = 0.0, 0, 0, 0
        for r, pk in enumerate(pkgs):
            if isinstance(pk, TraceError):
                k = next((c for t, c in kinds.items() if isinstance(pk, t)), -1)
                bad_status += int(st[r] != k or fs[r] != pk.surf)
                continue
            bad_status += int(st[r] != abi.OK)
            ref = np.array([np.concatenate([s_[0], s_[1], [s_[2]], s_[3]]) for s_ in pk[0]])
            dev = seg[:, :, r]
            same = ref == dev
            n_bits += int(same.sum())
            n_vals += ref.size + 1
            n_bits += int(pk[1] == op[r])
            err = np.abs(ref - dev) / np.maximum(1.0, np.abs(ref))
            worst = max(worst, float(err.max()), abs(pk[1] - op[r]) / max(1.0, abs(pk[1])))
        parity = {'rays': R, 'rows_of_the_timed_grid': [row0, row0 + rows],
                  'status_or_surface_mismatches': bad_status,
                  'values_compared': n_vals, 'values_bit_identical': n_bits,
                  'max_scaled_abs_diff': worst, 'tolerance_north_star': 1e-10,
                  'what': 'every packet rt.trace returned for these rays vs the same rays of the '
                          'timed HIP launch: 13 segments x 10 doubles + op_delta per surviving ray; '
                          'failure class and surface for the others'}
    try:
        fanned = cpu_reference_fanned(fi, wi, num, R / dt_raw)
    except Exception as e:                      # noqa: BLE001  (the one-core figure stands)
        import traceback
        traceback.print_exc(file=sys.stderr)
        fanned = {'error': repr(e)}
    return {
        'fanned': fanned,
        'host': {'cpu': _cpu_model(), 'nproc': os.cpu_count(), 'python': platform.python_version(),
                 'numpy': np.__version__, 'blas': _blas_info()},
        'reference_build': {'where': 'oracle/_ref (sourceless byte code, oracle/stage_reference.py)'
                            if refshim.is_staged() else refshim.REFERENCE_SRC,
                            'stamp': _ref_stamp()},
        'raw_rt_trace': {'seconds': dt_raw, 'rays': R, 'intersections': inters,
                         'rays_per_s': R / dt_raw, 'intersections_per_s': inters / dt_raw,
                         'sample': f'the reference\'s rt.trace (raytrace.py:51-80), one core, over pupil '
                                   f'rows {row0}..{row0 + rows - 1} x {num} = {R} rays of the timed '
                                   f'{num}x{num} grid (field {fi}, {wvl} nm, check_apertures), {dt_raw:.1f} s'},
        'driver_trace_grid': {'seconds': dt_drv, 'rays': n_drv * n_drv, 'rays_through': len(g),
                              'rays_per_s': n_drv * n_drv / dt_drv,
                              'extrapolated_1M_ray_spot_s': dt_drv * (1024 * 1024) / (n_drv * n_drv),
                              'what': 'trace.trace_grid (trace.py:563-605), spot filter, 256x256, one core'},
        'parity_vs_timed_launch': parity}


def _ref_grid_rows(opm, fi, wi, num, row0, rows):
    """(sm, wvl, N, starts) of pupil rows [row0, row0 + rows) of the num x num grid of field fi:
    pupil coordinates by repeated += (trace.py:563-605), ray starts by the reference's own
    apply_vignetting / ray_start_from_osp"""
    sm, osp = opm['seq_model'], opm['optical_spec']
    fld = osp['fov'].fields[fi]
    wvl = osp['wvls'].wavelengths[wi]
    xs = np.empty(num)
    step = 2.0 / (num - 1)
    v = -1.0
    for k in range(num):
        xs[k] = v
        v += step
    starts = []
    for i in range(row0, row0 + rows):
        for j in range(num):
            pupil = fld.apply_vignetting(np.array([xs[i], xs[j]]))
            pt0, dir0 = osp.ray_start_from_osp(pupil, fld, 'rel pupil')
            if dir0[2] * sm.z_dir[0] < 0:
                dir0 = -dir0
            starts.append((pt0, dir0))
    return sm, wvl, len(sm.ifcs), starts


def ref_worker(spec_json):
    """one process of cpu_reference_fanned(): rt.trace over its own pupil rows of the timed grid,
    started at a common wall-clock instant; prints one JSON line"""
    spec = json.loads(spec_json)
    gold = os.path.join(ROOT, 'tests', 'golden')
    if gold not in sys.path:
        sys.path.insert(0, gold)
    import refmodels as rm                      # installs the import shim
    import rayoptics.raytr.raytrace as rt
    from rayoptics.raytr.traceerror import TraceError
    opm = rm.dblgauss()
    sm, wvl, N, starts = _ref_grid_rows(opm, spec['fi'], spec['wi'], spec['num'], spec['row0'], spec['rows'])
    late = time.time() - spec['t_start']
    while time.time() < spec['t_start']:
        time.sleep(0.002)
    inters = 0
    t0 = time.time()
    for pt0, dir0 in starts:
        try:
            rt.trace(sm, pt0, dir0, wvl, check_apertures=True)
            inters += N - 1
        except TraceError as e:
            inters += e.surf
    t1 = time.time()
    print(json.dumps({'rays': len(starts), 'intersections': inters, 't_begin': t0, 't_end': t1,
                      'late_s': max(late, 0.0)}), flush=True)
    return 0


def cpu_reference_fanned(fi, wi, num, rays_per_s_one_core, budget_s=None):
    """the reference over the cores of THIS host -- rt.trace fanned over
    min(os.cpu_count(), 64) processes (the shape of rayoptics/raytr/tests/time_trace.py:37-45, one
    Python process per core: the reference has no threading of its own), each on its own pupil
    rows of the timed grid, all starting at one wall-clock instant; rate = everything traced /
    (last finish - common start).  Bounded: ~budget_s seconds of tracing per process."""
    import subprocess
    if budget_s is None:
        budget_s = float(os.environ.get('ROX_REF_FAN_SECONDS', '20'))
    procs = max(1, min(os.cpu_count() or 1, 64))
    rows_total = int(min(num, max(procs, budget_s * rays_per_s_one_core * procs / num)))
    procs = min(procs, rows_total)
    bounds = [(rows_total * k) // procs for k in range(procs + 1)]
    row_base = (num - rows_total) // 2
    # ray starts are made before the common start: the slowest start-up (import + model + the
    # Python ray-start loop, ~0.2 ms per ray) decides how far ahead the instant must lie
    per = max(b1 - b0 for b0, b1 in zip(bounds, bounds[1:])) * num
    t_start = time.time() + 12.0 + per / 2500.0
    env = dict(os.environ, OMP_NUM_THREADS='1', OPENBLAS_NUM_THREADS='1', MKL_NUM_THREADS='1')
    children = []
    for k in range(procs):
        spec = {'fi': fi, 'wi': wi, 'num': num, 'row0': row_base + bounds[k], 'rows': bounds[k + 1] - bounds[k],
                't_start': t_start}
        children.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), '--ref-worker',
                                          json.dumps(spec)], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL,
                                         env=env, text=True))
    recs = []
    for c in children:
        out, _ = c.communicate(timeout=600)
        line = [ln for ln in out.splitlines() if ln.startswith('{')]
        if c.returncode != 0 or not line:
            raise RuntimeError(f'a reference worker failed (rc {c.returncode})')
        recs.append(json.loads(line[-1]))
    rays = sum(r['rays'] for r in recs)
    inters = sum(r['intersections'] for r in recs)
    t_begin = min(r['t_begin'] for r in recs)
    t_end = max(r['t_end'] for r in recs)
    wall = t_end - min(t_begin, t_start)
    return {'processes': procs, 'rays': rays, 'intersections': inters, 'seconds': wall,
            'rays_per_s': rays / wall, 'intersections_per_s': inters / wall,
            'per_process_seconds_min_max': [min(r['t_end'] - r['t_begin'] for r in recs),
                                            max(r['t_end'] - r['t_begin'] for r in recs)],
            'started_late_s_max': max(r['late_s'] for r in recs),
            'host': {'cpu': _cpu_model(), 'nproc': os.cpu_count()},
            'speedup_over_one_core': (rays / wall) / rays_per_s_one_core,
            'how': f'the reference\'s rt.trace in {procs} Python processes (one per core, OMP/BLAS threads = 1), '
                   f'pupil rows {row_base}..{row_base + rows_total - 1} x {num} of the timed grid split between '
                   'them, common start; rate = all rays / (last finish - start)'}


def _cpu_model():
    try:
        with open('/proc/cpuinfo') as f:
            for ln in f:
                if ln.startswith('model name'):
                    return ln.split(':', 1)[1].strip()
    except OSError:
        pass
    return 'unknown'


def _blas_info():
    try:
        b = np.show_config(mode='dicts').get('Build Dependencies', {}).get('blas', {})
        return f"{b.get('name')} {b.get('version')}"
    except Exception:
        return 'unknown'


def _ref_stamp():
    try:
        with open(os.path.join(ROOT, 'oracle', '_ref', 'stamp.json')) as f:
            st = json.load(f)
        return {k: st.get(k) for k in ('source_hash', 'python', 'modules')}
    except (OSError, ValueError):
        return None


def cpu_port(wl, fld, wi, opts, num, rows):
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
    ref = reference_python()
    out = {'value': inters / dt, 'unit': 'ray-surface intersections/s', 'cores': 1,
           'kind': 'port',
           'sample': f'{passes} passes over {rows} pupil rows x {num} = {rows * num} rays of the '
                     f'same grid, FULL packets, oracle/rox_oracle.c -O2 single thread, {dt:.1f} s',
           'rays_per_s': passes * rows * num / dt,
           'host_cpu_count': os.cpu_count(), 'all_cores': allc,
           'reference_python': ref}
    if ref and ref.get('port_same_host'):
        # the bridge between the two hosts: port / reference on ONE host (build container),
        # and from it the reference's rate this host would show if it could run here
        ratio = ref['port_same_host']['port_over_reference']
        out['port_over_reference_same_host'] = ratio
        out['reference_estimated_on_this_host'] = {
            'value': inters / dt / ratio, 'unit': 'ray-surface intersections/s',
            'how': 'this host\'s single-thread port rate / the build container\'s port:reference ratio'}
    return out


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
