#!/usr/bin/env python3
"""Sustained-launch behaviour of the trace kernels: per-batch mean duration of
back-to-back launches over a few seconds, with clocks / power sampled from
rocm-smi meanwhile (is the FULL kernel power-capped when launched continuously?).

    [ROX_LIB=variant.so] python tools/sustained_probe.py [--mode full|hits|hits_fast|full_fast] [--seconds 3]"""
import argparse
import json
import os
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def smi():
    try:
        out = subprocess.run(['rocm-smi', '--showclocks', '--showpower', '--showtemp', '--json'],
                             capture_output=True, text=True, timeout=10).stdout
        j = json.loads(out)
        c = j.get('card0', {})
        keep = {k: v for k, v in c.items() if any(t in k.lower() for t in ('sclk', 'mclk', 'power', 'junction', 'hbm', 'fclk'))}
        return keep
    except Exception as e:
        return {'error': repr(e)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--mode', default='full')
    ap.add_argument('--seconds', type=float, default=3.0)
    ap.add_argument('--batch', type=int, default=10)
    ap.add_argument('--workload', default='dblgauss_c2')
    ap.add_argument('--field', type=int, default=0)
    ap.add_argument('--num', type=int, default=1024)
    ap.add_argument('--uncached', action='store_true',
                    help='packet buffer from hipExtMallocWithFlags(hipDeviceMallocUncached) instead of torch')
    args = ap.parse_args()
    import torch
    import rayoptics_amd  # noqa: F401
    from rayoptics_amd import abi, workloads, engine
    from rayoptics_amd.engine import TraceEngine, make_opts, make_grid, DeviceResult
    wl = workloads.load(args.workload)
    N = wl.n_ifcs
    eng = TraceEngine(wl.table)
    fld = wl.fields[args.field]
    flags = abi.INTERSECT_OBJ | abi.CHECK_APERTURES | abi.APPLY_VIGNETTING
    if fld.kind == abi.FLD_EPD_WIDE or fld.z_dir0 == 0.0:
        flags &= ~abi.INTERSECT_OBJ
    mode = abi.OUT_FULL if args.mode in ('full', 'full_fast') else abi.OUT_HITS
    if args.mode in ('hits_fast', 'full_fast'):           # the tolerance-mode twin (ROX_FAST_FP64)
        flags |= abi.FAST_FP64
    o = make_opts(flags=flags, out_mode=mode, first_surf=1, last_surf=N - 2, foc=wl.foc,
                  image_pt=wl.image_pts[args.field])
    R = args.num * args.num
    grid = make_grid((-1., -1.), (1., 1.), args.num)
    out = DeviceResult(torch, eng.device, eng.num_segments(flags), R, mode,
                       want_pupil=(mode == abi.OUT_FULL), nan_fill=False)
    if args.uncached:
        import ctypes as C
        hip = C.CDLL('libamdhip64.so')
        ptr = C.c_void_p()
        nbytes = out._seg.numel() * 8
        rc = hip.hipExtMallocWithFlags(C.byref(ptr), C.c_size_t(nbytes), C.c_uint(3))   # hipDeviceMallocUncached
        if rc != 0:
            raise SystemExit(f'hipExtMallocWithFlags: {rc}')
        inner = out.out_struct

        def patched():
            o_ = inner()
            o_.seg = ptr.value
            return o_
        out.out_struct = patched
    samples = []
    stop = threading.Event()

    def sampler():
        while not stop.is_set():
            samples.append((time.perf_counter(), smi()))
            time.sleep(0.25)
    th = threading.Thread(target=sampler)
    idle = smi()
    th.start()
    time.sleep(0.6)
    series = []
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < args.seconds:
        ms = eng.time_pupil_grid(fld, grid, wl.ref_wvl_idx, o, out, args.batch)
        series.append((round(time.perf_counter() - t0, 3), round(ms * 1e3, 1)))
    t_end = time.perf_counter()
    time.sleep(0.6)
    stop.set()
    th.join()
    n = len(series)
    print(json.dumps({'lib': os.path.basename(engine.LIB_PATH), 'mode': args.mode, 'uncached': args.uncached,
                      'workload': args.workload, 'field': args.field, 'num': args.num,
                      'first_batches_us': [s[1] for s in series[:12]],
                      'every_50th_batch_us': [s[1] for s in series[::max(n // 40, 1)]],
                      'mean_us': sum(s[1] for s in series) / n,
                      'last_quarter_mean_us': sum(s[1] for s in series[3 * n // 4:]) / (n - 3 * n // 4),
                      'idle': idle,
                      'during': [s[1] for s in samples if t0 < s[0] < t_end][:8]}))


if __name__ == '__main__':
    main()
