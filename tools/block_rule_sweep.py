#!/usr/bin/env python3
"""Workgroup size of a pupil launch vs launch size (roxtrace.hip want_small()).

    ROX_SMALL_BLOCKS=0|1 python tools/block_rule_sweep.py [--shapes ...] >> profiles/r05_block_rule.jsonl

One JSON line per shape: steady-state ms per pass (HIP events on the launch stream) of the
batched FULL and HITS launches of `fields` grids of num x num rays.  The environment variable is
read once per process, so the two forms are two runs; `auto` (unset) is what ships."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

# (workload, field indices, num)
SHAPES = {
    'c1': ('singlet_c1', [0], 64),
    'dg64': ('dblgauss_c2', [0], 64),
    'dg3x64': ('dblgauss_c2', [0, 1, 2], 64),
    'dg128': ('dblgauss_c2', [0], 128),
    'dg256': ('dblgauss_c2', [0], 256),
    'dg3x256': ('dblgauss_c2', [0, 1, 2], 256),
    'dg512': ('dblgauss_c2', [0], 512),
    'dg640': ('dblgauss_c2', [0], 640),
    'dg724': ('dblgauss_c2', [0], 724),
    'dg1024': ('dblgauss_c2', [0], 1024),
    'c4': ('rc_telescope_c4', [0, 1, 2, 3, 4], 256),
    'zmx512': ('zmx_evenasph_c3', [0], 512),
    'zmx1024': ('zmx_evenasph_c3', [0], 1024),
    'nik512': ('nikkor_c3', [0], 512),
    'phone512': ('cell_phone', [0], 512),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--shapes', default='c1,dg64,dg3x64,dg128,dg256,dg3x256,dg512,dg640,dg724,dg1024,c4')
    args = ap.parse_args()
    import torch
    import rayoptics_amd  # noqa: F401
    from rayoptics_amd import abi, workloads, engine
    from rayoptics_amd.engine import TraceEngine, make_opts, make_grid, DeviceResult
    flags0 = abi.CHECK_APERTURES | abi.APPLY_VIGNETTING
    for key in args.shapes.split(','):
        name, fis, num = SHAPES[key]
        wl = workloads.load(name)
        N = wl.n_ifcs
        eng = TraceEngine(wl.table)
        R = num * num
        grid = make_grid((-1., -1.), (1., 1.), num)
        rec = {'shape': key, 'workload': name, 'grids': len(fis), 'num': num, 'rays': R * len(fis),
               'waves': (R + 63) // 64 * len(fis),
               'small_blocks_env': os.environ.get('ROX_SMALL_BLOCKS', 'auto'),
               'small_waves_per_cu_env': os.environ.get('ROX_SMALL_WAVES_PER_CU', ''),
               'lib': os.path.basename(engine.LIB_PATH)}
        for mode, label in ((abi.OUT_FULL, 'full'), (abi.OUT_HITS, 'hits')):
            fl = [wl.fields[f] for f in fis]
            ress = [DeviceResult(torch, eng.device, eng.num_segments(0), R, mode,
                                 want_pupil=(mode == abi.OUT_FULL), nan_fill=False) for _ in fis]
            optl = [make_opts(flags=flags0 | (abi.INTERSECT_OBJ if (wl.fields[f].kind != abi.FLD_EPD_WIDE
                                                                    and wl.fields[f].z_dir0 != 0.0) else 0),
                              out_mode=mode, first_surf=1, last_surf=N - 2, foc=wl.foc,
                              image_pt=wl.image_pts[f]) for f in fis]
            wv = [wl.ref_wvl_idx] * len(fis)

            def one_pass():
                if len(fis) == 1:
                    eng.trace_pupil_grid(fl[0], grid, wv[0], optl[0], out=ress[0])
                else:
                    eng.trace_pupil_grids(fl, wv, grid, optl, outs=ress)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            n_warm = 0
            while (time.perf_counter() - t0) < 0.15 or n_warm < 2:
                one_pass()
                n_warm += 1
                if n_warm % 16 == 0:
                    torch.cuda.synchronize()
            torch.cuda.synchronize()
            per = max(1, min(200, int(0.02 / max((time.perf_counter() - t0) / n_warm, 1e-6))))
            ts = []
            for _ in range(7):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(per):
                    one_pass()
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) / per)
            rec[label + '_us'] = round(sorted(ts)[len(ts) // 2] * 1e3, 2)
            rec[label + '_min_us'] = round(min(ts) * 1e3, 2)
            del ress
        print(json.dumps(rec), flush=True)
        eng.close()
        torch.cuda.empty_cache()


if __name__ == '__main__':
    main()
