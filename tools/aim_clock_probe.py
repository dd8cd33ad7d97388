import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import rayoptics_amd
from rayoptics_amd import abi, workloads
from rayoptics_amd.engine import TraceEngine, make_opts, make_grid
for name in ('dblgauss_c2', 'cell_phone'):
    wl = workloads.load(name)
    eng = TraceEngine(wl.table)
    probs = []
    for m in wl.aim:
        a = abi.Aim()
        for i in range(3): a.pt0[i] = m['pt0'][i]
        a.z_enp, a.y_target, a.z_dir0 = m['z_enp'], 0.0, m['z_dir0']
        a.wvl_idx, a.surf, a.flip = m['wvl_idx'], m['surf'], 1
        probs.append(a)
    for m in wl.aim2d or []:
        a = abi.Aim()
        for i in range(3): a.pt0[i] = m['pt0'][i]
        a.z_enp, a.y_target, a.z_dir0 = m['z_enp'], 0.0, m['z_dir0']
        a.wvl_idx, a.surf, a.flip = m['wvl_idx'], m['surf'], 1
        a.two_d, a.epsfcn = 1, m['epsfcn']
        probs.append(a)
    def timed(pre):
        ts, ws = [], []
        for _ in range(30):
            if pre:
                x = torch.randn(8192, 8192, device='cuda')
                for _ in range(10):
                    y = x @ x
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            t0 = time.perf_counter()
            eng.aim_chief_rays(probs)
            w = time.perf_counter() - t0
            e1.record(); e1.synchronize()
            ts.append(e0.elapsed_time(e1)); ws.append(w * 1e3)
        return float(np.median(ts)), float(np.median(ws))
    for _ in range(5): eng.aim_chief_rays(probs)
    print(json.dumps({'workload': name, 'idle_gpu_events_ms_wall_ms': timed(False), 'behind_a_gemm_burst': timed(True)}))
