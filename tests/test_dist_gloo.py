"""N > 1 path on CPU: world_size-2 gloo process group, row-block partition +
gather of the hits.  The per-rank trace is served by the oracle-backed test
double (tests/oracle_engine.py) since there is no GPU here; on the GPU box the
same code runs over RCCL with the HIP engine (bench.py --gpus N)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_partition_covers_everything_once():
    from rayoptics_amd.dist import partition
    for nf, nw, num, world in [(3, 3, 16, 2), (5, 1, 8, 4), (9, 5, 7, 8), (1, 1, 5, 8), (2, 2, 4, 3)]:
        plan = partition(nf, nw, num, world)
        seen = np.zeros((nf, nw, num), dtype=int)
        for blocks in plan:
            for b in blocks:
                assert 0 < b.row_count and b.row_begin + b.row_count <= num
                seen[b.fi, b.wi, b.row_begin:b.row_begin + b.row_count] += 1
        assert (seen == 1).all()
        rows = [sum(b.row_count for b in blocks) for blocks in plan]
        assert max(rows) - min(rows) <= 1


def _worker(rank, world, port, q, all_ranks):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import rayoptics_amd  # noqa: F401
    from rayoptics_amd import workloads
    from rayoptics_amd.dist import trace_spot_sharded
    from oracle_engine import OracleEngine
    wl = workloads.load('dblgauss_c2')
    eng = OracleEngine(wl.table)
    eng.device = 'cpu'
    out = trace_spot_sharded(eng, wl.fields, wl.image_pts, len(wl.table.wvls), 12,
                             wl.foc, all_ranks=all_ranks)
    if out is not None:
        q.put((rank, {k: (v[0].copy(), v[1].copy()) for k, v in out.items()}))
    else:
        q.put((rank, None))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('all_ranks', [False, True])
def test_sharded_spot_matches_single_process(all_ranks):
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import rayoptics_amd  # noqa: F401
    from rayoptics_amd import workloads, abi
    from oracle import oracle
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + (1 if all_ranks else 0)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, all_ranks)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got[0] is not None
    assert (got[1] is not None) == all_ranks
    # single-process truth: whole grids through the oracle
    wl = workloads.load('dblgauss_c2')
    N = wl.n_ifcs
    for (fi, wi), (xy, st) in got[0].items():
        opts = oracle.make_opts(flags=abi.INTERSECT_OBJ | abi.CHECK_APERTURES | abi.APPLY_VIGNETTING,
                                out_mode=abi.OUT_HITS, first_surf=1, last_surf=N - 2,
                                foc=wl.foc, image_pt=wl.image_pts[fi])
        ref = oracle.trace_pupil_grid(wl.table, wl.fields[fi], oracle.make_grid((-1., -1.), (1., 1.), 12),
                                      wi, opts)
        np.testing.assert_array_equal(st, ref.status)
        np.testing.assert_array_equal(xy, ref.seg.T)
    assert len(got[0]) == 9
    if all_ranks:
        for k in got[0]:
            np.testing.assert_array_equal(got[0][k][0], got[1][k][0])
