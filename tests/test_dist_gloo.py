"""N > 1 path on CPU: gloo process groups of 2 and 3 ranks -- partition, per-rank
packed hits (ROX_OUT_HITS_COMPACT | ROX_HITS_APPEND), the count exchange, the
variable-size gather to rank 0 and the shared-host-segment alternative.  The
per-rank trace is served by the oracle-backed test double
(tests/oracle_engine.py) since there is no GPU here; on the GPU box the same
code runs over RCCL with the HIP engine (bench.py --gpus N)."""
import os
import sys

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_partition_covers_everything_once():
    from rayoptics_amd.dist import partition
    for by in ('rows', 'field'):
        for nf, nw, num, world in [(3, 3, 16, 2), (5, 1, 8, 4), (9, 5, 7, 8), (1, 1, 5, 8), (2, 2, 4, 3)]:
            plan = partition(nf, nw, num, world, by)
            assert len(plan) == world
            seen = np.zeros((nf, nw, num), dtype=int)
            order = []
            for blocks in plan:
                for b in blocks:
                    assert 0 < b.row_count and b.row_begin + b.row_count <= num
                    seen[b.fi, b.wi, b.row_begin:b.row_begin + b.row_count] += 1
                    order.append(((b.fi * nw + b.wi) * num + b.row_begin))
            assert (seen == 1).all()
            assert order == sorted(order)           # ranks own contiguous runs of the global order
            rows = [sum(b.row_count for b in blocks) for blocks in plan]
            if by == 'rows':
                assert max(rows) - min(rows) <= 1
            else:
                per_field = [len({b.fi for b in blocks}) for blocks in plan]
                assert max(per_field) - min(per_field) <= 1 or nf < world
                for blocks in plan:                 # whole fields only
                    assert all(b.row_begin == 0 and b.row_count == num for b in blocks)


def test_shard_by_field_of_baseline_config_3():
    """5 fields over 4 ranks: 2 / 1 / 1 / 1"""
    from rayoptics_amd.dist import partition
    plan = partition(5, 1, 256, 4, by='field')
    assert [sorted({b.fi for b in blocks}) for blocks in plan] == [[0, 1], [2], [3], [4]]


def test_spot_views_are_slices_of_one_buffer():
    from rayoptics_amd.dist import partition, spot_views
    plan = partition(2, 2, 4, 3)
    rng = np.random.default_rng(5)
    counts = [rng.integers(0, 9, size=len(b)) for b in plan]
    total = int(sum(c.sum() for c in counts))
    buf = np.arange(2 * total, dtype=float).reshape(-1, 2)
    v = spot_views(buf, plan, counts)
    assert sum(len(a) for a in v.values()) == total
    at = 0
    for key in sorted(v):
        assert v[key].base is not None and np.shares_memory(v[key], buf)
        np.testing.assert_array_equal(v[key], buf[at:at + len(v[key])])
        at += len(v[key])
    # rank slices with gaps (the shared host segment): cut grids are concatenated
    caps = [int(c.sum()) + 3 for c in counts]
    offs = np.concatenate([[0], np.cumsum(caps)])
    gap = np.full((offs[-1], 2), -1.0)
    at = 0
    for k, c in enumerate(counts):
        n = int(c.sum())
        gap[offs[k]:offs[k] + n] = buf[at:at + n]
        at += n
    v2 = spot_views(gap, plan, counts, offs)
    for key in v:
        np.testing.assert_array_equal(v2[key], v[key])


def _worker(rank, world, port, q, by, exchange, num, name):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import rayoptics_amd  # noqa: F401
    from rayoptics_amd import workloads
    from rayoptics_amd import dist as rdist
    from oracle_engine import OracleEngine
    wl = workloads.load(name)
    eng = OracleEngine(wl.table)
    nw = len(wl.table.wvls)
    seg = None
    if exchange == 'host':
        plan = rdist.partition(len(wl.fields), nw, num, world, by)
        caps = [rdist.rays_of(b, num) for b in plan]
        if rank == 0:
            seg = rdist.HostSegment(eng, f'rox_test_{port}', caps, rank, create=True, dir='/tmp')
        dist.barrier()
        if rank != 0:
            seg = rdist.HostSegment(eng, f'rox_test_{port}', caps, rank, create=False, dir='/tmp')
    tm = {}
    out = rdist.trace_spot_sharded(eng, wl.fields, wl.image_pts, nw, num, wl.foc, by=by,
                                   exchange=exchange, segment=seg, timings=tm)
    q.put((rank, None if out is None else {k: v.copy() for k, v in out.items()}, tm))
    dist.barrier()
    if seg is not None:
        seg.close(unlink=(rank == 0))
    dist.destroy_process_group()


def _run(world, by, exchange, num, name, salt):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29500 + (os.getpid() * 7 + salt) % 2000
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, by, exchange, num, name))
             for r in range(world)]
    for p in procs:
        p.start()
    got = {}
    for _ in procs:
        r, out, tm = q.get(timeout=180)
        got[r] = (out, tm)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return got


@pytest.mark.parametrize('world,by,exchange,name', [
    (2, 'rows', 'rccl', 'dblgauss_c2'),
    (3, 'rows', 'rccl', 'dblgauss_c2'),
    (2, 'rows', 'host', 'dblgauss_c2'),
    (4, 'field', 'rccl', 'rc_telescope_c4'),
])
def test_sharded_spot_matches_single_process(world, by, exchange, name):
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import rayoptics_amd  # noqa: F401
    from rayoptics_amd import workloads, abi
    from oracle import oracle
    num = 12
    got = _run(world, by, exchange, num, name, salt=world * 10 + len(by) + len(exchange))
    assert got[0][0] is not None
    for r in range(1, world):
        assert got[r][0] is None
    res, tm = got[0]
    wl = workloads.load(name)
    N = wl.n_ifcs
    nw = len(wl.table.wvls)
    assert len(res) == len(wl.fields) * nw
    total = 0
    blocked = 0
    for (fi, wi), xy in res.items():
        # single-process truth: the whole grid through the oracle, survivors in ray order
        opts = oracle.make_opts(flags=abi.INTERSECT_OBJ | abi.CHECK_APERTURES | abi.APPLY_VIGNETTING,
                                out_mode=abi.OUT_HITS_COMPACT, first_surf=1, last_surf=N - 2,
                                foc=wl.foc, image_pt=wl.image_pts[fi])
        ref = oracle.trace_pupil_grid(wl.table, wl.fields[fi], oracle.make_grid((-1., -1.), (1., 1.), num),
                                      wi, opts)
        np.testing.assert_array_equal(xy, ref.hits)
        total += len(xy)
        blocked += num * num - len(xy)
    assert blocked > 0 and total > 0            # a variable-size exchange: no padding went round
    assert tm['pairs_total'] == total and len(tm['pairs_per_rank']) == world
    for k in ('trace_ms', 'counts_ms', 'gather_ms', 'd2h_ms', 'reassembly_ms'):
        assert k in tm
