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


def test_pipeline_schedule_orders_shared_grids_first_and_last():
    """dist.schedule: pieces tile every rank's rows exactly once, regions do not overlap, and a
    rank traces the head of the grid it shares with the next rank first, the tail of the grid it
    shares with the previous rank last (BASELINE configs[4] over 8 ranks)"""
    from rayoptics_amd.dist import partition, schedule
    nf, nw, num, world = 9, 5, 2048, 8
    plan = partition(nf, nw, num, world)
    pieces, order = schedule(plan, num, nw)
    seen = np.zeros((nf * nw, num), dtype=int)
    for r, lst in enumerate(pieces):
        assert sorted(order[r]) == list(range(len(lst)))
        roff = 0
        for p in lst:
            assert p.rank == r and p.roff == roff and p.row_count * num <= (1 << 22)
            roff += p.row_count * num
            seen[p.g, p.row_begin:p.row_begin + p.row_count] += 1
        assert roff == sum(b.row_count for b in plan[r]) * num
        first, last = lst[order[r][0]], lst[order[r][-1]]
        if r < world - 1:       # shares its last grid with the next rank
            assert first.g == lst[-1].g and lst[-1].row_begin + lst[-1].row_count < num
        if r > 0:               # shares its first grid with the previous rank
            assert last.g == lst[0].g and lst[0].row_begin > 0
    assert (seen == 1).all()
    # the piece every rank traces last was cut into a half and two quarters: only a quarter's
    # pack / copy / send is left behind the last kernel
    flat, flat_order = schedule(plan, num, nw, taper=False)
    n_cut = 0
    for r, lst in enumerate(pieces):
        was = flat[r][flat_order[r][-1]]
        cut = was.row_count * num >= (1 << 20)
        n_cut += cut
        assert len(lst) == len(flat[r]) + (2 if cut else 0)
        last = lst[order[r][-1]]
        assert last.row_count * num <= (1 << 20) + num
    assert n_cut >= 2
    # smaller pieces: the same cover
    pieces2, _ = schedule(plan, num, nw, max_rays=300 * num)
    assert sum(len(p) for p in pieces2) > sum(len(p) for p in pieces)
    assert sum(p.row_count for lst in pieces2 for p in lst) == nf * nw * num


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


def _worker(rank, world, port, q, by, exchange, num, name, pipeline=True, piece_rays=None, taper_min=None,
            subgroup=None):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import rayoptics_amd  # noqa: F401
    from rayoptics_amd import workloads
    from rayoptics_amd import dist as rdist
    from oracle_engine import OracleEngine
    if taper_min is not None:           # (the cut of a rank's last piece starts at 2^20 rays)
        rdist.TAPER_MIN_RAYS = taper_min
    wl = workloads.load(name)
    eng = OracleEngine(wl.table)
    nw = len(wl.table.wvls)
    seg = None
    group = None
    if subgroup is not None:
        # a real sub-group whose members are NOT global ranks 0..n-1 (every process creates it)
        group = dist.new_group(ranks=list(subgroup))
        if rank not in subgroup:
            q.put((rank, 'outside', {}))
            dist.barrier()
            dist.destroy_process_group()
            return
        tm = {}
        out = rdist.trace_spot_sharded(eng, wl.fields, wl.image_pts, nw, num, wl.foc, by=by,
                                       exchange=exchange, timings=tm, pipeline=pipeline,
                                       max_piece_rays=piece_rays, group=group)
        q.put((rank, None if out is None else {k: v.copy() for k, v in out.items()}, tm))
        dist.barrier()
        dist.destroy_process_group()
        return
    if exchange == 'host':
        plan = rdist.partition(len(wl.fields), nw, num, world, by)
        # pipelined: one region per (field, wavelength) grid; round 3's form: one slice per rank
        caps = [num * num] * (len(wl.fields) * nw) if pipeline else [rdist.rays_of(b, num) for b in plan]
        if rank == 0:
            seg = rdist.HostSegment(eng, f'rox_test_{port}', caps, rank, create=True, dir='/tmp')
        dist.barrier()
        if rank != 0:
            seg = rdist.HostSegment(eng, f'rox_test_{port}', caps, rank, create=False, dir='/tmp')
    tm = {}
    out = rdist.trace_spot_sharded(eng, wl.fields, wl.image_pts, nw, num, wl.foc, by=by,
                                   exchange=exchange, segment=seg, timings=tm, pipeline=pipeline,
                                   max_piece_rays=piece_rays)
    q.put((rank, None if out is None else {k: v.copy() for k, v in out.items()}, tm))
    dist.barrier()
    if seg is not None:
        seg.close(unlink=(rank == 0))
    dist.destroy_process_group()


def _run(world, by, exchange, num, name, salt, pipeline=True, piece_rays=None, taper_min=None, subgroup=None):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29500 + (os.getpid() * 7 + salt) % 2000
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, by, exchange, num, name, pipeline,
                                               piece_rays, taper_min, subgroup))
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


@pytest.mark.parametrize('world,by,exchange,name,pipeline,piece_rays', [
    # pipelined (the default): pieces move on while later pieces are traced; small pieces so
    # that every rank runs several stages and grids are cut between pieces and between ranks
    # (piece_rays < 0: also with the last piece of every rank cut into a half and two quarters)
    (2, 'rows', 'rccl', 'dblgauss_c2', True, -96),
    (3, 'rows', 'host', 'dblgauss_c2', True, -96),
    (2, 'rows', 'rccl', 'dblgauss_c2', True, 36),
    (3, 'rows', 'rccl', 'dblgauss_c2', True, 60),
    (2, 'rows', 'host', 'dblgauss_c2', True, 36),
    (3, 'rows', 'host', 'dblgauss_c2', True, None),
    (4, 'field', 'rccl', 'rc_telescope_c4', True, 48),
    (4, 'field', 'host', 'rc_telescope_c4', True, None),
    # a grid held by three ranks (the middle rank's only grid is shared on both sides: its
    # pieces wait for the previous rank's counts and the next rank's wait for its own)
    (5, 'rows', 'rccl', 'singlet_c1', True, 24),
    (5, 'rows', 'host', 'singlet_c1', True, 24),
    # round 3's trace-everything-then-exchange form
    (2, 'rows', 'rccl', 'dblgauss_c2', False, None),
    (2, 'rows', 'host', 'dblgauss_c2', False, None),
    (4, 'field', 'rccl', 'rc_telescope_c4', False, None),
])
def test_sharded_spot_matches_single_process(world, by, exchange, name, pipeline, piece_rays):
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import rayoptics_amd  # noqa: F401
    from rayoptics_amd import workloads, abi
    from oracle import oracle
    num = 12
    taper_min = None
    if piece_rays is not None and piece_rays < 0:
        piece_rays, taper_min = -piece_rays, 48
    got = _run(world, by, exchange, num, name,
               salt=world * 10 + len(by) + len(exchange) + 100 * pipeline + (piece_rays or 0) + (taper_min or 0),
               pipeline=pipeline, piece_rays=piece_rays, taper_min=taper_min)
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
    if pipeline:
        assert tm['pipelined'] and tm['stages'] == max(tm['pieces']) and len(tm['pieces']) == world
        if piece_rays:
            assert tm['stages'] > 2


@pytest.mark.parametrize('pipeline', [True, False])
def test_sharded_spot_on_a_sub_group(pipeline):
    """the exchange names its peers by group-local rank; torch's point-to-point operations take
    GLOBAL ranks.  On the sub-group {3, 1, 2} of a 4-process world (group rank 0 = global 3) the
    pairs must still reach the group's root, and the process outside the group is not involved"""
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import rayoptics_amd  # noqa: F401
    from rayoptics_amd import workloads, abi
    from oracle import oracle
    num, name = 12, 'dblgauss_c2'
    got = _run(4, 'rows', 'rccl', num, name, salt=777 + pipeline, pipeline=pipeline,
               piece_rays=36 if pipeline else None, subgroup=(3, 1, 2))
    assert got[0][0] == 'outside'
    roots = [r for r in (1, 2, 3) if got[r][0] is not None]
    assert roots == [1] or roots == [3] or roots == [2]
    # torch orders a new group's ranks by global rank: group rank 0 is global rank 1
    res, tm = got[roots[0]]
    wl = workloads.load(name)
    N = wl.n_ifcs
    assert len(res) == len(wl.fields) * len(wl.table.wvls)
    for (fi, wi), xy in res.items():
        opts = oracle.make_opts(flags=abi.INTERSECT_OBJ | abi.CHECK_APERTURES | abi.APPLY_VIGNETTING,
                                out_mode=abi.OUT_HITS_COMPACT, first_surf=1, last_surf=N - 2,
                                foc=wl.foc, image_pt=wl.image_pts[fi])
        ref = oracle.trace_pupil_grid(wl.table, wl.fields[fi], oracle.make_grid((-1., -1.), (1., 1.), num),
                                      wi, opts)
        np.testing.assert_array_equal(xy, ref.hits)
    assert len(tm['pairs_per_rank']) == 3


def _packets_worker(rank, world, port, q, num, name, rays, dst):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import rayoptics_amd  # noqa: F401
    from rayoptics_amd import workloads, abi
    from rayoptics_amd import dist as rdist
    from rayoptics_amd.engine import make_opts
    from oracle_engine import OracleEngine
    wl = workloads.load(name)
    eng = OracleEngine(wl.table)
    opts = make_opts(flags=abi.INTERSECT_OBJ | abi.CHECK_APERTURES | abi.APPLY_VIGNETTING,
                     out_mode=abi.OUT_FULL, first_surf=1, last_surf=wl.n_ifcs - 2)
    sp = rdist.trace_packets_sharded(eng, wl.fields[-1], 0, num, opts)
    held = 0 if sp.local is None else int(sp.local.status.shape[0])
    got = sp.fetch(rays, dst=dst)
    none_again = sp.fetch([], dst=dst)          # an empty request is a valid collective too
    q.put((rank, held, got, none_again is None or len(none_again['op']) == 0))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('world,dst', [(2, 0), (3, 1), (5, 0)])
def test_full_packets_stay_on_their_rank_and_are_fetched_lazily(world, dst):
    """FULL packets are never exchanged wholesale -- a grid cut by pupil rows keeps
    each rank's packets where they were traced; `ShardedPackets.fetch` moves the requested rays
    only (grouped point-to-point to one rank), and they equal the single-process trace"""
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import rayoptics_amd  # noqa: F401
    from rayoptics_amd import workloads, abi
    from rayoptics_amd.engine import make_opts, make_grid
    from oracle import oracle
    name, num = 'dblgauss_c2', 14
    rng = np.random.default_rng(5 + world)
    rays = rng.permutation(num * num)[:41].tolist() + [0, num * num - 1]
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29500 + (os.getpid() * 11 + 600 + world) % 2000
    procs = [ctx.Process(target=_packets_worker, args=(r, world, port, q, num, name, rays, dst))
             for r in range(world)]
    for p in procs:
        p.start()
    out = {}
    for _ in procs:
        r, held, got, ok_empty = q.get(timeout=180)
        out[r] = (held, got)
        assert ok_empty
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # every ray is held exactly once, by pupil rows
    assert sum(h for h, _ in out.values()) == num * num
    assert all(h % num == 0 for h, _ in out.values())
    for r in range(world):
        assert (out[r][1] is not None) == (r == dst)
    got = out[dst][1]
    wl = workloads.load(name)
    opts = make_opts(flags=abi.INTERSECT_OBJ | abi.CHECK_APERTURES | abi.APPLY_VIGNETTING,
                     out_mode=abi.OUT_FULL, first_surf=1, last_surf=wl.n_ifcs - 2)
    ref = oracle.trace_pupil_grid(wl.table, wl.fields[-1], make_grid((-1., -1.), (1., 1.), num), 0, opts)
    idx = np.asarray(rays)
    np.testing.assert_array_equal(got['status'], ref.status[idx])
    np.testing.assert_array_equal(got['fail_surf'], ref.fail_surf[idx])
    ok = ref.status[idx] == abi.OK
    assert ok.any() and (~ok).any()
    np.testing.assert_array_equal(got['op'][ok], ref.op[idx][ok])
    np.testing.assert_array_equal(got['seg'][:, :, ok], ref.seg[:wl.n_ifcs][:, :, idx][:, :, ok])


# ---------------------------------------------------------------- bench.py's exchange pre-flight
def _preflight_worker(rank, world, port, q, skip_on_rank1):
    sys.path.insert(0, ROOT)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    import importlib.util
    import torch
    dist.init_process_group('gloo', rank=rank, world_size=world)
    spec = importlib.util.spec_from_file_location('rox_bench', os.path.join(ROOT, 'bench.py'))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    skip = skip_on_rank1 if rank == 1 else ()
    out = b.exchange_preflight(torch, dist, world, rank, None, 3.0, skip=skip)
    q.put((rank, out))
    q.close()
    q.join_thread()     # (the feeder thread must have written the record before the exit below)
    os._exit(0)         # (a rank left waiting in a collective never returns from it)


def _preflight(world, skip_on_rank1, salt):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29500 + (os.getpid() * 7 + salt) % 2000
    procs = [ctx.Process(target=_preflight_worker, args=(r, world, port, q, skip_on_rank1)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(30)
    return got


def test_bench_exchange_preflight_on_three_ranks():
    """bench.py --gpus N runs the exchange's collectives once, tiny, before anything is timed: all
    three steps complete on a healthy group and are timed"""
    got = _preflight(3, (), 901)
    for r in range(3):
        assert got[r]['ok'], got[r]
        assert list(got[r]['steps_ms']) == ['all_gather_on_side_stream', 'grouped_isend_irecv_to_rank0',
                                            'all_reduce_fence']


def test_bench_exchange_preflight_names_the_collective_that_hangs():
    """rank 1 stays away from the grouped send: rank 0, waiting for it, reports within the
    watchdog's 3 s WHICH step hung instead of sitting in the backend's own timeout"""
    got = _preflight(2, ('grouped_isend_irecv_to_rank0',), 902)
    assert got[0]['ok'] is False and got[0]['hung_in'] == 'grouped_isend_irecv_to_rank0', got[0]
    assert 'all_gather_on_side_stream' in got[0]['steps_ms']
