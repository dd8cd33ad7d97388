"""-m gpu, round 4: packed hits in two passes (plain HITS launch + csrc/pack.hip) must be
indistinguishable from the fused HITS_COMPACT instance; the capacity clamp of packed hits."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

from rayoptics_amd import abi

pytestmark = pytest.mark.gpu

SPOT = abi.INTERSECT_OBJ | abi.CHECK_APERTURES | abi.APPLY_VIGNETTING
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def pack_counts(lib):
    c = (C.c_uint64 * 2)()
    assert lib.rox_diag_pack_launches(c) == 0
    return int(c[0]), int(c[1])


def oracle_hits(wl, fi, wi, **grid_kw):
    from oracle import oracle
    N = wl.n_ifcs
    o = oracle.make_opts(flags=SPOT, out_mode=abi.OUT_HITS_COMPACT, first_surf=1, last_surf=N - 2,
                         foc=wl.foc, image_pt=wl.image_pts[fi])
    return oracle.trace_pupil_grid(wl.table, wl.fields[fi],
                                   oracle.make_grid((-1., -1.), (1., 1.), **grid_kw), wi, o).hits


@pytest.fixture()
def form():
    """sets ROX_PACK_TWO_PASS for the library (read per call) and restores it"""
    saved = os.environ.get('ROX_PACK_TWO_PASS')

    def set_(v):
        os.environ['ROX_PACK_TWO_PASS'] = v
    yield set_
    if saved is None:
        os.environ.pop('ROX_PACK_TWO_PASS', None)
    else:
        os.environ['ROX_PACK_TWO_PASS'] = saved


@pytest.mark.parametrize('name,num', [('dblgauss_c2', 300), ('nikkor_c3', 257), ('litho_c5', 200),
                                      ('cell_phone', 129), ('rc_telescope_c4', 64)])
def test_two_pass_packed_hits_equal_fused_and_oracle(form, name, num):
    """every field at two wavelengths, appended block after block (whole grids and ragged row
    blocks, sizes that are no multiple of the 4 096-ray pack tile) into one HBM buffer: the two
    forms give the same bytes and counts, which are the oracle's"""
    from rayoptics_amd import workloads
    from rayoptics_amd.engine import TraceEngine, make_opts, make_grid
    wl = workloads.load(name)
    N = wl.n_ifcs
    eng = TraceEngine(wl.table)
    nw = len(wl.table.wvls)
    jobs = []
    for fi in range(len(wl.fields)):
        for wi in sorted({0, nw - 1}):
            jobs.append((fi, wi, dict(num=num)))
        jobs.append((fi, 0, dict(num=num, row_begin=3, row_count=num // 3)))
        jobs.append((fi, 0, dict(num=num, row_begin=num - 1, row_count=1)))
    cap = sum((kw.get('row_count') or kw['num']) * kw['num'] for _f, _w, kw in jobs)
    got = {}
    for mode in ('0', '1'):
        form(mode)
        before = pack_counts(eng.lib)
        pack = eng.hits_pack(cap, len(jobs))
        for fi, wi, kw in jobs:
            o = make_opts(flags=SPOT | abi.HITS_APPEND, out_mode=abi.OUT_HITS_COMPACT, first_surf=1,
                          last_surf=N - 2, foc=wl.foc, image_pt=wl.image_pts[fi])
            eng.trace_pupil_grid_hits_append(wl.fields[fi], make_grid((-1., -1.), (1., 1.), **kw),
                                             wi, o, pack)
        counts = pack.counts()
        after = pack_counts(eng.lib)
        took = (after[0] - before[0], after[1] - before[1])
        assert took == ((len(jobs), 0) if mode == '0' else (0, len(jobs))), took
        got[mode] = (counts, pack.xy[:int(counts.sum())].cpu().numpy())
    np.testing.assert_array_equal(got['0'][0], got['1'][0])
    assert np.array_equal(got['0'][1], got['1'][1])
    want = [oracle_hits(wl, fi, wi, **kw) for fi, wi, kw in jobs]
    np.testing.assert_array_equal(got['1'][0], [len(w) for w in want])
    assert np.array_equal(got['1'][1], np.concatenate(want))
    eng.close()


def test_the_rule_picks_two_pass_for_deep_tables_into_hbm_only(form):
    """ROX_PACK_TWO_PASS unset: packed hits into device memory take two passes from 65 536
    rays up (measured never slower there, profiles/r04_pack_crossover.jsonl); small launches
    and pinned-host destinations (the pairs cross PCIe while later tiles are traced) keep
    the fused instance"""
    import torch
    from rayoptics_amd import workloads
    from rayoptics_amd.engine import TraceEngine, make_opts, make_grid, _pool
    form('')
    for name, num, where, want_two in [('litho_c5', 300, 'hbm', True), ('nikkor_c3', 300, 'hbm', True),
                                       ('litho_c5', 300, 'pinned', False), ('litho_c5', 100, 'hbm', False),
                                       ('dblgauss_c2', 300, 'hbm', True), ('dblgauss_c2', 300, 'pinned', False)]:
        wl = workloads.load(name)
        N = wl.n_ifcs
        eng = TraceEngine(wl.table)
        R = num * num
        dest = None
        if where == 'pinned':
            lease = _pool.take(torch, 16 * R)
            dest = (lease.ptr, R)
        pack = eng.hits_pack(R, 1, dest=dest)
        o = make_opts(flags=SPOT | abi.HITS_APPEND, out_mode=abi.OUT_HITS_COMPACT, first_surf=1,
                      last_surf=N - 2, foc=wl.foc, image_pt=wl.image_pts[1])
        before = pack_counts(eng.lib)
        eng.trace_pupil_grid_hits_append(wl.fields[1], make_grid((-1., -1.), (1., 1.), num), 0, o, pack)
        n = int(pack.counts()[0])
        after = pack_counts(eng.lib)
        assert (after[1] - before[1] == 1) == want_two, (name, num, where)
        assert (after[0] - before[0] == 1) == (not want_two), (name, num, where)
        xy = pack.xy[:n].cpu().numpy() if dest is None else lease.array((n, 2), np.float64)
        assert np.array_equal(xy, oracle_hits(wl, 1, 0, num=num)), (name, where)
        eng.close()


@pytest.mark.parametrize('mode', ['0', '1'])
def test_packed_hits_never_write_beyond_the_capacity(form, mode):
    """rox_out.ld is the capacity of seg in pairs: an appending call that needs
    more room stores nothing at or beyond it and leaves the NEGATED pair count it needed in
    n_hits; a call that does not append is refused when ld < n_rays"""
    import torch
    from rayoptics_amd import workloads
    from rayoptics_amd.engine import TraceEngine, make_opts, make_grid, EngineError
    form(mode)
    wl = workloads.load('dblgauss_c2')
    N = wl.n_ifcs
    eng = TraceEngine(wl.table)
    num = 150
    grid = make_grid((-1., -1.), (1., 1.), num)
    want = oracle_hits(wl, 0, 1, num=num)
    n_ok = len(want)
    cap = n_ok + n_ok // 2                  # the second appended grid does not fit
    buf = torch.full((cap + 4096, 2), -7.0, dtype=torch.float64, device=eng.device)
    count = torch.zeros(1, dtype=torch.int64, device=eng.device)
    o = abi.Out()
    o.seg, o.n_hits, o.ld = buf.data_ptr(), count.data_ptr(), cap
    opts = make_opts(flags=SPOT | abi.HITS_APPEND, out_mode=abi.OUT_HITS_COMPACT, first_surf=1,
                     last_surf=N - 2, foc=wl.foc, image_pt=wl.image_pts[0])
    st = C.c_void_p(torch.cuda.current_stream(eng.device).cuda_stream)
    for k in range(3):
        rc = eng.lib.rox_trace_pupil_grid(eng._handle, C.byref(wl.fields[0]), C.byref(grid), 1,
                                          C.byref(opts), C.byref(o), st)
        assert rc == 0, eng.lib.rox_last_error()
        torch.cuda.synchronize()
        n = int(count.item())
        assert n == (n_ok if k == 0 else -(k + 1) * n_ok), (k, n)
    got = buf.cpu().numpy()
    assert np.array_equal(got[:n_ok], want)
    assert np.array_equal(got[n_ok:cap], want[:cap - n_ok])         # what fitted of the second grid
    assert (got[cap:] == -7.0).all()                                  # nothing beyond the capacity
    # not appending: the capacity must cover every ray of the call
    opts2 = make_opts(flags=SPOT, out_mode=abi.OUT_HITS_COMPACT, first_surf=1, last_surf=N - 2,
                      foc=wl.foc, image_pt=wl.image_pts[0])
    o.ld = num * num - 1
    rc = eng.lib.rox_trace_pupil_grid(eng._handle, C.byref(wl.fields[0]), C.byref(grid), 1,
                                      C.byref(opts2), C.byref(o), st)
    assert rc != 0 and b'ld' in eng.lib.rox_last_error()
    # the Python wrapper reports an overflowing pack
    pack = eng.hits_pack(2 * num * num, 4)
    pack.cap = n_ok + 10                     # lie about the room: the kernel's clamp must hold
    o3 = make_opts(flags=SPOT | abi.HITS_APPEND, out_mode=abi.OUT_HITS_COMPACT, first_surf=1,
                   last_surf=N - 2, foc=wl.foc, image_pt=wl.image_pts[0])
    pack.rays = -4 * num * num               # (and get past the wrapper's own ray-count check)
    eng.trace_pupil_grid_hits_append(wl.fields[0], grid, 1, o3, pack)
    eng.trace_pupil_grid_hits_append(wl.fields[0], grid, 1, o3, pack)
    with pytest.raises(EngineError, match='overflow'):
        pack.counts()
    eng.close()


def test_the_compaction_suites_pass_in_the_two_pass_form():
    """the existing packed-hits tests -- compaction at 1 ... 10^6 rays on one and two streams,
    appended and chunked launches, lists and explicit rays, the sharded spot diagram -- re-run
    in a subprocess with ROX_PACK_TWO_PASS=1"""
    env = dict(os.environ, ROX_PACK_TWO_PASS='1')
    r = subprocess.run([sys.executable, '-m', 'pytest', '-x', '-q', '-m', 'gpu', '-p', 'no:cacheprovider',
                        os.path.join(ROOT, 'tests', 'test_gpu_r02.py'),
                        os.path.join(ROOT, 'tests', 'test_gpu_r03.py'),
                        os.path.join(ROOT, 'tests', 'test_gpu_parity.py'),
                        '-k', 'compact or append or hits or sharded or chunked or spot or stream'],
                       env=env, capture_output=True, text=True, timeout=1500, cwd=ROOT)
    tail = r.stdout[-1500:] + r.stderr[-1500:]
    assert r.returncode == 0, tail
    assert ' passed' in r.stdout and 'failed' not in r.stdout, tail


def test_iterate_ray_raw_returns_the_last_trial_ray():
    """rox_iterate_ray_raw (trace.py:866-961): aim point, result code AND the last trial ray
    the iteration evaluated -- its pupil-plane coordinates and trace status, what the
    reference keeps as `rr` -- device == oracle bit for bit, both branches, perturbed problems
    incl. ones whose trial rays fail; with last_xy = NULL it is rox_aim_chief_rays"""
    from oracle import oracle
    from rayoptics_amd import workloads
    from rayoptics_amd.engine import TraceEngine
    from test_gpu_r03 import aim2d_problem
    rng = np.random.default_rng(404)
    n_fail_last = n_2d = n = 0
    for name in ('dblgauss_c2', 'nikkor_c3', 'cell_phone', 'zmx_evenasph_c3', 'rc_telescope_c4', 'litho_c5'):
        wl = workloads.load(name)
        eng = TraceEngine(wl.table)
        allp = []
        for w in range(len(wl.table.wvls)):
            for m in (wl.aim2d or []):
                allp.append(aim2d_problem(m, w))
                for _ in range(3):
                    p = aim2d_problem(m, w)
                    p.pt0[0] *= rng.uniform(0.2, 3.0)
                    p.pt0[1] *= rng.uniform(0.2, 3.0)
                    p.z_enp *= rng.uniform(0.7, 1.3)
                    p.epsfcn *= 10.0 ** rng.uniform(-3, 1)
                    p.x_target, p.y_target = rng.normal(size=2) * 0.05
                    allp.append(p)
            for m in wl.aim or []:
                for scale in (1.0, rng.uniform(0.3, 2.5), rng.uniform(2.0, 6.0)):
                    a = abi.Aim()
                    for i in range(3):
                        a.pt0[i] = m['pt0'][i] * (scale if i < 2 else 1.0)
                    a.z_enp, a.y_target, a.z_dir0 = m['z_enp'], 0.0, m['z_dir0']
                    a.wvl_idx, a.surf, a.flip = w, m['surf'], 1
                    allp.append(a)
        if not allp:            # (a workload without stored aiming problems)
            eng.close()
            continue
        a_dev, r_dev, l_dev, s_dev = eng.iterate_ray_raw(allp)
        a_orc, r_orc, l_orc, s_orc = oracle.iterate_ray_raw(wl.table, allp)
        np.testing.assert_array_equal(r_dev, r_orc)
        np.testing.assert_array_equal(s_dev, s_orc)
        assert np.array_equal(a_dev, a_orc, equal_nan=True), name
        assert np.array_equal(l_dev, l_orc, equal_nan=True), name
        a2, r2 = eng.aim_chief_rays(allp)
        assert np.array_equal(a2, a_dev, equal_nan=True) and np.array_equal(r2, r_dev)
        n += len(allp)
        n_2d += sum(p.two_d for p in allp)
        n_fail_last += int((s_dev != abi.OK).sum())
        eng.close()
    assert n > 300 and n_2d > 50 and n_fail_last > 0


def test_iterate_pupil_rays_on_the_device():
    """rox_iterate_pupil_rays (vigcalc.iterate_pupil_ray, vigcalc.py:396-461, as set_pupil
    calls it): the pupil coordinate whose ray meets an interface at a target radius, device vs
    oracle -- within 1e-12 (the objective's `p[0]**2` is libm pow in the reference and in the
    oracle, a correctly rounded product on the device), incl. problems whose trial
    rays miss before the target surface (the reference's 0.9 x rule)"""
    from oracle import oracle
    from rayoptics_amd import workloads
    from rayoptics_amd.engine import TraceEngine
    rng = np.random.default_rng(77)
    n = n_far = 0
    for name in ('dblgauss_c2', 'nikkor_c3', 'cell_phone', 'rc_telescope_c4', 'singlet_c1'):
        wl = workloads.load(name)
        N = wl.n_ifcs
        eng = TraceEngine(wl.table)
        stop = wl.table.stop_idx if wl.table.stop_idx is not None else 1
        probs = []
        for fld in wl.fields:
            for wi in range(len(wl.table.wvls)):
                for indx in sorted({stop, 1, N - 2}):
                    r_edge = wl.table.rows[indx].max_aperture
                    for xy in (0, 1):
                        for r0, sgn, scale in ((1.0, 1.0, 1.0), (0.6, 1.0, 0.5), (-1.0, -1.0, 1.0),
                                               (float(rng.uniform(0.2, 1.5)), 1.0, float(rng.uniform(0.3, 3.0)))):
                            p = abi.PupilIter()
                            p.fld = fld
                            p.start_r0, p.r_target = r0, sgn * scale * r_edge
                            p.xy, p.wvl_idx, p.indx = xy, wi, indx
                            probs.append(p)
        dev = eng.iterate_pupil_rays(probs)
        orc = oracle.iterate_pupil_rays(wl.table, probs)
        assert np.isfinite(orc).all() == np.isfinite(dev).all()
        ok = np.isfinite(orc)
        assert np.array_equal(ok, np.isfinite(dev)), name
        np.testing.assert_allclose(dev[ok], orc[ok], rtol=0, atol=1e-11, err_msg=name)
        n += len(probs)
        n_far += int((np.abs(orc[ok]) > 1.2).sum())
        eng.close()
    assert n > 500


def test_two_host_threads_share_one_stream():
    """two Python threads (ctypes releases the GIL) enqueueing pupil grids of
    DIFFERENT definitions on the SAME stream of one handle.  The cached pupil axes (their
    key, their size, a reallocation when a larger grid arrives) are per-stream state that
    prepare_grid rewrites: the whole prepare-and-launch of a call is one critical section per
    stream, so every call still traces its own grid"""
    import threading
    from oracle import oracle
    from rayoptics_amd import workloads
    from rayoptics_amd.engine import TraceEngine, make_opts, make_grid
    wl = workloads.load('dblgauss_c2')
    N = wl.n_ifcs
    eng = TraceEngine(wl.table)
    defs = [((-1., -1.), (1., 1.), 97), ((-0.7, -0.9), (0.8, 0.6), 301), ((-0.2, -1.0), (1.0, 0.3), 160)]
    opts = make_opts(flags=SPOT, out_mode=abi.OUT_HITS_COMPACT, first_surf=1, last_surf=N - 2,
                     foc=wl.foc, image_pt=wl.image_pts[2])
    want = []
    for k, (a, b, num) in enumerate(defs):
        o = oracle.make_opts(flags=SPOT, out_mode=abi.OUT_HITS_COMPACT, first_surf=1, last_surf=N - 2,
                             foc=wl.foc, image_pt=wl.image_pts[2])
        want.append(oracle.trace_pupil_grid(wl.table, wl.fields[2], oracle.make_grid(a, b, num), k, o).hits.copy())
    errs = []

    def worker(k):
        try:
            a, b, num = defs[k]
            grid = make_grid(a, b, num)
            for _ in range(40):             # (all threads: the device's current stream)
                xy = eng.trace_pupil_grid_hits(wl.fields[2], grid, k, opts)
                if not np.array_equal(xy, want[k]):
                    errs.append((k, 'mismatch'))
                    return
        except Exception as e:      # noqa: BLE001
            errs.append((k, repr(e)))
    ths = [threading.Thread(target=worker, args=(k,)) for k in range(3)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    assert not errs, errs
    eng.close()


def test_a_wave_or_a_lane_per_problem_give_the_same_answers():
    """Up to 1024 problems a search call runs one wave per problem and traces the trial rays
    that do not depend on each other side by side in its lanes (the secant iteration's two
    starting values; hybrd's value at the start and the two forward-difference points of its
    Jacobian; the ends of find_edge; the eight samples of a degenerate bracket); beyond that a
    lane per problem evaluates them one after the other.  Same roots, result codes and last
    trial rays, bit for bit -- and both equal the oracle's."""
    from oracle import oracle
    from rayoptics_amd import workloads
    from rayoptics_amd.engine import TraceEngine
    rng = np.random.default_rng(404)
    n = 0
    for name in ('dblgauss_c2', 'nikkor_c3', 'cell_phone', 'rc_telescope_c4'):
        wl = workloads.load(name)
        eng = TraceEngine(wl.table)
        probs = []
        for _ in range(8):
            for m in (wl.aim or []) + (wl.aim2d or []):
                a = abi.Aim()
                two_d = 'epsfcn' in m
                a.pt0[0] = m['pt0'][0] * rng.uniform(0.3, 1.5) if two_d else 0.0
                a.pt0[1], a.pt0[2] = m['pt0'][1] * rng.uniform(0.3, 1.4), m['pt0'][2]
                a.z_enp = m['z_enp'] * rng.uniform(0.95, 1.05)
                a.z_dir0 = m['z_dir0']
                a.wvl_idx, a.surf, a.flip = int(rng.integers(0, len(wl.table.wvls))), m['surf'], 1
                if two_d:
                    a.two_d, a.epsfcn = 1, m['epsfcn']
                probs.append(a)
        assert 0 < len(probs) <= 1024
        k = 1024 // len(probs) + 1
        wave = eng.iterate_ray_raw(probs)
        lane = eng.iterate_ray_raw(probs * k)
        orc = oracle.iterate_ray_raw(wl.table, probs)
        for w, l, o in zip(wave, lane, orc):
            o = np.asarray(o)
            assert np.array_equal(np.asarray(w), o, equal_nan=True), name
            assert np.array_equal(np.asarray(l), np.concatenate([o] * k), equal_nan=True), name
        n += len(probs)
        eng.close()
    assert n > 300


def test_sharded_packets_fetch_on_the_device():
    """dist.trace_packets_sharded / ShardedPackets.fetch with the HIP engine (one rank: the
    packets are selected out of the resident DeviceResult on the device and come back in the
    order asked for) == the oracle's packets of those rays"""
    from oracle import oracle
    from rayoptics_amd import workloads
    from rayoptics_amd import dist as rdist
    from rayoptics_amd.engine import TraceEngine, make_opts, make_grid
    wl = workloads.load('nikkor_c3')
    eng = TraceEngine(wl.table)
    num = 96
    opts = make_opts(flags=abi.INTERSECT_OBJ | abi.CHECK_APERTURES | abi.APPLY_VIGNETTING,
                     out_mode=abi.OUT_FULL, first_surf=1, last_surf=wl.n_ifcs - 2)
    sp = rdist.trace_packets_sharded(eng, wl.fields[-1], 1, num, opts)
    assert sp.world == 1 and sp.row_count == num
    rays = np.random.default_rng(12).permutation(num * num)[:500]
    got = sp.fetch(rays)
    ref = oracle.trace_pupil_grid(wl.table, wl.fields[-1], make_grid((-1., -1.), (1., 1.), num), 1, opts)
    np.testing.assert_array_equal(got['status'], ref.status[rays])
    np.testing.assert_array_equal(got['fail_surf'], ref.fail_surf[rays])
    ok = ref.status[rays] == abi.OK
    assert 50 < ok.sum() < 500
    assert np.array_equal(got['op'][ok].view(np.int64), ref.op[rays][ok].view(np.int64))
    want = ref.seg[:wl.n_ifcs][:, :, rays][:, :, ok]
    assert np.array_equal(np.ascontiguousarray(got['seg'][:, :, ok]).view(np.int64),
                          np.ascontiguousarray(want).view(np.int64))
    eng.close()


def test_rccl_calls_of_the_pipelined_exchange_over_the_real_backend():
    """RCCL refuses two ranks on one GPU, so the multi-rank logic of the pipelined exchange runs
    over gloo (tests/test_dist_gloo.py); what CAN run here over the real backend is its call
    pattern with one rank -- a slice all-gathered on a side stream and copied to pinned memory
    behind it, grouped isend / irecv of row slices of 2-D tensors (to itself), events for the
    copy stream: tools/rccl_self_p2p_probe.py"""
    import json
    import subprocess
    import sys
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(29300 + os.getpid() % 600))
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK'):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'rccl_self_p2p_probe.py')],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith('{')][-1]
    d = json.loads(line)
    assert d['backend'] == 'nccl' and d['self_p2p_and_count_gather_ok'] is True


def test_null_and_out_of_range_arguments_are_errors_not_crashes():
    """every entry point of include/roxtrace.h called through raw ctypes with null handles, null
    structures, negative counts and out-of-range indices: a negative rox_err and a message from
    rox_last_error(), never a crash (the process survives to make the last, valid call)"""
    import ctypes as C
    from rayoptics_amd import workloads
    from rayoptics_amd.engine import TraceEngine, load_library, make_opts, make_grid
    lib = load_library()
    wl = workloads.load('dblgauss_c2')
    eng = TraceEngine(wl.table)
    h = eng._handle
    N = wl.n_ifcs
    fld = wl.fields[0]
    opts = make_opts(flags=abi.INTERSECT_OBJ | abi.CHECK_APERTURES | abi.HOST_POINTERS, out_mode=abi.OUT_HITS,
                     first_surf=1, last_surf=N - 2)
    grid = make_grid((-1., -1.), (1., 1.), 8)
    seg = np.zeros((2, 64))
    st = np.zeros(64, dtype=np.uint8)
    out = abi.Out()
    out.seg, out.status, out.ld = seg.ctypes.data, st.ctypes.data, 64
    null = None
    bad = []

    def expect_error(what, rc):
        msg = lib.rox_last_error().decode()
        if not (rc < 0 and msg):
            bad.append((what, rc, msg))

    expect_error('device_count(NULL)', lib.rox_device_count(null))
    expect_error('pin(NULL)', lib.rox_pin_host_memory(null, 4096, C.byref(C.c_void_p())))
    expect_error('pin(size 0)', lib.rox_pin_host_memory(seg.ctypes.data, 0, C.byref(C.c_void_p())))
    expect_error('unpin(NULL)', lib.rox_unpin_host_memory(null))
    expect_error('copy_async(NULL)', lib.rox_copy_async(null, seg.ctypes.data, 16, null))
    sys_out = C.c_void_p()
    expect_error('create(NULL rows)', lib.rox_system_create(null, N, wl.table.n_table.ctypes.data, null, 1,
                                                           C.byref(sys_out)))
    expect_error('create(1 interface)', lib.rox_system_create(wl.table.rows, 1, wl.table.n_table.ctypes.data,
                                                             null, 1, C.byref(sys_out)))
    expect_error('create(0 wavelengths)', lib.rox_system_create(wl.table.rows, N, wl.table.n_table.ctypes.data,
                                                               null, 0, C.byref(sys_out)))
    assert lib.rox_system_destroy(null) == 0        # like free(NULL)
    expect_error('num_segments(NULL)', lib.rox_system_num_segments(null, 0, C.byref(C.c_int32())))
    expect_error('num_segments(NULL out)', lib.rox_system_num_segments(h, 0, null))
    # trace entries
    expect_error('pupil_grid(NULL sys)', lib.rox_trace_pupil_grid(null, C.byref(fld), C.byref(grid), 0,
                                                                 C.byref(opts), C.byref(out), null))
    expect_error('pupil_grid(NULL fld)', lib.rox_trace_pupil_grid(h, null, C.byref(grid), 0, C.byref(opts),
                                                                 C.byref(out), null))
    expect_error('pupil_grid(NULL grid)', lib.rox_trace_pupil_grid(h, C.byref(fld), null, 0, C.byref(opts),
                                                                  C.byref(out), null))
    expect_error('pupil_grid(NULL opts)', lib.rox_trace_pupil_grid(h, C.byref(fld), C.byref(grid), 0, null,
                                                                  C.byref(out), null))
    expect_error('pupil_grid(NULL out)', lib.rox_trace_pupil_grid(h, C.byref(fld), C.byref(grid), 0,
                                                                 C.byref(opts), null, null))
    expect_error('pupil_grid(wvl -1)', lib.rox_trace_pupil_grid(h, C.byref(fld), C.byref(grid), -1,
                                                               C.byref(opts), C.byref(out), null))
    g0 = make_grid((-1., -1.), (1., 1.), 0)
    expect_error('pupil_grid(num 0)', lib.rox_trace_pupil_grid(h, C.byref(fld), C.byref(g0), 0, C.byref(opts),
                                                              C.byref(out), null))
    small = abi.Out()
    small.seg, small.status, small.ld = seg.ctypes.data, st.ctypes.data, 8
    expect_error('pupil_grid(ld < rays)', lib.rox_trace_pupil_grid(h, C.byref(fld), C.byref(grid), 0,
                                                                  C.byref(opts), C.byref(small), null))
    expect_error('trace_rays(NULL rays)', lib.rox_trace_rays(h, 4, null, null, null, 0, C.byref(opts),
                                                            C.byref(out), null))
    expect_error('trace_rays(n < 0)', lib.rox_trace_rays(h, -4, seg.ctypes.data, seg.ctypes.data, null, 0,
                                                        C.byref(opts), C.byref(out), null))
    expect_error('pupil_list(NULL px)', lib.rox_trace_pupil_list(h, C.byref(fld), 4, null, null, 0,
                                                                C.byref(opts), C.byref(out), null))
    expect_error('pupil_grids(-1 grids)', lib.rox_trace_pupil_grids(h, -1, C.byref(fld), null, C.byref(grid),
                                                                  C.byref(opts), C.byref(out), null))
    # search entries
    res = np.zeros(8)
    ires = np.zeros(8, dtype=np.int32)
    expect_error('aim(NULL probs)', lib.rox_aim_chief_rays(h, 2, null, 1e-12, res.ctypes.data,
                                                          ires.ctypes.data, null))
    expect_error('aim(n < 0)', lib.rox_aim_chief_rays(h, -1, null, 1e-12, res.ctypes.data, ires.ctypes.data, null))
    a = abi.Aim()
    a.wvl_idx, a.surf = 99, 1
    expect_error('aim(wvl 99)', lib.rox_aim_chief_rays(h, 1, C.byref(a), 1e-12, res.ctypes.data,
                                                      ires.ctypes.data, null))
    a.wvl_idx, a.surf = 0, N + 5
    expect_error('aim(surf out of range)', lib.rox_aim_chief_rays(h, 1, C.byref(a), 1e-12, res.ctypes.data,
                                                                 ires.ctypes.data, null))
    expect_error('enp(NULL probs)', lib.rox_find_real_enp(h, 1, null, 1e-12, res.ctypes.data, ires.ctypes.data, null))
    expect_error('vig(NULL probs)', lib.rox_calc_vignetting(h, 1, null, 1e-12, res.ctypes.data,
                                                           ires.ctypes.data, null))
    expect_error('pupil_iter(NULL probs)', lib.rox_iterate_pupil_rays(h, 1, null, 1e-12, res.ctypes.data, null))
    expect_error('psf(NULL)', lib.rox_calc_psf(null, 8, 32, res.ctypes.data, abi.HOST_POINTERS, null))
    expect_error('psf(odd ndim)', lib.rox_calc_psf(res.ctypes.data, 7, 32, res.ctypes.data, abi.HOST_POINTERS, null))
    assert not bad, bad
    # and the handle still works
    assert lib.rox_trace_pupil_grid(h, C.byref(fld), C.byref(grid), 0, C.byref(opts), C.byref(out), null) == 0
    assert (st == abi.OK).any()
    eng.close()


def test_a_grid_of_two_to_the_32_rays_in_one_launch():
    """maximum size: a 65536 x 65536 pupil grid = 2^32 rays in ONE launch (73 GB of HITS output
    on the 288 GB part) -- ray indices above 2^31 land where they belong: four rows at 0.6 num
    equal the same rows traced as a row block, and sampled rays equal the oracle's, bit for bit
    (tools/huge_grid_check.py, in its own process so that the memory goes back at once)"""
    import json
    import subprocess
    import sys
    import torch
    free, _total = torch.cuda.mem_get_info()
    if free < 100 << 30:
        pytest.skip('needs 100 GB of free HBM')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'huge_grid_check.py'), '65536'],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{')][-1])
    assert d['rays'] == 2 ** 32 and d['above_int32']
    assert d['rows_equal_their_row_block'] and d['sampled_rays_equal_the_oracle']
    assert d['of_them_through'] > 100 and 0 < d['rays_through'] < d['rays']
    # FULL packets of an 8192 x 8192 grid: 70 GB, element offsets up to 8.7e9
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'huge_grid_check.py'), '--full'],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{')][-1])
    assert d['largest_element_offset'] > 2 ** 32 and d['rows_equal_their_row_block']
    assert d['rays_through_in_them'] > 10000
