"""session.engine_for: the device handle of a live model is reused while the model is
unchanged, re-validated after an update_model() that changed nothing, replaced after an edit the
reference would see (with or without update_model()), and evicted least-recently-used first."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.needs_reference


class CountingEngine:
    made, closed = 0, 0

    def __init__(self, table, device=None):
        type(self).made += 1
        self.table = table
        self.open = True

    def close(self):
        if self.open:
            type(self).closed += 1
        self.open = False


@pytest.fixture()
def sess():
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), 'golden'))
    import refmodels as rm
    from rayoptics_amd import session
    session.clear()
    CountingEngine.made = CountingEngine.closed = 0
    session._set_engine_factory(CountingEngine)
    yield session, rm
    session.clear()
    session._set_engine_factory(None)


def test_handle_reuse_revalidation_and_replacement(sess):
    session, rm = sess
    opm = rm.dblgauss()
    sm = opm['seq_model']
    e1 = session.engine_for(opm)
    assert session.engine_for(opm) is e1 and CountingEngine.made == 1
    # update_model() rebuilds lcl_tfrms / rndx (new list objects) but nothing changed:
    # the table is re-extracted once, found identical, and the handle is kept
    sm.update_model()
    assert session.engine_for(opm) is e1 and CountingEngine.made == 1
    assert session.engine_for(opm) is e1
    # an edit the reference sees through the live profile object, without update_model()
    sm.ifcs[3].profile.cv *= 1.01
    e2 = session.engine_for(opm)
    assert e2 is not e1 and CountingEngine.made == 2 and CountingEngine.closed == 1 and not e1.open
    # an aperture edit, and a thickness edit followed by update_model()
    sm.ifcs[4].max_aperture *= 0.9
    e3 = session.engine_for(opm)
    assert e3 is not e2
    sm.gaps[2].thi += 0.05
    sm.update_model()
    e4 = session.engine_for(opm)
    assert e4 is not e3 and e4.table.rows[2].t[2] == sm.gaps[2].thi
    assert CountingEngine.made == 4 and CountingEngine.closed == 3


def test_phase_element_edits_are_seen(sess):
    session, rm = sess
    from rayoptics.oprops import doe
    opm = rm.singlet()
    sm = opm['seq_model']
    sm.ifcs[2].phase_element = doe.DiffractionGrating(grating_lpmm=300.0, order=1)
    e1 = session.engine_for(opm)
    assert session.engine_for(opm) is e1
    sm.ifcs[2].phase_element.order = 2
    e2 = session.engine_for(opm)
    assert e2 is not e1 and e2.table.rows[2].ph.order == 2.0


def test_least_recently_used_handles_are_closed(sess):
    session, rm = sess
    session.MAX_ENGINES, saved = 3, session.MAX_ENGINES
    try:
        models = [rm.singlet() for _ in range(5)]
        engines = [session.engine_for(m) for m in models[:3]]
        session.engine_for(models[0])                           # touch: 1 is now the oldest
        engines.append(session.engine_for(models[3]))
        assert not engines[1].open and engines[0].open and engines[2].open
        engines.append(session.engine_for(models[4]))
        assert not engines[2].open and engines[0].open
        assert len(session._cache) == 3
    finally:
        session.MAX_ENGINES = saved


def test_table_keyed_handles_for_explicit_paths(sess):
    session, rm = sess
    from rayoptics_amd import SurfaceTable
    session.MAX_TABLE_ENGINES, saved = 2, session.MAX_TABLE_ENGINES
    try:
        tabs = []
        for k in range(3):
            opm = rm.singlet()
            opm['seq_model'].ifcs[1].profile.cv *= 1 + 0.01 * k
            tabs.append(SurfaceTable.from_paths([list(opm['seq_model'].path(opm['seq_model'].central_wavelength()))],
                                                [opm['seq_model'].central_wavelength()]))
        a = session.engine_for_table(tabs[0])
        assert session.engine_for_table(SurfaceTable.from_dict(tabs[0].to_dict())) is a   # same bytes
        b = session.engine_for_table(tabs[1])
        session.engine_for_table(tabs[0])                       # touch
        c = session.engine_for_table(tabs[2])
        assert not b.open and a.open and c.open
    finally:
        session.MAX_TABLE_ENGINES = saved


def test_pinned_pool_keeps_blocks_by_a_byte_budget():
    """engine.PinnedPool (no GPU needed: a stand-in for torch's pinned tensors): a figure's nine
    same-size arrays all come back to the pool and are reused (a bound of four per size cost
    3 ms of un-pinning / re-pinning per block and refresh); a sweep over many sizes stays within
    the budget by dropping other sizes first; one block larger than the whole budget is kept"""
    import gc
    from rayoptics_amd.engine import PinnedPool

    pinned = []

    class _T:
        def __init__(self, n):
            self.n = n
            pinned.append(n)

        def data_ptr(self):
            return 0

    class _Torch:
        uint8 = None

        @staticmethod
        def empty(n, dtype=None):
            class _E:
                def pin_memory(self):
                    return _T(n)
            return _E()

    pool = PinnedPool()
    pool.budget = 64 << 20
    one = 1 << 20
    for _ in range(3):                      # three refreshes of a figure with nine 1 MiB arrays
        held = [pool.take(_Torch, one) for _ in range(9)]
        del held
        gc.collect()
    assert len(pinned) == 9 and pool._held == 9 * one
    for k in range(1, 40):                  # a sweep over 39 other sizes, 4 MiB each rounded up
        lease = pool.take(_Torch, 3 * one + 4096 * k)
        del lease
    gc.collect()
    assert pool._held <= pool.budget
    big = pool.take(_Torch, 200 << 20)      # larger than the whole budget
    del big
    gc.collect()
    n_pins = len(pinned)
    again = pool.take(_Torch, 200 << 20)
    assert len(pinned) == n_pins            # ... and still reused
    del again


def test_pinned_pool_under_threads():
    """take / give_back from eight threads (leases dropped on any thread, some inside
    collections): no block is handed out twice at the same time, nothing is lost, and the
    byte budget holds"""
    import gc
    import threading
    from rayoptics_amd.engine import PinnedPool

    made = []
    lock = threading.Lock()

    class _T:
        def __init__(self, n):
            self.n = n
            self.owner = None
            with lock:
                made.append(self)

        def data_ptr(self):
            return id(self)

    class _Torch:
        uint8 = None

        @staticmethod
        def empty(n, dtype=None):
            class _E:
                def pin_memory(self):
                    return _T(n)
            return _E()

    pool = PinnedPool()
    pool.budget = 8 << 20
    errors = []

    def worker(tid):
        rng = np.random.default_rng(tid)
        held = []
        for it in range(3000):
            lease = pool.take(_Torch, int(rng.choice([4096, 65536, 1 << 20])))
            t = lease.tensor
            if t.owner is not None:
                errors.append('block handed out twice')
                return
            t.owner = tid
            held.append(lease)
            if len(held) > 4 or rng.random() < 0.3:
                old = held.pop(0)
                old.tensor.owner = None
                del old
            if it % 500 == 0:
                gc.collect()
        for l in held:
            l.tensor.owner = None

    ths = [threading.Thread(target=worker, args=(t,)) for t in range(8)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    assert not errors
    gc.collect()
    assert pool._held <= pool.budget
    assert pool.trim() >= 0 and pool._held == 0


def test_engine_cache_under_threads(sess):
    """engine_for from six threads over more models than MAX_ENGINES: one engine per model
    state at any time (no two threads build the same model's engine concurrently), the cache
    never exceeds its bound"""
    import threading
    session, rm = sess
    session.MAX_ENGINES, saved = 4, session.MAX_ENGINES
    try:
        models = [rm.singlet() for _ in range(7)]
        errors = []

        def worker(tid):
            rng = np.random.default_rng(tid)
            for _ in range(300):
                m = models[int(rng.integers(len(models)))]
                e = session.engine_for(m)
                if e.table is None or len(session._cache) > session.MAX_ENGINES:
                    errors.append('bad cache state')
        ths = [threading.Thread(target=worker, args=(t,)) for t in range(6)]
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        assert not errors
        assert len(session._cache) <= 4
        # every engine ever made and no longer cached was closed exactly once
        assert CountingEngine.made - CountingEngine.closed == len(session._cache)
    finally:
        session.MAX_ENGINES = saved
