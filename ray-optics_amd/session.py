"""engine cache + policy for models the kernels do not cover.

``engine_for(opt_model)`` returns the device engine of the model's current
surface table.  Validity is checked on every call, cheaply:

* the snapshots the reference itself traces from -- ``seq_model.lcl_tfrms`` and
  ``seq_model.rndx`` are rebuilt as new lists by ``SequentialModel.update_model``
  (rayoptics/seq/sequential.py:600-668, where the reference clears its own
  ``path_sequence`` cache) -- are compared by identity, and
* everything ``path()`` reads *through* the live interface objects (profile
  parameters, apertures, phase elements, interact modes) is compared by value
  (:func:`_fingerprint`), because the reference sees such edits without an
  ``update_model()``.

Only when either differs is the table re-extracted (``SurfaceTable.from_seq_model``)
and, if its bytes changed, a new device handle created: a stale table is never
traced.  Models that carry a prebuilt table (``seq_model.surface_table``:
:class:`~.workloads.TableModel`, tables parsed from prescription files) use it
as is.

FALLBACK decides what happens for *models* outside the kernels' scope (unknown
interface, profile or phase classes):
  'raise'      (default) UnsupportedModelError
  'reference'  the call is handed to the reference's own, unmodified function
A missing library or GPU is never a fallback case: that always raises.

Threads: the caches are guarded by one lock (``engine_for`` / ``engine_for_table`` /
``clear`` are safe to call from any thread; a new handle is created under the lock).  Eviction
``close()``s the least recently used engine; ``TraceEngine.close`` defers the destruction of
the device handle until no call is in flight on it, and an evicted engine that a thread still
holds re-creates its handle on its next call (engine.py), so an engine returned by
``engine_for`` stays usable whatever other threads do to the cache.
"""
import threading
import weakref

from .table import SurfaceTable

FALLBACK = 'raise'
# TOLERANCE_MODE = True: every launch the drop-in layer makes carries ROX_FAST_FP64
# (include/roxtrace.h): the reduced-output modes -- spot diagrams, wavefront grids, ray fans,
# 'last'-filtered traces -- run on the tolerance-mode kernels (results within 1e-10 of the
# reference instead of bit-identical, about twice as fast); FULL ray packets stay bit-exact.
# Set by ``rayoptics_amd.install(tolerance_mode=True)`` or :func:`set_tolerance_mode`.
TOLERANCE_MODE = False
MAX_ENGINES = 16            # device handles kept alive (least recently used out first)
_cache = {}                 # id(seq_model) -> _Entry, in LRU order
_lock = threading.RLock()
_engine_factory = None      # None -> engine.TraceEngine (the HIP path)


def set_tolerance_mode(on=True):
    """opt in to (or out of) the tolerance-mode kernels for the reduced-output modes; returns the
    previous setting"""
    global TOLERANCE_MODE
    was, TOLERANCE_MODE = TOLERANCE_MODE, bool(on)
    return was


def _set_engine_factory(factory):
    """TEST HOOK (tests/oracle_engine.py, tests/test_session_cache.py): the class the caches
    instantiate instead of engine.TraceEngine; ``None`` restores the HIP engine.  Nothing in
    the product calls it."""
    global _engine_factory
    with _lock:
        _engine_factory = factory


def _factory():
    if _engine_factory is not None:
        return _engine_factory
    from .engine import TraceEngine
    return TraceEngine


def _prof_key(p):
    if p is None:
        return None
    c = getattr(p, 'coefs', None)
    return (type(p), p.cv, getattr(p, 'cc', None), getattr(p, 'ec', None),
            getattr(p, 'cR', None), tuple(c) if c is not None else None)


def _ap_key(ca):
    return (type(ca), getattr(ca, 'radius', None), getattr(ca, 'x_half_width', None),
            getattr(ca, 'y_half_width', None), getattr(ca, 'x_offset', 0.0),
            getattr(ca, 'y_offset', 0.0), getattr(ca, 'is_obscuration', False))


def _phase_key(pe):
    if pe is None:
        return None
    d = getattr(pe, '__dict__', {})
    return (type(pe),) + tuple((k, tuple(v) if hasattr(v, '__len__') and not isinstance(v, str)
                                else v) for k, v in sorted(d.items()) if k != 'debug_output')


def _fingerprint(sm):
    """what path() reads through the live interface objects, by value"""
    out = []
    for ifc in sm.ifcs:
        cas = getattr(ifc, 'clear_apertures', None)
        out.append((type(ifc), ifc.interact_mode, ifc.max_aperture,
                    _prof_key(getattr(ifc, 'profile', None)),
                    tuple(_ap_key(ca) for ca in cas) if cas else None,
                    _phase_key(getattr(ifc, 'phase_element', None))
                    if hasattr(ifc, 'phase_element') else None))
    return out


class _Entry:
    __slots__ = ('ref', 'lcl_tfrms', 'rndx', 'wvls', 'finger', 'sig', 'engine')


def _evict():
    while len(_cache) > MAX_ENGINES:
        key = next(iter(_cache))
        _cache.pop(key).engine.close()


_held = threading.local()       # .models: {id(seq_model): [depth, engine, memo dict]}


class hold:
    """``with session.hold(opt_model):`` -- for the duration of ONE drop-in call that asks for
    the model's engine many times (a fan figure's chief rays and launches: 7 requests per fan),
    validate the model once and answer the rest from that: the fingerprint walk over the
    interfaces costs ~25 us a time.  Per thread; the model is not expected to change inside
    one call of the reference's API (the reference's own ``path()`` cache assumes as much).
    ``memo`` is a scratch dict for values that are constant over the same span
    (table.field_from_model)."""

    def __init__(self, opt_model):
        self._key = id(opt_model['seq_model'])
        self._model = opt_model

    def __enter__(self):
        models = _held.__dict__.setdefault('models', {})
        ent = models.get(self._key)
        if ent is None:
            models[self._key] = ent = [0, engine_for(self._model), {}]
        ent[0] += 1
        return ent[1]

    def __exit__(self, *exc):
        models = _held.models
        ent = models[self._key]
        ent[0] -= 1
        if ent[0] == 0:
            del models[self._key]
        return False


def held_memo(opt_model):
    """the scratch dict of the innermost ``hold`` on this model in this thread, or None"""
    models = getattr(_held, 'models', None)
    if not models:
        return None
    ent = models.get(id(opt_model['seq_model']))
    return ent[2] if ent is not None else None


def engine_for(opt_model):
    models = getattr(_held, 'models', None)
    if models:
        ent = models.get(id(opt_model['seq_model']))
        if ent is not None:
            return ent[1]
    with _lock:
        return _engine_for(opt_model)


def _engine_for(opt_model):
    sm = opt_model['seq_model']
    key = id(sm)
    ent = _cache.get(key)
    if ent is not None and ent.ref() is not sm:        # id reused by another object
        _cache.pop(key).engine.close()
        ent = None
    prebuilt = getattr(sm, 'surface_table', None)
    if prebuilt is not None:
        if ent is not None and ent.sig is prebuilt:
            return ent.engine
        table, sig, finger, wvls = prebuilt, prebuilt, None, None
    else:
        wvls = tuple(opt_model['osp']['wvls'].wavelengths)
        finger = _fingerprint(sm)
        if (ent is not None and ent.lcl_tfrms is sm.lcl_tfrms and ent.rndx is sm.rndx
                and ent.wvls == wvls and ent.finger == finger):
            _cache[key] = _cache.pop(key)              # most recently used
            return ent.engine
        table = SurfaceTable.from_seq_model(sm)
        sig = bytes(table.rows) + table.n_table.tobytes() + repr(table.wvls).encode()
        if ent is not None and ent.sig == sig:         # re-validated: same table bytes
            ent.lcl_tfrms, ent.rndx, ent.wvls, ent.finger = sm.lcl_tfrms, sm.rndx, wvls, finger
            return ent.engine
    if ent is not None:
        _cache.pop(key).engine.close()
    new = _Entry()
    new.ref = weakref.ref(sm) if _weakrefable(sm) else (lambda sm=sm: sm)
    new.lcl_tfrms = getattr(sm, 'lcl_tfrms', None)
    new.rndx = getattr(sm, 'rndx', None)
    new.wvls, new.finger, new.sig = wvls, finger, sig
    new.engine = _factory()(table)
    _cache[key] = new
    _evict()
    return new.engine


def _weakrefable(obj):
    try:
        weakref.ref(obj)
        return True
    except TypeError:
        return False


_by_table = {}              # table bytes -> engine (paths handed to trace_raw), LRU order
MAX_TABLE_ENGINES = 8


def engine_for_table(table):
    """the engine of a surface table that did not come from a SequentialModel (a path
    list handed to ``raytrace.trace_raw``: ``gen_sequence`` output, reversed paths),
    cached by the table's bytes"""
    sig = bytes(table.rows) + table.n_table.tobytes() + repr(table.wvls).encode()
    with _lock:
        eng = _by_table.pop(sig, None)
        if eng is None:
            eng = _factory()(table)
        _by_table[sig] = eng                # most recently used last
        while len(_by_table) > MAX_TABLE_ENGINES:
            _by_table.pop(next(iter(_by_table))).close()
        return eng


def clear():
    """close every cached engine and un-pin the host blocks the pinned pool keeps"""
    with _lock:
        for ent in _cache.values():
            ent.engine.close()
        _cache.clear()
        for eng in _by_table.values():
            eng.close()
        _by_table.clear()
    from . import engine
    engine._pool.trim()
