"""engine cache + policy for models the kernels do not cover.

``engine_for(opt_model)`` re-reads the surface table on every call (a walk
over N interfaces) and reuses the device handle while the table bytes are
unchanged, so a model edit can never be traced with a stale table.

FALLBACK decides what happens for *models* outside the kernels' scope (phase
elements, toroids, wide-angle ray starts ...):
  'raise'      (default) UnsupportedModelError
  'reference'  the call is handed to the reference's own, unmodified function
A missing library or GPU is never a fallback case: that always raises.
"""
from .table import SurfaceTable

ENGINE_FACTORY = None       # None -> engine.TraceEngine (the HIP path)
FALLBACK = 'raise'
_cache = {}


def _factory():
    if ENGINE_FACTORY is not None:
        return ENGINE_FACTORY
    from .engine import TraceEngine
    return TraceEngine


def engine_for(opt_model):
    sm = opt_model['seq_model']
    table = SurfaceTable.from_seq_model(sm)
    sig = bytes(table.rows) + table.n_table.tobytes() + repr(table.wvls).encode()
    hit = _cache.get(id(sm))
    if hit is not None and hit[0] == sig:
        return hit[1]
    if hit is not None:
        hit[1].close()
    eng = _factory()(table)
    _cache[id(sm)] = (sig, eng)
    return eng


def clear():
    for _sig, eng in _cache.values():
        eng.close()
    _cache.clear()
