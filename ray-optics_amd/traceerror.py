"""Per-ray failures.  The kernels report a status byte + failing surface; the
host side turns them back into the reference's exception objects
(rayoptics/raytr/traceerror.py:12-52) only when a caller asks for them
(``rayerr_filter``, rayoptics/raytr/trace.py:193-205).  When the reference is
importable its own classes are used, so ``isinstance`` checks in reference
consumers keep working; otherwise equivalent local classes are defined."""
from . import abi

try:
    from rayoptics.raytr.traceerror import (TraceError, TraceMissedSurfaceError,  # noqa: F401
                                            TraceTIRError, TraceRayBlockedError,
                                            TraceEvanescentRayError)
except Exception:      # reference not installed: equivalent classes, built here
    class TraceError(Exception):
        """per-ray trace failure; carries .surf and .ray_pkg like the reference's"""

        def __init__(self, *args, **kw):
            super().__init__()
            self.surf = kw.get('surf')
            self.ray_pkg = kw.get('ray_pkg')
            for name, val in zip(self._fields, args):
                setattr(self, name, val)

        _fields = ()

    def _kind(name, fields, doc):
        return type(name, (TraceError,), {'_fields': fields, '__doc__': doc,
                                          **{f: None for f in fields}})

    TraceMissedSurfaceError = _kind('TraceMissedSurfaceError', ('ifc', 'prev_seg'),
                                    'the ray misses an interface')
    TraceTIRError = _kind('TraceTIRError', ('inc_dir', 'normal', 'prev_indx', 'follow_indx'),
                          'total internal reflection at an interface')
    TraceTIRError.ifc = TraceTIRError.int_pt = None
    TraceRayBlockedError = _kind('TraceRayBlockedError', ('ifc', 'int_pt'),
                                 'the ray is blocked by an aperture')
    TraceEvanescentRayError = _kind('TraceEvanescentRayError',
                                    ('ifc', 'int_pt', 'inc_dir', 'normal', 'prev_indx',
                                     'follow_indx'), 'evanescent diffracted ray')


def make_error(status, surf, ifc=None, ray_pkg=None, int_pt=None, inc_dir=None,
               normal=None, n_in=None, n_out=None):
    """rebuild the exception trace_raw would have raised
    (rayoptics/raytr/raytrace.py:231-257 sets .surf/.ifc/.ray_pkg/.int_pt)"""
    if status == abi.MISSED_SURFACE:
        e = TraceMissedSurfaceError(ifc, None)
    elif status == abi.TIR:
        e = TraceTIRError(inc_dir, normal, n_in, n_out)
        e.ifc = ifc
        e.int_pt = int_pt
    elif status == abi.BLOCKED:
        e = TraceRayBlockedError(ifc, int_pt)
    else:
        e = TraceEvanescentRayError(ifc, int_pt, inc_dir, normal, n_in, n_out)
    e.surf = surf
    e.ray_pkg = ray_pkg
    return e
