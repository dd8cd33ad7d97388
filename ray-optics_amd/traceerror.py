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
except Exception:      # reference not installed: same names, same attributes
    class TraceError(Exception):
        def __init__(self, surf=None, ray_pkg=None):
            self.surf = surf
            self.ray_pkg = ray_pkg

    class TraceMissedSurfaceError(TraceError):
        def __init__(self, ifc=None, prev_seg=None):
            self.ifc = ifc
            self.prev_seg = prev_seg

    class TraceTIRError(TraceError):
        def __init__(self, inc_dir, normal, prev_indx, follow_indx):
            self.ifc = None
            self.int_pt = None
            self.inc_dir = inc_dir
            self.normal = normal
            self.prev_indx = prev_indx
            self.follow_indx = follow_indx

    class TraceEvanescentRayError(TraceError):
        def __init__(self, ifc, int_pt, inc_dir, normal, prev_indx, follow_indx):
            self.ifc = ifc
            self.int_pt = int_pt

    class TraceRayBlockedError(TraceError):
        def __init__(self, ifc, int_pt):
            self.ifc = ifc
            self.int_pt = int_pt


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
