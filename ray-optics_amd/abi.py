"""ctypes mirror of include/roxtrace.h (the C ABI of libroxtrace.so).

Only plain structs and constants live here; loading the HIP library is done
by :mod:`rayoptics_amd.engine`, which fails loudly when it is missing.
"""
import ctypes as C

ABI_VERSION = 8
MAX_COEF = 10
MAX_AP = 4
SEG_DOUBLES = 10

# interact_mode  (rayoptics/raytr/raytrace.py:211-221)
TRANSMIT, REFLECT, DUMMY, PHANTOM = 0, 1, 2, 3
MODE_NAMES = {'transmit': TRANSMIT, 'reflect': REFLECT, 'dummy': DUMMY,
              'phantom': PHANTOM}
# profile kinds  (rayoptics/elem/profiles.py)
SPHERICAL, CONIC, EVENPOLY, RADIALPOLY, YTOROID, XTOROID, THINLENS = 0, 1, 2, 3, 4, 5, 6
PROFILE_NAMES = {'Spherical': SPHERICAL, 'Conic': CONIC,
                 'EvenPolynomial': EVENPOLY, 'RadialPolynomial': RADIALPOLY,
                 'YToroid': YTOROID, 'XToroid': XTOROID}
# phase elements (rayoptics/oprops/doe.py)
PH_NONE, PH_GRATING, PH_DOE_RADIAL, PH_HOLOGRAM = 0, 1, 2, 3
# aperture kinds (rayoptics/elem/surface.py:398-494)
AP_CIRCULAR, AP_RECTANGULAR, AP_ALWAYS_BLOCK = 0, 1, 2
# per-ray status (rayoptics/raytr/traceerror.py)
OK, MISSED_SURFACE, TIR, BLOCKED, EVANESCENT = 0, 1, 2, 3, 4
# output modes
OUT_FULL, OUT_LAST, OUT_HITS, OUT_OPD, OUT_HITS_COMPACT, OUT_FAN = 0, 1, 2, 3, 4, 5
# flags
CHECK_APERTURES = 1
INTERSECT_OBJ = 2
FILTER_PHANTOMS = 4
APPLY_VIGNETTING = 8
HOST_POINTERS = 16
HITS_APPEND = 32
FAST_FP64 = 64            # tolerance mode (include/roxtrace.h ROX_FAST_FP64): reduced-output modes only
# summation order of rt.dot(v) (see include/roxtrace.h)
RT_F_ORDER, RT_C_ORDER = 0, 1
# grid kinds
GRID_PRODUCT, GRID_FAN = 0, 1
# rox_wavefront.kind
WF_FINITE, WF_INF_FULL, WF_INF_SPLIT = 0, 1, 2
# rox_field.kind: the branches of ray_start_from_osp (opticalspec.py:289-400)
FLD_EPD, FLD_EPD_WIDE, FLD_AIM_PT, FLD_NA, FLD_FNO, FLD_AIM_DIR = 0, 1, 2, 3, 4, 5
# rox_aim_chief_rays result codes
AIM_CONVERGED, AIM_NOT_CONVERGED, AIM_TRACE_ERROR = 0, 1, 2


class Aperture(C.Structure):
    _fields_ = [('kind', C.c_int32), ('is_obscuration', C.c_int32),
                ('x_offset', C.c_double), ('y_offset', C.c_double),
                ('a', C.c_double), ('b', C.c_double)]


class Phase(C.Structure):
    _fields_ = [('kind', C.c_int32), ('ncoef', C.c_int32),
                ('flags', C.c_int32), ('reserved', C.c_int32),
                ('order', C.c_double), ('ref_wl', C.c_double),
                ('spacing_nm', C.c_double),
                ('a', C.c_double * 3), ('b', C.c_double * 3),
                ('coefs', C.c_double * MAX_COEF)]


SURF_CV_INT_ZERO = 1        # rox_surface.flags (include/roxtrace.h ROX_SURF_CV_INT_ZERO)


class Surface(C.Structure):
    _fields_ = [('mode', C.c_int32), ('profile', C.c_int32),
                ('ncoef', C.c_int32), ('n_ap', C.c_int32),
                ('rt_order', C.c_int32), ('flags', C.c_int32),
                ('cv', C.c_double), ('cc', C.c_double), ('ec', C.c_double),
                ('cR', C.c_double),
                ('coefs', C.c_double * MAX_COEF),
                ('rt', C.c_double * 9), ('t', C.c_double * 3),
                ('z_dir', C.c_double), ('max_aperture', C.c_double),
                ('ap', Aperture * MAX_AP), ('ph', Phase)]


class Wavefront(C.Structure):
    _fields_ = [('cr1_p', C.c_double * 3), ('cr0_d', C.c_double * 3),
                ('crk_p', C.c_double * 3), ('crk_d', C.c_double * 3),
                ('cr_op', C.c_double), ('cr_exp_pt', C.c_double * 3),
                ('cr_exp_dist', C.c_double), ('ref_dir', C.c_double * 3),
                ('ref_radius', C.c_double), ('n_obj', C.c_double),
                ('n_img', C.c_double), ('sign_soln', C.c_double),
                ('after_kind', C.c_int32), ('after_order', C.c_int32),
                ('after_rt', C.c_double * 9), ('after_t', C.c_double * 3),
                ('kind', C.c_int32), ('last_kind', C.c_int32),
                ('last_order', C.c_int32), ('reserved', C.c_int32),
                ('last_rt', C.c_double * 9), ('last_t', C.c_double * 3),
                ('cr_last_p', C.c_double * 3), ('cr_last_d', C.c_double * 3),
                ('d_cr_b4', C.c_double * 3), ('v_be', C.c_double),
                ('image_pt', C.c_double * 3)]


class Opts(C.Structure):
    _fields_ = [('flags', C.c_uint32), ('out_mode', C.c_int32),
                ('first_surf', C.c_int32), ('last_surf', C.c_int32),
                ('eps', C.c_double), ('fuzz', C.c_double),
                ('foc', C.c_double), ('image_pt', C.c_double * 2),
                ('wf', Wavefront)]


class Field(C.Structure):
    _fields_ = [('pt0', C.c_double * 3), ('aim', C.c_double * 2),
                ('eprad', C.c_double), ('z_enp', C.c_double),
                ('vlx', C.c_double), ('vux', C.c_double),
                ('vly', C.c_double), ('vuy', C.c_double),
                ('z_dir0', C.c_double),
                ('kind', C.c_int32), ('rot_order', C.c_int32),
                ('rot', C.c_double * 9), ('cr_dir', C.c_double * 2)]


class Grid(C.Structure):
    _fields_ = [('start', C.c_double * 2), ('stop', C.c_double * 2),
                ('num', C.c_int32), ('kind', C.c_int32),
                ('row_begin', C.c_int32), ('row_count', C.c_int32)]


class Out(C.Structure):
    _fields_ = [('seg', C.c_void_p), ('op', C.c_void_p),
                ('status', C.c_void_p), ('fail_surf', C.c_void_p),
                ('pupil', C.c_void_p), ('ld', C.c_int64),
                ('n_hits', C.c_void_p)]


class Aim(C.Structure):
    _fields_ = [('pt0', C.c_double * 3), ('z_enp', C.c_double),
                ('y_target', C.c_double), ('z_dir0', C.c_double),
                ('wvl_idx', C.c_int32), ('surf', C.c_int32),
                ('flip', C.c_int32), ('two_d', C.c_int32),
                ('x_target', C.c_double), ('epsfcn', C.c_double)]


class Enp(C.Structure):
    """rox_enp: one wide-angle pupil search (wideangle.find_real_enp)"""
    _fields_ = [('dir0', C.c_double * 3), ('rot', C.c_double * 9),
                ('obj_dist', C.c_double), ('z_enp_0', C.c_double), ('aim_info', C.c_double),
                ('wvl_idx', C.c_int32), ('surf', C.c_int32),
                ('rot_order', C.c_int32), ('check_direction', C.c_int32)]


ENP_FOUND, ENP_NO_CHIEF_RAY, ENP_REFERENCE_RAISES = 0, 1, 3


class SpotSummary(C.Structure):
    """rox_spot_summary: count, sums, sums of squares, min / max of the image-plane hits"""
    _fields_ = [('n', C.c_int64), ('sum', C.c_double * 2), ('sum_sq', C.c_double * 2),
                ('min', C.c_double * 2), ('max', C.c_double * 2)]


SPOT_ROWS, SPOT_PAIRS = 0, 1


class Vig(C.Structure):
    _fields_ = [('fld', Field), ('start_dir', C.c_double * 2), ('unit_dir', C.c_double * 2),
                ('xy', C.c_int32), ('wvl_idx', C.c_int32), ('stop_surf', C.c_int32),
                ('max_iter', C.c_int32)]


class PupilIter(C.Structure):
    _fields_ = [('fld', Field), ('start_r0', C.c_double), ('r_target', C.c_double),
                ('xy', C.c_int32), ('wvl_idx', C.c_int32), ('indx', C.c_int32), ('pad', C.c_int32)]


assert C.sizeof(Aperture) == 40
assert C.sizeof(Vig) == 240
assert C.sizeof(Phase) == 168
assert C.sizeof(Surface) == 576
assert C.sizeof(Wavefront) == 512
assert C.sizeof(Opts) == 568
assert C.sizeof(Field) == 192
assert C.sizeof(Grid) == 48
assert C.sizeof(Out) == 56
assert C.sizeof(Aim) == 80
assert C.sizeof(Enp) == 136

# every symbol include/roxtrace.h declares (checked by tests/test_abi.py) ...
EXPORTS = ('rox_abi_version', 'rox_device_count', 'rox_set_device',
           'rox_last_error', 'rox_system_create', 'rox_system_destroy',
           'rox_system_num_segments', 'rox_trace_rays',
           'rox_trace_pupil_grid', 'rox_trace_pupil_grids', 'rox_trace_pupil_list',
           'rox_aim_chief_rays', 'rox_iterate_ray_raw', 'rox_find_real_enp', 'rox_calc_vignetting',
           'rox_iterate_pupil_rays', 'rox_calc_psf',
           'rox_pin_host_memory', 'rox_unpin_host_memory', 'rox_copy_async', 'rox_synchronize',
           'rox_spot_stats')
# ... and the measurement / self-test helpers of include/roxtrace_diag.h
DIAG_EXPORTS = ('rox_time_pupil_grid', 'rox_selftest_fp64', 'rox_diag_pack_launches')


def declare(lib):
    """attach argtypes/restype for every export of libroxtrace.so"""
    vp, i32, i64, dbl = C.c_void_p, C.c_int32, C.c_int64, C.c_double
    P = C.POINTER
    lib.rox_abi_version.restype = C.c_int
    lib.rox_abi_version.argtypes = []
    lib.rox_device_count.restype = C.c_int
    lib.rox_device_count.argtypes = [P(C.c_int)]
    lib.rox_set_device.restype = C.c_int
    lib.rox_set_device.argtypes = [C.c_int]
    lib.rox_last_error.restype = C.c_char_p
    lib.rox_last_error.argtypes = []
    lib.rox_system_create.restype = C.c_int
    lib.rox_system_create.argtypes = [P(Surface), i32, vp, vp, i32, P(vp)]
    lib.rox_system_destroy.restype = C.c_int
    lib.rox_system_destroy.argtypes = [vp]
    lib.rox_system_num_segments.restype = C.c_int
    lib.rox_system_num_segments.argtypes = [vp, C.c_uint32, P(i32)]
    lib.rox_trace_rays.restype = C.c_int
    lib.rox_trace_rays.argtypes = [vp, i64, vp, vp, vp, i32, P(Opts), P(Out), vp]
    lib.rox_trace_pupil_grid.restype = C.c_int
    lib.rox_trace_pupil_grid.argtypes = [vp, P(Field), P(Grid), i32, P(Opts),
                                         P(Out), vp]
    lib.rox_trace_pupil_grids.restype = C.c_int
    lib.rox_trace_pupil_grids.argtypes = [vp, i32, P(Field), P(i32), P(Grid), P(Opts),
                                          P(Out), vp]
    lib.rox_trace_pupil_list.restype = C.c_int
    lib.rox_trace_pupil_list.argtypes = [vp, P(Field), i64, vp, vp, i32,
                                         P(Opts), P(Out), vp]
    lib.rox_aim_chief_rays.restype = C.c_int
    lib.rox_aim_chief_rays.argtypes = [vp, i32, P(Aim), dbl, vp, vp, vp]
    lib.rox_iterate_ray_raw.restype = C.c_int
    lib.rox_iterate_ray_raw.argtypes = [vp, i32, P(Aim), dbl, vp, vp, vp, vp, vp]
    lib.rox_find_real_enp.restype = C.c_int
    lib.rox_find_real_enp.argtypes = [vp, i32, P(Enp), dbl, vp, vp, vp]
    lib.rox_calc_vignetting.restype = C.c_int
    lib.rox_calc_vignetting.argtypes = [vp, i32, P(Vig), dbl, vp, vp, vp]
    lib.rox_iterate_pupil_rays.restype = C.c_int
    lib.rox_iterate_pupil_rays.argtypes = [vp, i32, P(PupilIter), dbl, vp, vp]
    lib.rox_calc_psf.restype = C.c_int
    lib.rox_calc_psf.argtypes = [vp, i32, i32, vp, C.c_uint32, vp]
    lib.rox_pin_host_memory.restype = C.c_int
    lib.rox_pin_host_memory.argtypes = [vp, C.c_size_t, P(vp)]
    lib.rox_unpin_host_memory.restype = C.c_int
    lib.rox_unpin_host_memory.argtypes = [vp]
    lib.rox_copy_async.restype = C.c_int
    lib.rox_copy_async.argtypes = [vp, vp, C.c_size_t, vp]
    lib.rox_synchronize.restype = C.c_int
    lib.rox_synchronize.argtypes = [vp]
    lib.rox_spot_stats.restype = C.c_int
    lib.rox_spot_stats.argtypes = [vp, i64, vp, vp, i64, i32, vp, i32, vp, i32, P(SpotSummary), vp, vp]
    lib.rox_time_pupil_grid.restype = C.c_int
    lib.rox_time_pupil_grid.argtypes = [vp, P(Field), P(Grid), i32, P(Opts),
                                        P(Out), vp, i32, P(dbl)]
    lib.rox_selftest_fp64.restype = C.c_int
    lib.rox_selftest_fp64.argtypes = [C.c_uint64, C.c_uint64, P(C.c_uint64)]
    lib.rox_diag_pack_launches.restype = C.c_int
    lib.rox_diag_pack_launches.argtypes = [P(C.c_uint64)]
    return lib
