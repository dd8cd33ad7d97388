/* roxtrace.h -- C ABI of libroxtrace.so, the MI355X (gfx950) sequential
 * real-ray trace engine that sits behind ray-optics' `raytr` hot path.
 *
 * The reference (mjhoptics/ray-optics) is pure Python and has no FFI; the
 * seams this ABI replaces are module-level functions:
 *
 *   rox_trace_rays         <- rayoptics/raytr/raytrace.py:51-80   trace()
 *                             rayoptics/raytr/raytrace.py:83-264  trace_raw()
 *                             rayoptics/raytr/analyses.py:458-510 trace_list_of_rays()
 *   rox_trace_pupil_grid   <- rayoptics/raytr/trace.py:563-605    trace_grid()
 *                             rayoptics/raytr/trace.py:537-560    trace_fan()
 *                             rayoptics/raytr/analyses.py:666-696 trace_ray_grid()
 *                             rayoptics/raytr/analyses.py:212-230 trace_ray_fan()
 *   rox_trace_pupil_grids  <- rayoptics/seq/sequential.py:1058-1114 SequentialModel.trace_grid /
 *                             trace_wavefront (the loop over wavelengths), one launch
 *   rox_trace_pupil_list   <- rayoptics/raytr/analyses.py:437-455 trace_ray_list()
 *                             (each of the above through trace.py:160-221
 *                             trace_safe -> trace.py:253-310 trace_base ->
 *                             opticalspec.py:289-400 ray_start_from_osp,
 *                             opticalspec.py:1339-1353 apply_vignetting)
 *   rox_aim_chief_rays     <- rayoptics/raytr/trace.py:313-415    iterate_ray() (both branches)
 *                             rayoptics/raytr/trace.py:627-640    aim_chief_ray()
 *   rox_iterate_ray_raw    <- rayoptics/raytr/trace.py:866-961    iterate_ray_raw() -- the reverse
 *                             chief ray of wideangle.py:620-665 eval_real_image_ht()
 *   rox_find_real_enp      <- rayoptics/raytr/wideangle.py:86-427 find_real_enp() /
 *                             find_edge() / find_z_enp_on_interval(), :46-83
 *                             enp_z_coordinate()
 *   rox_calc_vignetting    <- rayoptics/raytr/vigcalc.py:233-340, 396-461
 *                             calc_vignetting_for_field() / calc_vignetted_ray() /
 *                             iterate_pupil_ray()
 *   rox_iterate_pupil_rays <- rayoptics/raytr/vigcalc.py:396-461 iterate_pupil_ray() as
 *                             vigcalc.set_pupil (:123-230) calls it
 *   rox_system_create      <- rayoptics/seq/sequential.py:149-202 path()/path_sequence()
 *                             (the per-wavelength (Intfc, Gap, Tfrm, Indx, Zdir)
 *                             list flattened into one POD table)
 *
 * All arithmetic is IEEE binary64.  Plain pointers and sizes only; no torch or
 * numpy types appear in any signature.  The Python binding a ray-optics
 * maintainer would add is a ctypes stub: see INTEGRATION.md.  Timing and
 * self-test helpers used by bench.py and tests/ are declared separately in
 * roxtrace_diag.h: they are not part of the drop-in boundary.
 *
 * Ownership: the caller owns every input/output buffer; the library owns only
 * the rox_system handle.  Every entry point returns 0 on success and a
 * negative rox_err on failure, with a message retrievable through
 * rox_last_error().  Per-ray trace failures are *not* errors: they are
 * reported in rox_out.status / rox_out.fail_surf.
 */
#ifndef ROXTRACE_H
#define ROXTRACE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* libroxtrace.so is built with -fvisibility=hidden: the entry points declared between this
 * push and the pop at the end of the header are the library's whole dynamic symbol table
 * (tests/test_abi.py compares `nm -D` with the declarations). */
#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility push(default)
#endif

/* ABI history: 4 = packed hits appended over launches (ROX_HITS_APPEND), rox_pin_host_memory,
 * rox_aim carries both branches of iterate_ray;  5 = rox_trace_pupil_grids (several grids, one
 * launch), rox_find_real_enp / rox_enp (the wide-angle pupil search);  6 = rox_out.ld is the
 * capacity of seg in pairs for ROX_OUT_HITS_COMPACT (overflow: n_hits < 0), rox_copy_async,
 * rox_iterate_ray_raw, rox_iterate_pupil_rays;  7 = rox_synchronize (a binding without the
 * HIP runtime can wait for the asynchronous entries);  8 = ROX_FAST_FP64 (tolerance-mode
 * kernels for the reduced-output modes), rox_surface.flags validated by rox_system_create,
 * the dynamic symbol table is exactly this header's (+ roxtrace_diag.h's) functions and
 * DT_SONAME is libroxtrace.so.8, rox_spot_stats.
 * rox_abi_version() of the library must equal the header a binding was written against. */
#define ROX_ABI_VERSION 8
#define ROX_MAX_COEF 10   /* EvenPolynomial r^2..r^20 / RadialPolynomial r^1..r^10 */
#define ROX_MAX_AP 4      /* clear apertures per surface carried in the table */
#define ROX_SEG_DOUBLES 10 /* p[3], d[3], dst, nrml[3]  (model_constants.py:31) */

/* interface interact_mode (raytrace.py:211-221) */
enum { ROX_TRANSMIT = 0, ROX_REFLECT = 1, ROX_DUMMY = 2, ROX_PHANTOM = 3 };
/* surface profile kinds (rayoptics/elem/profiles.py).  ROX_THINLENS is not a
 * profile in the reference but the ThinLens interface's own intersect/normal
 * (rayoptics/oprops/thinlens.py:128-136): s = -p.z/d.z, normal (0,0,1). */
enum { ROX_SPHERICAL = 0, ROX_CONIC = 1, ROX_EVENPOLY = 2, ROX_RADIALPOLY = 3,
       ROX_YTOROID = 4, ROX_XTOROID = 5, ROX_THINLENS = 6 };
/* phase elements (raytrace.py:41-48, 205-210; rayoptics/oprops/doe.py) */
enum { ROX_PH_NONE = 0,
       ROX_PH_GRATING = 1,   /* DiffractionGrating.phase_ludwig, doe.py:124-175 */
       ROX_PH_DOE_RADIAL = 2,/* DiffractiveElement.phase with radial_phase_fct,
                                doe.py:28-54, 272-323                           */
       ROX_PH_HOLOGRAM = 3 };/* HolographicElement.phase, doe.py:375-397 (the
                                phase element of every ThinLens)               */
/* clear-aperture kinds (rayoptics/elem/surface.py:398-494).  Elliptical has
 * no point_inside() in the reference, so it returns None and the ray is
 * always blocked; ROX_AP_ALWAYS_BLOCK reproduces that. */
enum { ROX_AP_CIRCULAR = 0, ROX_AP_RECTANGULAR = 1, ROX_AP_ALWAYS_BLOCK = 2 };
/* per-ray status (rayoptics/raytr/traceerror.py:12-52) */
enum { ROX_OK = 0, ROX_MISSED_SURFACE = 1, ROX_TIR = 2, ROX_BLOCKED = 3,
       ROX_EVANESCENT = 4 };
/* what rox_out.seg receives */
enum { ROX_OUT_FULL = 0,  /* seg[n_seg][10][ld]: the whole RayPkg.ray          */
       ROX_OUT_LAST = 1,  /* seg[10][ld]: ray[-1] only (trace_safe 'last')      */
       ROX_OUT_HITS = 2,  /* seg[2][ld]: SpotDiagramFigure's `spot` filter,
                             (ray[-1].p + (foc/ray[-1].d[2])*ray[-1].d - image_pt).xy
                             (rayoptics/mpl/axisarrayfigure.py:229-238)         */
       ROX_OUT_OPD = 3,   /* seg[1][ld]: wave_abr_full_calc_finite_pup, the OPD of
                             the ray w.r.t. the chief ray on a finite reference
                             sphere, system units (rayoptics/raytr/waveabr.py:256-307);
                             constants in rox_opts.wf; on an infinite reference
                             sphere wave_abr_full_calc_inf_ref (:356-424)       */
       ROX_OUT_HITS_COMPACT = 4,
                          /* seg[n_hits][2]: the HITS pair (x, y) of the rays that
                             reach the image only, interleaved, packed in ray order
                             -- the (R_ok, 2) array SequentialModel.trace_grid(...,
                             form='list', append_if_none=False) hands to
                             SpotDiagramFigure (rayoptics/seq/sequential.py:1058-1085,
                             rayoptics/mpl/axisarrayfigure.py:229-263).  The count goes
                             to rox_out.n_hits.  seg / n_hits may be device memory or
                             device-mapped pinned host memory (hipHostMalloc): the
                             kernel then writes the spot straight into host memory.
                             op / fail_surf / pupil are not written in this mode;
                             status is optional.                                */
       ROX_OUT_FAN = 5 }; /* seg[3][ld]: what analyses.eval_fan / focus_fan compute per
                             ray of a RayFan (rayoptics/raytr/analyses.py:233-274,
                             317-345): rows 0, 1 = the HITS pair (transverse
                             aberration at rox_opts.foc w.r.t. rox_opts.image_pt),
                             row 2 = the OPD as ROX_OUT_OPD gives it (rox_opts.wf) */
/* rox_opts.flags */
enum { ROX_CHECK_APERTURES = 1u,     /* raytrace.py:198-202                    */
       ROX_INTERSECT_OBJ = 2u,       /* raytrace.py:147-154                    */
       ROX_FILTER_PHANTOMS = 4u,     /* raytrace.py:185-188                    */
       ROX_APPLY_VIGNETTING = 8u,    /* trace.py:298-300 (pupil entries only)  */
       ROX_HOST_POINTERS = 16u };    /* every buffer is ordinary host memory: the
                                        library stages it itself (small batches
                                        through a device-mapped pinned block the
                                        kernel reads and writes directly, large
                                        ones through HBM) and the call returns
                                        when the results are in place.  seg
                                        slots the trace does not produce come
                                        back as NaN in this mode                */
/* ROX_HITS_APPEND (HITS_COMPACT only): the pairs of this call go *behind* the
 * *rox_out.n_hits pairs already in rox_out.seg, and n_hits becomes the new total --
 * a running count kept on the device side, so that several grids (the pupil-row
 * blocks of one rank of a sharded spot diagram, every (field, wavelength) of a
 * figure) pack into one buffer with no host round trip between the launches.
 * rox_out.ld is the capacity of seg in pairs: nothing is stored at or beyond it, and a
 * call that needed more room leaves the NEGATED pair count it needed in n_hits (later
 * appending calls keep it negative).                                          */
#define ROX_HITS_APPEND 32u
/* ROX_FAST_FP64: tolerance mode.  The default kernels reproduce the reference's NumPy path bit
 * for bit; with this flag the caller accepts ray intercepts, directions, optical paths and OPDs
 * within 1e-10 * max(1, |reference value|) of it -- the tolerance ray-optics users compare ray
 * traces at -- and the reduced-output modes (LAST, HITS, HITS_COMPACT, OPD, FAN), which are bound
 * by instruction issue, run on kernels that are still IEEE binary64 but use reciprocal /
 * reciprocal-square-root seeds with Newton-Raphson refinement, fused multiply-adds and Horner
 * series instead of NumPy's operation order (csrc/rox_device.hpp, "tolerance mode"; observed
 * deviation <= 1e-12).  A ray whose miss / total-internal-reflection radicand or aperture margin
 * lies within rounding of zero may be classified the other way.  The flag is a permission: the
 * library takes it where it buys time and answers bit-exactly (which is within any tolerance)
 * elsewhere -- ROX_OUT_FULL packets come from the tolerance-mode kernels only for systems
 * made mostly of aspheres, whose FULL launches are bound by arithmetic too (every segment
 * within the same bar, partial records of failed rays included); FULL launches of other
 * systems (bound by their stores), FULL with ROX_FILTER_PHANTOMS and the search entries
 * (rox_aim_chief_rays, rox_find_real_enp, rox_calc_vignetting, rox_iterate_*) stay
 * bit-exact.  In a rox_trace_pupil_grids call every item must carry the same setting.   */
#define ROX_FAST_FP64 64u
/* rox_surface.rt_order: NumPy hands `rt.dot(v)` to OpenBLAS dgemv, whose FMA
 * chain runs over the columns in a different order for an F-ordered rt (the
 * transpose view compute_local_transforms makes, rayoptics/elem/transform.py:86)
 * and a C-ordered one (np.identity, or the transpose of a transpose that
 * 'dec and return' decenters produce).  Probed on OpenBLAS 0.3.29/Haswell:
 *   F: y_i = fma(a_i2,x2, fma(a_i1,x1, fma(a_i0,x0, 0)))
 *   C: y_i = fma(a_i2,x2, fma(a_i0,x0, fma(a_i1,x1, 0)))                       */
enum { ROX_RT_F_ORDER = 0, ROX_RT_C_ORDER = 1 };
/* rox_wavefront.kind */
enum { ROX_WF_FINITE = 0, ROX_WF_INF_FULL = 1, ROX_WF_INF_SPLIT = 2 };
/* rox_grid.kind */
enum { ROX_GRID_PRODUCT = 0, /* trace_grid: ray r=(i*num+j), x_i outer, y_j inner */
       ROX_GRID_FAN = 1 };   /* trace_fan: ray r at (x_r, y_r), num rays          */

typedef enum { ROX_E_OK = 0, ROX_E_ARG = -1, ROX_E_HIP = -2, ROX_E_NOMEM = -3,
               ROX_E_UNSUPPORTED = -4, ROX_E_NO_DEVICE = -5 } rox_err;

typedef struct rox_aperture {
    int32_t kind;            /* ROX_AP_* */
    int32_t is_obscuration;  /* surface.py:419,457: result inverted            */
    double x_offset, y_offset;   /* surface.py:391-394 (rotation never applied) */
    double a;                /* radius | x_half_width                          */
    double b;                /* unused | y_half_width                          */
} rox_aperture;              /* 40 bytes */

/* One row per interface, object and image included.  Row i carries the
 * transform and gap data *from* interface i *to* interface i+1, exactly as
 * the reference's path tuple does (sequential.py:167-202). */
/* Phase element attached to an interface (ifc.phase_element).  Which fields
 * are read depends on kind:
 *   GRATING    a = grating_normal, order, spacing_nm = _grating_spacing_nm
 *   DOE_RADIAL coefs[ncoef] = coefficients (r^2, r^4, ...), order, ref_wl
 *   HOLOGRAM   a = ref_pt, b = obj_pt, flags bit0 ref_virtual, bit1 obj_virtual,
 *              ref_wl                                                          */
typedef struct rox_phase {
    int32_t kind;            /* ROX_PH_*                                       */
    int32_t ncoef;
    int32_t flags;
    int32_t reserved;
    double order;
    double ref_wl;           /* nm                                             */
    double spacing_nm;
    double a[3];
    double b[3];
    double coefs[ROX_MAX_COEF];
} rox_phase;                 /* 168 bytes */

/* rox_surface.flags.  ROX_SURF_CV_INT_ZERO: the reference model holds this Spherical / Conic
 * curvature as the Python *integer* 0 (its own double Gauss data does,
 * rayoptics/raytr/tests/ag_dblgauss_s.py).  `-self.cv*p[0]` in Spherical/Conic.df
 * (profiles.py:360-362, 605-609) is then `0 * x` -- +0 for x > 0 -- where the float 0.0 gives
 * `-0.0 * x` = -0; intersect() and sag() see +0.0 either way.  cv itself is 0.0.            */
#define ROX_SURF_CV_INT_ZERO 1

typedef struct rox_surface {
    int32_t mode;            /* ROX_TRANSMIT...                                */
    int32_t profile;         /* ROX_SPHERICAL...                               */
    int32_t ncoef;           /* max_nonzero_coef (profiles.py:827-832)         */
    int32_t n_ap;            /* len(clear_apertures); 0 -> max_aperture test   */
    int32_t rt_order;        /* summation order of rt.dot(v), see ROX_RT_*      */
    int32_t flags;           /* ROX_SURF_* (0 from callers that predate it)     */
    double cv;               /* vertex curvature                               */
    double cc;               /* conic constant                                 */
    double ec;               /* cc + 1.0 as the reference evaluates it         */
    double cR;               /* Y/XToroid sweep curvature (profiles.py:1317-1437) */
    double coefs[ROX_MAX_COEF];
    double rt[9];            /* lcl_tfrms[i][0], row-major (already R^T)       */
    double t[3];             /* lcl_tfrms[i][1]                                */
    double z_dir;            /* z_dir[i] of the gap after this interface       */
    double max_aperture;     /* interface.py:113-122                           */
    rox_aperture ap[ROX_MAX_AP];
    rox_phase ph;            /* kind == ROX_PH_NONE: refract / reflect as usual */
} rox_surface;               /* 576 bytes */

/* Per (field, wavelength, focus) constants of the OPD calculation: the chief
 * ray package and reference sphere that trace.setup_pupil_coords() leaves in
 * fld.chief_ray / fld.ref_sphere (rayoptics/raytr/trace.py:608-624,
 * rayoptics/raytr/waveabr.py:23-76). */
typedef struct rox_wavefront {
    double cr1_p[3];         /* cr_ray[1][mc.p]                                 */
    double cr0_d[3];         /* cr_ray[0][mc.d]                                 */
    double crk_p[3];         /* cr_ray[-2][mc.p]                                */
    double crk_d[3];         /* cr_ray[-2][mc.d]                                */
    double cr_op;            /* chief-ray optical path                          */
    double cr_exp_pt[3];     /* cr_exp_seg[0]                                   */
    double cr_exp_dist;      /* cr_exp_seg[2]                                   */
    double ref_dir[3];       /* ref_sphere[1]                                   */
    double ref_radius;       /* ref_sphere[2]                                   */
    double n_obj, n_img;     /* abs(fod.n_obj), abs(fod.n_img)                  */
    double sign_soln;        /* -1 if ref_dir[2]*cr.ray[-1].d[2] < 0 else +1    */
    /* transform_after_surface(ifcs[-2], .) (rayoptics/elem/transform.py:234-258):
     * 0 = identity, 1 = p - t, 2 = rt.dot(p - t), rt.dot(d) */
    int32_t after_kind;
    int32_t after_order;     /* ROX_RT_* of after_rt                            */
    double after_rt[9];
    double after_t[3];
    /* Infinite reference sphere (is_kinda_big(ref_radius), waveabr.py:213-221):
     * wave_abr_full_calc_inf_ref (waveabr.py:356-424, kind ROX_WF_INF_FULL) or its
     * pre-calc / calc split (waveabr.py:427-488, ROX_WF_INF_SPLIT: the same
     * quantities, the final sum associated differently).  Chief-ray-only terms
     * are formed on the host as the reference forms them.                      */
    int32_t kind;            /* ROX_WF_*                                        */
    int32_t last_kind;       /* lcl_tfrm_last: 0 = None, 1 = (rt, t)            */
    int32_t last_order;      /* ROX_RT_* of last_rt                             */
    int32_t reserved;
    double last_rt[9];       /* seq_model.lcl_tfrms[-2][0]                      */
    double last_t[3];
    double cr_last_p[3];     /* cr_ray[-1][mc.p]                                */
    double cr_last_d[3];     /* cr_ray[-1][mc.d]                                */
    double d_cr_b4[3];       /* rt.dot(cr_ray[-2][mc.d])                        */
    double v_be;             /* cr_op + ray_dist_to_perp_from_origin(cr b4)     */
    double image_pt[3];      /* ref_sphere[0]                                   */
} rox_wavefront;

typedef struct rox_opts {
    uint32_t flags;          /* ROX_CHECK_APERTURES | ...                      */
    int32_t out_mode;        /* ROX_OUT_*                                      */
    int32_t first_surf;      /* raytrace.py:118 (trace() passes 1)             */
    int32_t last_surf;       /* raytrace.py:119; <0 means None (trace(): N-2)  */
    double eps;              /* Newton tolerance, raytrace.py:83 (1e-12)       */
    double fuzz;             /* pt_inside_fuzz, surface.py:198 (1e-5)          */
    double foc;              /* HITS only: defocus                             */
    double image_pt[2];      /* HITS only: fld.ref_sphere[0][:2]               */
    rox_wavefront wf;        /* OPD only                                       */
} rox_opts;

/* Per-field constants of ray_start_from_osp (opticalspec.py:289-400), one
 * kind per branch of that function, and of apply_vignetting
 * (opticalspec.py:1339-1353).  With pupil = (px, py) after vignetting:
 *   EPD       :358-366  pt1 = (eprad*px + aim[0], eprad*py + aim[1], z_enp);
 *                       dir0 = normalize(pt1 - pt0)
 *   EPD_WIDE  :340-356  pt1 = rot.(eprad*px, eprad*py, 0), pt1.z -= z_enp
 *                       (z_enp carries obj2enp_dist); dir0 = normalize(pt1 - pt0);
 *                       callers clear ROX_INTERSECT_OBJ (trace.py:302-303) and
 *                       the sign flip of trace.py:304-308 is skipped
 *   AIM_PT    :334-337  pupil_type 'aim pt': pt1 = (px, py, z_enp)
 *   NA        :372-376  pupil_dir = eprad*(px, py)  (eprad carries na/n)
 *   FNO       :377-384  slope = eprad (= -1/(2 fno)); hypt = sqrt(1 + (px*slope)^2
 *                       + (py*slope)^2); pupil_dir = slope*(px, py)/hypt
 *   AIM_DIR   :369-371  pupil_type 'aim dir': dir_tot = (px, py)
 *   the angular kinds (NA, FNO, AIM_DIR) finish with dir_tot = pupil_dir + cr_dir,
 *   dir0 = (dir_tot, sqrt(1 - dir_tot.dir_tot))  (:386-398)                      */
enum { ROX_FLD_EPD = 0, ROX_FLD_EPD_WIDE = 1, ROX_FLD_AIM_PT = 2, ROX_FLD_NA = 3,
       ROX_FLD_FNO = 4, ROX_FLD_AIM_DIR = 5 };

typedef struct rox_field {
    double pt0[3];           /* EPD: obj2enp_dist*[d0x/d0z, d0y/d0z, 0]; else p0 */
    double aim[2];           /* fld.aim_info or (0,0)                          */
    double eprad;            /* pupil_value/2 (see the kinds above)            */
    double z_enp;            /* fod.obj_dist + fod.enp_dist (pt1[2])           */
    double vlx, vux, vly, vuy;   /* opticalspec.py:1339-1353                   */
    double z_dir0;           /* seq_model.z_dir[0] (trace.py:307); 0 = wide-angle
                                model: dir0 is never flipped (trace.py:302-303;
                                callers also clear ROX_INTERSECT_OBJ)            */
    int32_t kind;            /* ROX_FLD_*                                      */
    int32_t rot_order;       /* ROX_RT_* of rot (np.matmul -> dgemv)           */
    double rot[9];           /* EPD_WIDE: rot_v1_into_v2(d0, z), row-major     */
    double cr_dir[2];        /* angular kinds: chief-ray direction (:386-392)  */
} rox_field;                 /* 192 bytes */

typedef struct rox_grid {
    double start[2];         /* grid_rng[0]                                    */
    double stop[2];          /* grid_rng[1]                                    */
    int32_t num;             /* grid_rng[2]                                    */
    int32_t kind;            /* ROX_GRID_PRODUCT | ROX_GRID_FAN                */
    int32_t row_begin;       /* PRODUCT only: trace pupil rows (x index i)     */
    int32_t row_count;       /*   [row_begin, row_begin+row_count); 0 = all    */
} rox_grid;                  /* (row blocks are the multi-GPU sharding unit)   */

typedef struct rox_out {
    double *seg;             /* see ROX_OUT_*; may be NULL only if unused      */
    double *op;              /* [ld] op_delta (on failure: opl, raytrace.py:236) or NULL */
    uint8_t *status;         /* [ld] ROX_OK...                                 */
    int16_t *fail_surf;      /* [ld] surface index at which the ray failed, -1 if ok; or NULL */
    double *pupil;           /* [2][ld] pupil coords after vignetting (pupil entries) or NULL */
    int64_t ld;              /* ray-axis leading dimension, >= number of rays;
                                ROX_OUT_HITS_COMPACT: capacity of seg in (x, y) pairs
                                (>= number of rays unless ROX_HITS_APPEND)         */
    int64_t *n_hits;         /* HITS_COMPACT only: receives the number of (x, y)
                                pairs written to seg                          */
} rox_out;

typedef struct rox_system rox_system;

/* library / device ------------------------------------------------------- */
int rox_abi_version(void);
int rox_device_count(int *count);
int rox_set_device(int device);
const char *rox_last_error(void);

/* Host memory the kernels write directly (ROX_OUT_HITS_COMPACT's seg / n_hits may be
 * such memory).  rox_pin_host_memory page-locks [p, p + bytes) -- ordinary or
 * MAP_SHARED memory, e.g. a segment several ranks of a node map (the multi-GPU
 * spot diagram: every rank's kernel writes its packed hits over its own PCIe link
 * into the consumer's address space) -- and returns the pointer device code uses
 * for it.  The registration lasts until rox_unpin_host_memory(p).               */
int rox_pin_host_memory(void *p, size_t bytes, void **device_ptr);
int rox_unpin_host_memory(void *p);
/* An asynchronous copy of `bytes` bytes in the order of `stream` between any two of device
 * memory, page-locked host memory and memory registered with rox_pin_host_memory (copy
 * engine; no kernel).  The pipelined multi-GPU spot diagram moves a row block's packed pairs
 * to the consumer's host memory with it while the next block is being traced.          */
int rox_copy_async(void *dst, const void *src, size_t bytes, void *stream);
/* Wait until everything enqueued on `stream` (a hipStream_t, NULL = the default stream) has
 * finished: the trace entries with device / pinned pointers and rox_copy_async return as soon
 * as their work is enqueued.  For callers that do not link the HIP runtime themselves
 * (examples/spot_diagram.c); the Python engine waits through torch's stream instead.        */
int rox_synchronize(void *stream);

/* system table ----------------------------------------------------------- */
/* rows[n_ifcs]; n_table[n_wvls][n_ifcs], n_table[w][i] = refractive index of
 * the gap after interface i at wavelength w (unsigned, sequential.py:649-655;
 * the last column is unused); wvls[n_wvls] = the wavelengths in nm (read by
 * phase elements only; may be NULL when no row carries one).  The handle's
 * table is immutable: a model edit means a new handle (mirrors
 * path_sequence.cache_clear(), sequential.py:666-668).
 *
 * Threading: launches on different HIP streams and from different host threads
 * may share one handle -- per-launch scratch (cached pupil axes, compaction
 * state) is kept per stream behind a mutex.  Launches on one stream run in
 * stream order as usual.  The scratch of a stream (a few KB, plus the staging
 * arena of ROX_HOST_POINTERS calls made on it) lives until rox_system_destroy:
 * use a bounded set of streams per handle, and do not issue device-pointer
 * launches on one stream from two host threads at once (ROX_HOST_POINTERS
 * calls, which are synchronous, take turns per stream by themselves). */
int rox_system_create(const rox_surface *rows, int32_t n_ifcs,
                      const double *n_table, const double *wvls, int32_t n_wvls,
                      rox_system **out_sys);
int rox_system_destroy(rox_system *sys);
/* number of segments a FULL packet holds (n_ifcs minus filtered phantoms) */
int rox_system_num_segments(const rox_system *sys, uint32_t flags,
                            int32_t *n_seg);

/* vignetting search ------------------------------------------------------ */
/* One problem per (field, pupil direction): vigcalc.calc_vignetted_ray
 * (rayoptics/raytr/vigcalc.py:259-340) -- trace the pupil-edge ray with aperture
 * checks (pt_inside_fuzz 1e-4); where it is clipped, iterate the pupil coordinate
 * (iterate_pupil_ray, vigcalc.py:396-461: scipy's secant, tol 1e-6, on rays traced
 * without aperture checks) until the ray grazes that aperture's edge
 * (edge_pt_target, rayoptics/elem/surface.py:210-218, 422-427, 459-464,
 * rayoptics/seq/interface.py:94-111); repeat until no other aperture clips.  The
 * four directions of calc_vignetting_for_field (vigcalc.py:233-256) of every
 * field run as lanes of one launch.  probs / vig / clip are host memory;
 * synchronous. */
typedef struct rox_vig {
    rox_field fld;           /* ray-start constants of the field               */
    double start_dir[2];     /* pupil.pupil_rays[1 + i]                         */
    double unit_dir[2];      /* normalize(start_dir), as NumPy forms it         */
    int32_t xy;              /* i // 2: the pupil axis searched                 */
    int32_t wvl_idx;
    int32_t stop_surf;       /* seq_model.stop_surface, < 0: floating stop      */
    int32_t max_iter;        /* max_iter_count (50)                             */
} rox_vig;                   /* 240 bytes */
int rox_calc_vignetting(rox_system *sys, int32_t n, const rox_vig *probs,
                        double eps, double *vig, int32_t *clip_surf, void *stream);

/* vigcalc.iterate_pupil_ray (rayoptics/raytr/vigcalc.py:396-461) on its own -- what
 * vigcalc.set_pupil (:123-230, from the stop size to the pupil specification) iterates the
 * axial marginal ray through the edge of the stop with: scipy's secant iteration (tol 1e-6)
 * of the pupil coordinate `xy` from start_r0 until the ray, traced without aperture checks,
 * meets interface `indx` at the radius r_target (the reference's "raised before the surface
 * => 0.9 x that trial" rule included).  start_r[n] = the pupil coordinate found. */
typedef struct rox_pupil_iter {
    rox_field fld;           /* ray-start constants of the field               */
    double start_r0;
    double r_target;
    int32_t xy;              /* 0 / 1: the pupil axis iterated                  */
    int32_t wvl_idx;
    int32_t indx;            /* the interface whose edge is the target          */
    int32_t pad;
} rox_pupil_iter;            /* 224 bytes */
int rox_iterate_pupil_rays(rox_system *sys, int32_t n, const rox_pupil_iter *probs,
                           double eps, double *start_r, void *stream);

/* point spread function ---------------------------------------------------
 * analyses.calc_psf(wavefront, ndim, maxdim) (rayoptics/raytr/analyses.py:848-875;
 * callers: analyses.update_psf_data :878-883, mpl/analysisfigure.py:418).
 * opd: [ndim][ndim] OPD in waves as ROX_OUT_OPD / eval_wavefront produce it, NaN =
 * no data; psf: [maxdim][maxdim], the normalised |FFT|^2 of the zero-padded pupil
 * function with the reference's fftshift conventions.  ndim must be even and the
 * block must fit (maxdim/2 + ndim/2 + 1 <= maxdim) -- the shapes for which the
 * reference's slice assignment is valid; maxdim need not be a power of two.
 * Computed as a pruned DFT (two complex GEMMs on the fp64 matrix cores).
 * flags: 0 (device pointers, asynchronous on `stream`) or ROX_HOST_POINTERS. */
int rox_calc_psf(const double *opd, int32_t ndim, int32_t maxdim, double *psf,
                 uint32_t flags, void *stream);

/* trace entries ---------------------------------------------------------- */
/* explicit rays: pt0, dir0 are SoA [3][n_rays] with leading dimension
 * n_rays; wvl_idx is [n_rays] or NULL (then wvl_idx_all is used for all).
 * `stream` is a hipStream_t (NULL = the default stream); the call is
 * asynchronous with respect to the host unless ROX_HOST_POINTERS is set. */
int rox_trace_rays(rox_system *sys, int64_t n_rays,
                   const double *pt0, const double *dir0,
                   const int32_t *wvl_idx, int32_t wvl_idx_all,
                   const rox_opts *opts, const rox_out *out, void *stream);

/* rays generated on the device from a pupil grid or fan */
int rox_trace_pupil_grid(rox_system *sys, const rox_field *fld,
                         const rox_grid *grid, int32_t wvl_idx,
                         const rox_opts *opts, const rox_out *out,
                         void *stream);

/* the same grid for n_grids (field, wavelength) pairs in ONE launch: item i traces
 * flds[i] at wvl_idx[i] with opts[i] into outs[i] -- the per-wavelength loop of
 * SequentialModel.trace_grid / trace_wavefront (rayoptics/seq/sequential.py:1058-1114)
 * and the per-field loops of the figures that call them
 * (rayoptics/mpl/axisarrayfigure.py:213-262).  Every item behaves exactly as the
 * rox_trace_pupil_grid call with the same arguments.  out_mode and
 * ROX_FILTER_PHANTOMS must be the same for all items; device pointers only
 * (no ROX_HOST_POINTERS), no ROX_HITS_APPEND; with ROX_OUT_HITS_COMPACT every item
 * has its own outs[i].seg and outs[i].n_hits. */
int rox_trace_pupil_grids(rox_system *sys, int32_t n_grids, const rox_field *flds,
                          const int32_t *wvl_idx, const rox_grid *grid,
                          const rox_opts *opts, const rox_out *outs, void *stream);

/* rays generated on the device from explicit pupil coordinates px,py[n_rays] */
int rox_trace_pupil_list(rox_system *sys, const rox_field *fld,
                         int64_t n_rays, const double *px, const double *py,
                         int32_t wvl_idx, const rox_opts *opts,
                         const rox_out *out, void *stream);

/* What a spot diagram's consumers reduce the image-plane hits to, computed on the device from
 * the output of a ROX_OUT_HITS launch (layout ROX_SPOT_ROWS: seg = (x, y)[2][ld] + status, rays
 * that did not reach the image skipped) or of a ROX_OUT_HITS_COMPACT launch (ROX_SPOT_PAIRS: seg
 * = interleaved pairs, status NULL, the count read from n_hits on the device when given):
 *   summary  count, sum x, sum y, sum x^2, sum y^2 (centroid, RMS spot radius), min / max of x and
 *            y -- RayGeoPSF.ray_data_bounds (rayoptics/mpl/analysisfigure.py:237-248);
 *   hist     optional [n_x_edges - 1][n_y_edges - 1] counts = numpy.histogram2d(x, y,
 *            bins=[x_edges, y_edges]) as RayGeoPSF.plot's `ax.hist2d` calls it (:250-290):
 *            edges[i] <= v < edges[i + 1], the last bin closed on the right, values outside
 *            dropped.  x_edges / y_edges / summary / hist are HOST pointers; the call is
 *            synchronous (two launches on `stream` + one synchronise).                      */
enum { ROX_SPOT_ROWS = 0, ROX_SPOT_PAIRS = 1 };
typedef struct rox_spot_summary {
    int64_t n;               /* entries counted                                 */
    double sum[2];           /* sum x, sum y                                    */
    double sum_sq[2];        /* sum x^2, sum y^2                                */
    double min[2], max[2];   /* +inf / -inf when n == 0                         */
} rox_spot_summary;          /* 72 bytes */
int rox_spot_stats(const double *seg, int64_t ld, const uint8_t *status, const int64_t *n_hits,
                   int64_t n, int32_t layout, const double *x_edges, int32_t n_x_edges,
                   const double *y_edges, int32_t n_y_edges, rox_spot_summary *summary,
                   uint32_t *hist, void *stream);

/* chief-ray aiming ------------------------------------------------------- */
/* One problem per (field, wavelength): trace.iterate_ray
 * (rayoptics/raytr/trace.py:313-415) as trace.aim_chief_ray calls it
 * (trace.py:627-640, from OpticalSpecs.update_optical_properties,
 * opticalspec.py:263-281): find the aim point (x1, y1) on the paraxial entrance
 * pupil plane such that the ray from pt0 towards (x1, y1, z_enp) meets interface
 * `surf` at (x_target, y_target), every trial ray traced through the whole
 * system (raytrace.trace defaults).  Both branches of the reference run on the
 * device, one lane per problem, all problems in one launch:
 *   two_d = 0  field and target on the y axis (trace.py:376-392): x1 = 0 and y1 by
 *              the secant iteration of scipy.optimize.newton (x0 = 0, tol 1.48e-8,
 *              maxiter 50)
 *   two_d = 1  any other field (trace.py:394-410): scipy.optimize.fsolve from (0, 0)
 *              with epsfcn = 0.0001 * fod.enp_radius -- MINPACK's hybrd (Powell's
 *              hybrid method: forward-difference Jacobian, dog-leg steps, Broyden
 *              updates; xtol 1.49012e-8, maxfev 600, factor 100, internal scaling)
 * probs / aim_xy / result are host memory; synchronous. */
enum { ROX_AIM_CONVERGED = 0,   /* aim = root                                   */
       ROX_AIM_NOT_CONVERGED = 1,/* aim = the last iterate, as iterate_ray keeps it
                                    (results.root; fsolve's x with ier != 1)      */
       ROX_AIM_TRACE_ERROR = 2 };/* a trial ray failed before `surf`: aim = (0, 0) */
typedef struct rox_aim {
    double pt0[3];           /* osp.obj_coords(fld)[0]                         */
    double z_enp;            /* fod.obj_dist + fod.enp_dist                    */
    double y_target;         /* xy_target[1]                                   */
    double z_dir0;           /* seq_model.z_dir[0]                             */
    int32_t wvl_idx;
    int32_t surf;            /* ifcx (the stop surface)                        */
    int32_t flip;            /* not wide angle: dir0 = -dir0 if dir0.z*z_dir0 < 0 */
    int32_t two_d;           /* which branch of iterate_ray (see above)        */
    double x_target;         /* xy_target[0]                                   */
    double epsfcn;           /* two_d: 0.0001 * fod.enp_radius                 */
} rox_aim;                   /* 80 bytes */
/* aim_xy: [n][2] = (x1, y1) per problem */
int rox_aim_chief_rays(rox_system *sys, int32_t n, const rox_aim *probs,
                       double eps, double *aim_xy, int32_t *result, void *stream);

/* rayoptics/raytr/trace.py:866-961 iterate_ray_raw: the same two iterations over an explicit
 * path list -- `sys` is the table of THAT path, e.g. the reversed path along which
 * wideangle.eval_real_image_ht (wideangle.py:620-665; FieldSpec.obj_coords of fields given as
 * real image heights, opticalspec.py:1019-1030, 1059-1070) sends the chief ray back from the
 * image point.  The reference also hands back `rr`, the RayResult of the LAST TRIAL RAY it
 * evaluated (not a ray through the root): last_xy[n][2] = that trial's (x1, y1) on the pupil
 * plane, last_status[n] = its trace status (both NULL: exactly rox_aim_chief_rays).  All the
 * problems of a model -- every field -- are one launch. */
int rox_iterate_ray_raw(rox_system *sys, int32_t n, const rox_aim *probs, double eps,
                        double *aim_xy, int32_t *result, double *last_xy,
                        int32_t *last_status, void *stream);

/* rayoptics/raytr/wideangle.py:86-427 find_real_enp (vselector 'rev1') +
 * find_z_enp_on_interval: the z of the real entrance pupil of a wide-angle
 * field, measured from the first interface -- what trace.aim_chief_ray
 * (trace.py:634-635) stores as fld.aim_info of a wide-angle model.  One lane
 * per (field, wavelength) runs the reference's whole search: the sampled walk
 * from the paraxial pupil, find_edge's bisections, scipy.optimize.newton's
 * secant iteration (tol 1.48e-8, rtol 1e-7) and the brentq fall-back (xtol
 * 2e-12, rtol 1e-7, 100 iterations), every trial ray traced by
 * enp_z_coordinate (wideangle.py:46-83; intersect_obj = False). */
enum { ROX_ENP_FOUND = 0,        /* z_enp as find_real_enp returns it               */
       ROX_ENP_NO_CHIEF_RAY = 1, /* "chief ray trace failed": no ray reaches the stop
                                    centre; z_enp = the last good sample (:283-291) */
       ROX_ENP_REFERENCE_RAISES = 3 }; /* the reference itself raises here (no sample
                                    got through: unpacking None; brentq without a sign
                                    change; a failed last ray indexed at the stop)  */
typedef struct rox_enp {
    double dir0[3];          /* osp.obj_coords(fld)[1]                          */
    double rot[9];           /* rot_v1_into_v2([0,0,1], dir0), row-major         */
    double obj_dist;         /* fod.obj_dist                                    */
    double z_enp_0;          /* fod.enp_dist (the paraxial entrance pupil)      */
    double aim_info;         /* fld.aim_info, NaN if None                       */
    int32_t wvl_idx;
    int32_t surf;            /* stop_idx (1 when the model has no stop surface) */
    int32_t rot_order;       /* ROX_RT_* of rot (np.matmul -> dgemv)            */
    int32_t check_direction; /* find_real_enp_rev1's keyword (True)             */
} rox_enp;                   /* 136 bytes */
/* z_out: [n][2] = (z_enp, z of the last trial ray traced: the `rr` the
 * reference returns beside z_enp is that ray) */
int rox_find_real_enp(rox_system *sys, int32_t n, const rox_enp *probs,
                      double eps, double *z_out, int32_t *result, void *stream);

#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* ROXTRACE_H */
