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
 *   rox_trace_pupil_list   <- rayoptics/raytr/analyses.py:437-455 trace_ray_list()
 *                             (each of the above through trace.py:160-221
 *                             trace_safe -> trace.py:253-310 trace_base ->
 *                             opticalspec.py:289-400 ray_start_from_osp,
 *                             opticalspec.py:1339-1353 apply_vignetting)
 *   rox_system_create      <- rayoptics/seq/sequential.py:149-202 path()/path_sequence()
 *                             (the per-wavelength (Intfc, Gap, Tfrm, Indx, Zdir)
 *                             list flattened into one POD table)
 *
 * All arithmetic is IEEE binary64.  Plain pointers and sizes only; no torch or
 * numpy types appear in any signature.  The Python binding a ray-optics
 * maintainer would add is a ctypes stub: see INTEGRATION.md.
 *
 * Ownership: the caller owns every input/output buffer; the library owns only
 * the rox_system handle.  Every entry point returns 0 on success and a
 * negative rox_err on failure, with a message retrievable through
 * rox_last_error().  Per-ray trace failures are *not* errors: they are
 * reported in rox_out.status / rox_out.fail_surf.
 */
#ifndef ROXTRACE_H
#define ROXTRACE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ROX_ABI_VERSION 2
#define ROX_MAX_COEF 10   /* EvenPolynomial r^2..r^20 / RadialPolynomial r^1..r^10 */
#define ROX_MAX_AP 4      /* clear apertures per surface carried in the table */
#define ROX_SEG_DOUBLES 10 /* p[3], d[3], dst, nrml[3]  (model_constants.py:31) */

/* interface interact_mode (raytrace.py:211-221) */
enum { ROX_TRANSMIT = 0, ROX_REFLECT = 1, ROX_DUMMY = 2, ROX_PHANTOM = 3 };
/* surface profile kinds (rayoptics/elem/profiles.py) */
enum { ROX_SPHERICAL = 0, ROX_CONIC = 1, ROX_EVENPOLY = 2, ROX_RADIALPOLY = 3,
       ROX_YTOROID = 4, ROX_XTOROID = 5 };
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
       ROX_OUT_OPD = 3 }; /* seg[1][ld]: wave_abr_full_calc_finite_pup, the OPD of
                             the ray w.r.t. the chief ray on a finite reference
                             sphere, system units (rayoptics/raytr/waveabr.py:256-307);
                             constants in rox_opts.wf                           */
/* rox_opts.flags */
enum { ROX_CHECK_APERTURES = 1u,     /* raytrace.py:198-202                    */
       ROX_INTERSECT_OBJ = 2u,       /* raytrace.py:147-154                    */
       ROX_FILTER_PHANTOMS = 4u,     /* raytrace.py:185-188                    */
       ROX_APPLY_VIGNETTING = 8u,    /* trace.py:298-300 (pupil entries only)  */
       ROX_HOST_POINTERS = 16u };    /* buffers are host memory: the library
                                        stages them through HBM itself        */
/* rox_surface.rt_order: NumPy hands `rt.dot(v)` to OpenBLAS dgemv, whose FMA
 * chain runs over the columns in a different order for an F-ordered rt (the
 * transpose view compute_local_transforms makes, rayoptics/elem/transform.py:86)
 * and a C-ordered one (np.identity, or the transpose of a transpose that
 * 'dec and return' decenters produce).  Probed on OpenBLAS 0.3.29/Haswell:
 *   F: y_i = fma(a_i2,x2, fma(a_i1,x1, fma(a_i0,x0, 0)))
 *   C: y_i = fma(a_i2,x2, fma(a_i0,x0, fma(a_i1,x1, 0)))                       */
enum { ROX_RT_F_ORDER = 0, ROX_RT_C_ORDER = 1 };
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
typedef struct rox_surface {
    int32_t mode;            /* ROX_TRANSMIT...                                */
    int32_t profile;         /* ROX_SPHERICAL...                               */
    int32_t ncoef;           /* max_nonzero_coef (profiles.py:827-832)         */
    int32_t n_ap;            /* len(clear_apertures); 0 -> max_aperture test   */
    int32_t rt_order;        /* summation order of rt.dot(v), see ROX_RT_*      */
    int32_t reserved;
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
} rox_surface;               /* 408 bytes */

/* Per (field, wavelength, focus) constants of the OPD calculation: the chief
 * ray package and reference sphere that trace.setup_pupil_coords() leaves in
 * fld.chief_ray / fld.ref_sphere (rayoptics/raytr/trace.py:608-624,
 * rayoptics/raytr/waveabr.py:23-76).  Finite reference sphere only: callers keep
 * is_kinda_big(ref_radius) cases on the host (waveabr.py:213-221). */
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

/* Per-field constants of the 'epd', non-wide-angle branch of
 * ray_start_from_osp (opticalspec.py:358-366) and of apply_vignetting. */
typedef struct rox_field {
    double pt0[3];           /* obj2enp_dist*[d0x/d0z, d0y/d0z, 0]             */
    double aim[2];           /* fld.aim_info or (0,0)                          */
    double eprad;            /* pupil_value/2                                  */
    double z_enp;            /* fod.obj_dist + fod.enp_dist (pt1[2])           */
    double vlx, vux, vly, vuy;   /* opticalspec.py:1339-1353                   */
    double z_dir0;           /* seq_model.z_dir[0] (trace.py:307)              */
} rox_field;

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
    int64_t ld;              /* ray-axis leading dimension, >= number of rays  */
} rox_out;

typedef struct rox_system rox_system;

/* library / device ------------------------------------------------------- */
int rox_abi_version(void);
int rox_device_count(int *count);
int rox_set_device(int device);
const char *rox_last_error(void);

/* system table ----------------------------------------------------------- */
/* rows[n_ifcs]; n_table[n_wvls][n_ifcs], n_table[w][i] = refractive index of
 * the gap after interface i at wavelength w (unsigned, sequential.py:649-655;
 * the last column is unused).  The handle is immutable: a model edit means a
 * new handle (mirrors path_sequence.cache_clear(), sequential.py:666-668). */
int rox_system_create(const rox_surface *rows, int32_t n_ifcs,
                      const double *n_table, int32_t n_wvls,
                      rox_system **out_sys);
int rox_system_destroy(rox_system *sys);
/* number of segments a FULL packet holds (n_ifcs minus filtered phantoms) */
int rox_system_num_segments(const rox_system *sys, uint32_t flags,
                            int32_t *n_seg);

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

/* rays generated on the device from explicit pupil coordinates px,py[n_rays] */
int rox_trace_pupil_list(rox_system *sys, const rox_field *fld,
                         int64_t n_rays, const double *px, const double *py,
                         int32_t wvl_idx, const rox_opts *opts,
                         const rox_out *out, void *stream);

/* timing helper for bench.py: runs `launches` back-to-back launches of the
 * pupil-grid kernel on `stream`, bracketed by HIP events recorded on that
 * same stream, and returns the mean kernel duration in milliseconds. */
int rox_time_pupil_grid(rox_system *sys, const rox_field *fld,
                        const rox_grid *grid, int32_t wvl_idx,
                        const rox_opts *opts, const rox_out *out,
                        void *stream, int32_t launches, double *mean_ms);

/* diagnostic: compares the kernels' exponent-band-guarded sqrt / division paths
 * with the plain IEEE operators on n pseudo-random operand sets (whole exponent
 * range, zeros, denormals, inf, nan).  counts[0] = sqrt mismatches, counts[1] =
 * division mismatches (both must be 0), counts[2] = operand sets that took a
 * guarded path. */
int rox_selftest_fp64(uint64_t n, uint64_t seed, uint64_t counts[3]);

#ifdef __cplusplus
}
#endif
#endif /* ROXTRACE_H */
