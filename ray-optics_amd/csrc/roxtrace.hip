// roxtrace.hip -- sequential real-ray trace for gfx950 (MI355X, CDNA4).
//
// Hot path of mjhoptics/ray-optics restated as hand-written HIP (reference
// paths relative to /root/reference/src/):
//   rayoptics/raytr/raytrace.py:83-264   trace_raw   -> trace_kernel (the loop)
//   rayoptics/raytr/raytrace.py:19-38    bend/reflect -> refract(), mirror()
//   rayoptics/elem/profiles.py:310-336   Spherical.intersect  \  quadric_hit()
//   rayoptics/elem/profiles.py:569-593   Conic.intersect      /
//   rayoptics/elem/profiles.py:155-186   intersect_spencer    -> newton_hit()
//   rayoptics/elem/profiles.py:849-885   EvenPolynomial.sag/df \ poly_eval()
//   rayoptics/elem/profiles.py:1070-1113 RadialPolynomial.sag/df/
//   rayoptics/elem/surface.py:198-208, 416-457 point_inside    -> inside_aperture()
//   rayoptics/raytr/opticalspec.py:358-366, 1339-1353; trace.py:298-308
//                                         pupil -> (pt0, dir0) -> launch_ray()
//   rayoptics/raytr/trace.py:563-605, 537-560 grid / fan pupil coordinates
//                                         -> pupil_axes_kernel (repeated +=)
//
// Execution model: one wavefront lane = one ray; 256-thread workgroups
// grid-stride over the ray batch.  Every per-surface parameter is
// wave-uniform: the surface table is staged once per workgroup in LDS and read
// with same-address (broadcast, conflict-free) ds_reads.  Ray packets are SoA
// [segment][component][ray]: each store is 64 lanes x 8 B = 512 B contiguous.
// All arithmetic is IEEE binary64 with the reference's operation order:
// this file is compiled with -ffp-contract=off, NumPy's BLAS dot sites are
// spelled as explicit fma chains (dot3), division and sqrt are the correctly
// rounded ones.  No MFMA: this is 3-vector arithmetic, not a contraction.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "../../include/roxtrace.h"

#pragma clang fp contract(off)

namespace {

// ---- build-time knobs (defaults = the shipped configuration; the others are
// kept for A/B measurement with tools/ab_bench.py) -------------
#ifndef ROX_TABLE_SCALAR     // 1: read the surface table through the scalar cache
#define ROX_TABLE_SCALAR 0   //    (s_load into SGPRs) instead of staging it in LDS
#endif
#ifndef ROX_MIN_WAVES        // __launch_bounds__ second argument (waves per SIMD)
#define ROX_MIN_WAVES 4       // 128 VGPRs: the general (asphere) instance gains 5-8 %, the lean one is unaffected
#endif
#ifndef ROX_STORE_NT         // 1: non-temporal packet stores (measured: FULL 236 us vs 257 us)
#define ROX_STORE_NT 1
#endif
#ifndef ROX_SLIM_FP64        // 1: range-guarded slim sqrt / shared-reciprocal division triples
#define ROX_SLIM_FP64 1       //    (bit-identical to sqrt() and `/`; see slim_* below)
#endif

#ifndef ROX_XCD_SWIZZLE      // 1: workgroups of one XCD (blockIdx % 8) take consecutive ray tiles
#define ROX_XCD_SWIZZLE 0
#endif
#ifndef ROX_BLOCK            // workgroup size (FULL: 128 -> 222, 256 -> 214, 512 -> 208, 1024 -> 212 us)
#define ROX_BLOCK 512
#endif
constexpr int kBlock = ROX_BLOCK;
static_assert(sizeof(rox_surface) == 408, "rox_surface layout");
// Device-side row = the public rox_surface + per-surface values that are the
// same for every ray and are therefore computed once at rox_system_create:
// dcoefs[i] = c_coef_i * coefs[i], the product the df() loops of the polynomial
// profiles form per evaluation (c_coef_i = 2(i+1), or i+1 for RadialPolynomial;
// exact small integers, so the host product has the reference's rounding).
struct dev_surface {
    rox_surface pub;
    double dcoefs[ROX_MAX_COEF];
};
constexpr int kRowDoubles = sizeof(dev_surface) / sizeof(double);   // 61
static_assert(sizeof(rox_aperture) == 40, "rox_aperture layout");

enum { GEN_RAYS = 0, GEN_PUPIL = 1 };
enum { AXIS_LIST = 0, AXIS_PRODUCT = 1 };

struct v3 { double x, y, z; };

#if ROX_TABLE_SCALAR
// constant address space: uniform-address loads become s_load_dwordx{2,4,8}
typedef const __attribute__((address_space(4))) double *tblp;
typedef const __attribute__((address_space(4))) int32_t *tbli;
#else
typedef const double *tblp;     // LDS (generic pointer into __shared__)
typedef const int32_t *tbli;
#endif

// ---------------------------------------------------------------- kernel args
struct TraceArgs {
    const double *rows;        // [N][49] raw rox_surface rows
    const double *n_table;     // [W][N]
    const int32_t *slots;      // [2][N]: slot[s] (-1 = filtered phantom), nslots_before[s]
    int32_t n_ifcs, n_wvls;
    int64_t n_rays;            // rays of this launch (<= 2^28: 32-bit lane byte offsets)
    int64_t ray_base;          // index of this launch's first ray within the batch
    int64_t in_ld;             // batch size = stride of the SoA inputs
    // explicit rays
    const double *pt0, *dir0;  // SoA [3][n_rays]
    const int32_t *wvl_idx;    // per ray or nullptr
    int32_t wvl_idx_all;
    // pupil rays
    const double *px, *py;     // axis / list coordinates
    int32_t axis_kind;         // AXIS_LIST: px[r],py[r]; AXIS_PRODUCT: px[r/num], py[r%num]
    int32_t axis_num;
    int32_t row_begin;         // AXIS_PRODUCT: first pupil row of this launch
    rox_field fld;
    rox_opts opts;
    rox_out out;
};

// ---------------------------------------------------------------- arithmetic
// np.dot / ndarray.dot / np.linalg.norm on float64[3] = OpenBLAS ddot:
// acc = 0; acc = fma(a_i, b_i, acc), i = 0, 1, 2.
__device__ __forceinline__ double dot3(const v3 &a, const v3 &b)
{
    double acc = fma(a.x, b.x, 0.0);
    acc = fma(a.y, b.y, acc);
    return fma(a.z, b.z, acc);
}

// Rt.dot(v) = OpenBLAS dgemv: an fma chain per output row; the column order is
// 0,1,2 for the F-ordered transpose view and 1,0,2 for a C-ordered array
// (include/roxtrace.h ROX_RT_*).  `order` is wave-uniform.
template <class P>
__device__ __forceinline__ v3 rotate(P rt, int order, const v3 &v)
{
    v3 r;
    if (order == ROX_RT_C_ORDER) {
        r.x = fma(rt[2], v.z, fma(rt[0], v.x, fma(rt[1], v.y, 0.0)));
        r.y = fma(rt[5], v.z, fma(rt[3], v.x, fma(rt[4], v.y, 0.0)));
        r.z = fma(rt[8], v.z, fma(rt[6], v.x, fma(rt[7], v.y, 0.0)));
    } else {
        r.x = fma(rt[2], v.z, fma(rt[1], v.y, fma(rt[0], v.x, 0.0)));
        r.y = fma(rt[5], v.z, fma(rt[4], v.y, fma(rt[3], v.x, 0.0)));
        r.z = fma(rt[8], v.z, fma(rt[7], v.y, fma(rt[6], v.x, 0.0)));
    }
    return r;
}


// ---------------------------------------------------------------- slim fp64
// hipcc expands an f64 sqrt into v_rsq_f64 + 9 mul/fma (correctly rounded) wrapped
// in input scaling (v_ldexp x2), a class test and selects; and every f64 `/` into
// v_div_scale x2 + v_rcp_f64 + two Newton steps + q, residual, v_div_fmas,
// v_div_fixup.  The scaling and fix-up only act on operands outside a band of
// exponents (or zero / inf / nan).  Inside the band the functions below execute
// the SAME instruction sequence minus those wrappers, so the results are
// bit-identical; three quotients by one divisor share the refined reciprocal.
// A wave takes the slim path only when every active lane passes the exponent
// test (one wave-uniform branch); otherwise it falls back to the plain operators.
//   band: biased exponent in [640, 1408)  <=>  2^-383 <= |x| < 2^385
//   (v_div_scale scales when exponents differ by >= 768 or the numerator's
//   exponent <= 53; the sqrt expansion scales below 2^-767)
__device__ __forceinline__ bool in_band(double x)
{
    const uint32_t h = (uint32_t)__double2hiint(x) & 0x7fffffffu;
    return (h - 0x28000000u) < 0x30000000u;
}

// numerators may also be exactly +-0 (the sign is restored below)
__device__ __forceinline__ bool in_band_or_zero(double x) { return in_band(x) || x == 0.0; }

// sqrt for x in the band: the expansion of llvm.sqrt.f64 without scaling/selects
__device__ __forceinline__ double sqrt_band(double x)
{
    const double y = __builtin_amdgcn_rsq(x);
    double s = x * y;
    double h = y * 0.5;
    const double r0 = fma(-h, s, 0.5);
    s = fma(s, r0, s);
    h = fma(h, r0, h);
    const double d0 = fma(-s, s, x);
    s = fma(d0, h, s);
    const double d1 = fma(-s, s, x);
    return fma(d1, h, s);
}

__device__ __forceinline__ double slim_sqrt(double x)
{
#if ROX_SLIM_FP64
    if (__all(in_band(x)))
        return sqrt_band(x);
#endif
    return sqrt(x);
}

// refined reciprocal exactly as the division expansion builds it (no scaling)
__device__ __forceinline__ double rcp_band(double b)
{
    double r = __builtin_amdgcn_rcp(b);
    r = fma(r, fma(-b, r, 1.0), r);
    r = fma(r, fma(-b, r, 1.0), r);
    return r;
}

// a / b given r = rcp_band(b): quotient, exact residual, correction; v_div_fixup's
// only effect inside the band is forcing the sign, which also covers a == +-0
__device__ __forceinline__ double div_band(double a, double b, double r)
{
    const double q = a * r;
    const double q1 = fma(fma(-b, q, a), r, q);
    return copysign(q1, q);
}

// (a.x / b, a.y / b, a.z / b)
__device__ __forceinline__ v3 slim_div3(const v3 &a, double b)
{
#if ROX_SLIM_FP64
    if (__all(in_band(b) && in_band_or_zero(a.x) && in_band_or_zero(a.y) && in_band_or_zero(a.z))) {
        const double r = rcp_band(b);
        return v3{div_band(a.x, b, r), div_band(a.y, b, r), div_band(a.z, b, r)};
    }
#endif
    return v3{a.x / b, a.y / b, a.z / b};
}

// misc_math.py:48-54 normalize
__device__ __forceinline__ v3 unit(const v3 &v)
{
    const double len = slim_sqrt(dot3(v, v));
    if (len == 0.0)
        return v;
    return slim_div3(v, len);
}

// raytrace.py:19-30.  false = TIR (math.sqrt ValueError)
__device__ __forceinline__ bool refract(const v3 &d, const v3 &nrm, double n_in,
                                        double n_out, v3 &out)
{
    const double nlen = slim_sqrt(dot3(nrm, nrm));
    const double cosI = dot3(d, nrm) / nlen;
    const double sin2 = 1.0 - cosI * cosI;
    const double rad = n_out * n_out - n_in * n_in * sin2;
    if (rad < 0.0)
        return false;
    const double n_cosIp = copysign(slim_sqrt(rad), cosI);
    const double alpha = n_cosIp - n_in * cosI;
    out = slim_div3(v3{n_in * d.x + alpha * nrm.x, n_in * d.y + alpha * nrm.y,
                       n_in * d.z + alpha * nrm.z}, n_out);
    return true;
}

// raytrace.py:33-38 (not renormalised)
__device__ __forceinline__ v3 mirror(const v3 &d, const v3 &nrm)
{
    const double nlen = slim_sqrt(dot3(nrm, nrm));
    const double cosI = dot3(d, nrm) / nlen;
    const double k = 2.0 * cosI;
    return v3{d.x - k * nrm.x, d.y - k * nrm.y, d.z - k * nrm.z};
}

// profiles.py:321-336 / 580-593: s = cx2 / (z_dir*sqrt(b*b - ax2*cx2) - b)
__device__ __forceinline__ bool quadric_root(double ax2, double cx2, double b,
                                             double z_dir, double &s)
{
    if ((b != 0) || (cx2 != 0) || (ax2 != 0)) {
        const double rad = b * b - ax2 * cx2;
        if (rad < 0.0)
            return false;                       // TraceMissedSurfaceError
        const double den = z_dir * slim_sqrt(rad) - b;
        // np.errstate(divide='raise') -> FloatingPointError -> s = 0 only for a
        // finite non-zero numerator; 0/0 and nan/0 stay NaN
        if (den == 0.0 && cx2 != 0.0 && isfinite(cx2))
            s = 0.0;
        else
            s = cx2 / den;
    } else {
        s = 0.0;
    }
    return true;
}

// Spherical (conic == false) / Conic closed-form intersection
__device__ __forceinline__ bool quadric_hit(bool conic, double cv, double cc, double ec,
                                            const v3 &p, const v3 &d, double z_dir,
                                            double &s, v3 &hit)
{
    double ax2, cx2, b;
    if (!conic) {
        ax2 = cv;
        cx2 = cv * dot3(p, p) - 2 * p.z;
        b = cv * dot3(d, p) - d.z;
    } else {
        ax2 = cv * (1. + cc * d.z * d.z);
        cx2 = cv * (p.x * p.x + p.y * p.y + ec * p.z * p.z) - 2.0 * p.z;
        b = cv * (d.x * p.x + d.y * p.y + ec * d.z * p.z) - d.z;
    }
    if (!quadric_root(ax2, cx2, b, z_dir, s))
        return false;
    hit = v3{p.x + s * d.x, p.y + s * d.y, p.z + s * d.z};
    return true;
}

// One evaluation of f(p) and df(p) for the polynomial aspheres
// (profiles.py:849-885 even, 1070-1113 radial; forward accumulation of the
// powers, not Horner).  Returns false when the sag square root goes negative.
// kind = ROX_EVENPOLY | ROX_RADIALPOLY | ROX_YTOROID | ROX_XTOROID (wave-uniform).
template <bool WANT_F>
__device__ __forceinline__ bool poly_eval(int kind, double cv, double cc1, double ec, double cR,
                                          int ncoef, tblp coefs,
                                          const v3 &p, double &f, v3 &df)
{
    tblp dcoefs = coefs + (offsetof(dev_surface, dcoefs) - offsetof(rox_surface, coefs)) / 8;
    if (kind >= ROX_YTOROID) {
        // profiles.py:1337-1377 YToroid.fY/f/df; XToroid swaps x and y (:1429-1434)
        const bool xt = (kind == ROX_XTOROID);
        const double px = xt ? p.y : p.x, py = xt ? p.x : p.y;
        const double y2 = py * py;
        const double rad = 1. - cc1 * cv * cv * y2;
        if (rad < 0.0)
            return false;
        const double srad = sqrt(rad);
        double z_asp = 0.0, y_pow = y2;
        double e_asp = 0.0, d_pow = 1;
        for (int i = 0; i < ncoef; ++i) {
            z_asp += coefs[i] * y_pow;
            y_pow *= y2;
            e_asp += dcoefs[i] * d_pow;         // (c_coef*coefs[i])*y_pow
            d_pow *= y2;
        }
        const double fY = cv * y2 / (1. + srad) + z_asp;
        if (WANT_F)
            f = p.z - fY - cR * (px * px + p.z * p.z - fY * fY) / 2;
        const double dfdY = cv / srad + e_asp;
        const double Fx = -cR * px;
        const double Fy = (cR * fY - 1) * (dfdY) * py;
        df = xt ? v3{Fy, Fx, 1 - cR * p.z} : v3{Fx, Fy, 1 - cR * p.z};
        return true;
    }
    const bool radial = (kind == ROX_RADIALPOLY);
    const double r2 = p.x * p.x + p.y * p.y;
    double e_tot;
    // sag() and df() take the square root of the same radicand when
    // (cc + 1.0) and ec are the same number (they are, unless a caller fills the
    // table otherwise): evaluate it once.  `same` is wave-uniform.
    const bool same = (cc1 == ec) || radial;
    const double rad_e = 1. - ec * cv * cv * r2;
    if (!radial) {
        double srad_e;
        if (WANT_F) {
            const double rad = 1. - cc1 * cv * cv * r2;     // (cc + 1.0)*cv*cv*r2
            if (rad < 0.0)
                return false;
            const double srad = sqrt(rad);
            srad_e = same ? srad : sqrt(rad_e);
            const double z = cv * r2 / (1. + srad);
            double z_asp = 0.0, r_pow = r2;
            for (int i = 0; i < ncoef; ++i) {
                z_asp += coefs[i] * r_pow;
                r_pow *= r2;
            }
            f = p.z - (z + z_asp);
        } else {
            srad_e = sqrt(rad_e);
        }
        const double e = cv / srad_e;
        double r_pow = 1, e_asp = 0.0;
        for (int i = 0; i < ncoef; ++i) {
            e_asp += dcoefs[i] * r_pow;         // (c_coef*coefs[i])*r_pow
            r_pow *= r2;
        }
        e_tot = e + e_asp;
    } else {
        const double r = sqrt(r2);
        const double srad_e = sqrt(rad_e);      // NaN when negative: caught below
        if (WANT_F) {
            if (rad_e < 0.0)
                return false;
            const double z = cv * r2 / (1. + srad_e);
            double z_asp = 0.0, r_pow = r;
            for (int i = 0; i < ncoef; ++i) {
                z_asp += coefs[i] * r_pow;
                r_pow *= r;
            }
            f = p.z - (z + z_asp);
        }
        const double e = cv / srad_e;
        double e_asp = 0.0;
        double r_pow = (r == 0.0) ? 1.0 : 1 / r;
        for (int i = 0; i < ncoef; ++i) {
            e_asp += dcoefs[i] * r_pow;         // (c_coef*coef)*r_pow
            r_pow *= r;
        }
        e_tot = e + e_asp;
    }
    df = v3{-e_tot * p.x, -e_tot * p.y, 1.0};
    return true;
}

// profiles.py:155-186 Spencer & Murty Newton iteration.  Returns the last
// *evaluated* iterate as the hit point (p0 itself when |s1| <= eps at once).
__device__ __forceinline__ bool newton_hit(int kind, double cv, double cc1, double ec, double cR,
                                           int ncoef, tblp coefs,
                                           const v3 &p0, const v3 &d, double eps,
                                           double &s, v3 &hit, v3 &df)
{
    v3 p = p0;
    double f;
    if (!poly_eval<true>(kind, cv, cc1, ec, cR, ncoef, coefs, p, f, df))
        return false;
    double s1 = -f / dot3(d, df);
    double delta = fabs(s1);
    int iter = 0;
    bool ok = true;
    // one Spencer-Murty step for the lanes that have not converged
    auto step = [&]() {
        p = v3{p0.x + s1 * d.x, p0.y + s1 * d.y, p0.z + s1 * d.z};
        if (!poly_eval<true>(kind, cv, cc1, ec, cR, ncoef, coefs, p, f, df)) {
            ok = false;
            delta = 0.0;            // leave the iteration; the caller reports the miss
            return;
        }
        const double s2 = s1 - f / dot3(d, df);
        delta = fabs(s2 - s1);
        s1 = s2;
        ++iter;
    };
    // measured on the reference's even-asphere zoom: 2 steps 20 %, 3 steps 73 %,
    // 4 steps 6 %, more < 1 % -- four steps straight-line and
    // predicated per lane, then the residual loop (cap 1000 as in the reference)
#pragma unroll
    for (int u = 0; u < 4; ++u)
        if (delta > eps)
            step();
    while (delta > eps && iter < 1000)
        step();
    if (!ok)
        return false;
    s = s1;
    hit = p;        // df already holds df(hit): normal() re-evaluates the same expression
    return true;
}

// surface.py:198-208 (+ interface.py:113-122, surface.py:416-419, 453-457)
__device__ __forceinline__ bool inside_aperture(tblp row, int n_ap, double x,
                                                double y, double fuzz)
{
    if (n_ap > 0) {
        tblp ap = row + (offsetof(rox_surface, ap) / sizeof(double));
        for (int k = 0; k < n_ap; ++k, ap += sizeof(rox_aperture) / sizeof(double)) {
            const int2 ki{((tbli)ap)[0], ((tbli)ap)[1]};            // kind, is_obscuration
            const double xx = x - ap[1];
            const double yy = y - ap[2];
            bool ans;
            if (ki.x == ROX_AP_CIRCULAR)
                ans = sqrt(xx * xx + yy * yy) <= ap[3] + fuzz;
            else if (ki.x == ROX_AP_RECTANGULAR)
                ans = (fabs(xx) <= ap[3] + fuzz) && (fabs(yy) <= ap[4] + fuzz);
            else
                return false;               // Elliptical: point_inside() returns None
            if (ki.y)
                ans = !ans;
            if (!ans)
                return false;
        }
        return true;
    }
    return sqrt(x * x + y * y) <= row[offsetof(rox_surface, max_aperture) / sizeof(double)] + fuzz;
}

// ------------------------------------------------------------------ OPD
// waveabr.py:117-132 eic_distance
__device__ __forceinline__ double eic_distance(const v3 &p, const v3 &d, const double *p0,
                                               const double *d0)
{
    const v3 sd{d.x + d0[0], d.y + d0[1], d.z + d0[2]};
    const v3 dp{p.x - p0[0], p.y - p0[1], p.z - p0[2]};
    return dot3(sd, dp) / (1. + dot3(d, v3{d0[0], d0[1], d0[2]}));
}

// waveabr.py:256-307 wave_abr_full_calc_finite_pup (+ transform.py:234-258)
__device__ __forceinline__ double wave_abr_finite_pup(const rox_wavefront &w, const v3 &ray1_p,
                                                      const v3 &ray0_d, const v3 &rayk_p,
                                                      const v3 &rayk_d, double ray_op)
{
    const double e1 = eic_distance(ray1_p, ray0_d, w.cr1_p, w.cr0_d);
    const double ekp = eic_distance(rayk_p, rayk_d, w.crk_p, w.crk_d);
    v3 b4p = rayk_p, b4d = rayk_d;
    if (w.after_kind != 0) {
        const v3 t{rayk_p.x - w.after_t[0], rayk_p.y - w.after_t[1], rayk_p.z - w.after_t[2]};
        if (w.after_kind == 1) {
            b4p = t;
        } else {
            b4p = rotate(w.after_rt, w.after_order, t);
            b4d = rotate(w.after_rt, w.after_order, rayk_d);
        }
    }
    const double dst = ekp - w.cr_exp_dist;
    const v3 pc{(b4p.x - dst * b4d.x) - w.cr_exp_pt[0], (b4p.y - dst * b4d.y) - w.cr_exp_pt[1],
                (b4p.z - dst * b4d.z) - w.cr_exp_pt[2]};
    const v3 rd{w.ref_dir[0], w.ref_dir[1], w.ref_dir[2]};
    const double R = w.ref_radius;
    const double F = dot3(rd, b4d) - dot3(b4d, pc) / R;
    const double J = dot3(pc, pc) / R - 2.0 * dot3(rd, pc);
    const double denom = F + w.sign_soln * sqrt(F * F + J / R);
    const double ep = (denom == 0) ? 0 : J / denom;
    return -w.n_obj * e1 - ray_op + w.n_img * ekp + w.cr_op - w.n_img * ep;
}

// ------------------------------------------------------------------ stores
// One packet component of ray r lives at seg[(slot*10 + c)*ld + r].  The
// (slot, c) part is wave-uniform, so it goes into an SGPR base; the ray part
// is one 32-bit byte offset per lane computed once per ray: the store is
// `global_store_dwordx2 v_off, v_data, s[base:base+1]` with no per-store
// 64-bit VALU address arithmetic.  (The host splits batches longer than 2^28
// rays into several launches so that the lane offset always fits 32 bits.)
struct SegOut {
    char *base;         // uniform: seg (already offset to this launch's first ray)
    int64_t row_bytes;  // uniform: ld * 8
    uint32_t voff;      // per lane: (r - first ray of the launch) * 8 < 2^32
    __device__ __forceinline__ void put(int slot, int c, double v) const
    {
        char *b = base + ((int64_t)slot * ROX_SEG_DOUBLES + c) * row_bytes;
        double *p = reinterpret_cast<double *>(b + (size_t)voff);
#if ROX_STORE_NT == 1
        __builtin_nontemporal_store(v, p);
#elif ROX_STORE_NT == 2       // write-through (sc1), experiment
        __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
        *p = v;
#endif
    }
    __device__ __forceinline__ void pdn(int slot, const v3 &p, const v3 &d, const v3 &n) const
    {
        put(slot, 0, p.x); put(slot, 1, p.y); put(slot, 2, p.z);
        put(slot, 3, d.x); put(slot, 4, d.y); put(slot, 5, d.z);
        put(slot, 7, n.x); put(slot, 8, n.y); put(slot, 9, n.z);
    }
    __device__ __forceinline__ void dst(int slot, double v) const { put(slot, 6, v); }
};

// system features a launch needs; the host picks the leanest instance
enum { F_POLY = 1,      // some interface is an Even/RadialPolynomial (Newton code)
       F_APLIST = 2,    // some interface carries clear_apertures
       F_PHFILT = 4 };  // filter_out_phantoms with phantoms present
constexpr int F_ALL = F_POLY | F_APLIST | F_PHFILT;

// ------------------------------------------------------------------ the kernel
template <int OUT_MODE, int GEN, bool PER_RAY_WVL, int FEAT>
__global__ void __launch_bounds__(kBlock, ROX_MIN_WAVES)
trace_kernel(const TraceArgs a)
{
    const int N = a.n_ifcs;
#if ROX_TABLE_SCALAR
    // wave-uniform table straight from the scalar cache: values land in SGPRs
    tblp tbl = (tblp)a.rows;
    tblp ntab = (tblp)a.n_table + (PER_RAY_WVL ? 0 : (size_t)a.wvl_idx_all * N);
    tbli slot = (tbli)a.slots;
    tbli nslots_before = slot + N;
#else
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double *tbl_w = lds;                               // [N][49]
    double *ntab_w = tbl_w + (size_t)N * kRowDoubles;  // [W][N] (or [N] for one wavelength)
    int32_t *slot_w = reinterpret_cast<int32_t *>(ntab_w + (PER_RAY_WVL ? (size_t)a.n_wvls * N : N));

    // stage the surface table once per workgroup
    for (int i = threadIdx.x; i < N * kRowDoubles; i += kBlock)
        tbl_w[i] = a.rows[i];
    if (PER_RAY_WVL) {
        for (int i = threadIdx.x; i < a.n_wvls * N; i += kBlock)
            ntab_w[i] = a.n_table[i];
    } else {
        for (int i = threadIdx.x; i < N; i += kBlock)
            ntab_w[i] = a.n_table[(size_t)a.wvl_idx_all * N + i];
    }
    for (int i = threadIdx.x; i < 2 * N; i += kBlock)
        slot_w[i] = a.slots[i];
    __syncthreads();
    tblp tbl = tbl_w;
    tblp ntab = ntab_w;
    tbli slot = slot_w;
    tbli nslots_before = slot + N;
#endif
    // without phantom filtering segment k of a packet is interface k
#define SLOT(s) ((FEAT & F_PHFILT) ? slot[s] : (s))
#define NSLOTS_BEFORE(s) ((FEAT & F_PHFILT) ? nslots_before[s] : (s))

    const uint32_t flags = a.opts.flags;
    const bool check_ap = flags & ROX_CHECK_APERTURES;
    const bool intersect_obj = flags & ROX_INTERSECT_OBJ;
    const bool filter_ph = (FEAT & F_PHFILT) && (flags & ROX_FILTER_PHANTOMS);
    const int first_surf = a.opts.first_surf, last_surf = a.opts.last_surf;
    const double eps = a.opts.eps, fuzz = a.opts.fuzz;
    const int64_t ld = a.out.ld;

    constexpr int O_CV = offsetof(rox_surface, cv) / 8, O_CC = offsetof(rox_surface, cc) / 8,
                  O_EC = offsetof(rox_surface, ec) / 8, O_CR = offsetof(rox_surface, cR) / 8,
                  O_COEF = offsetof(rox_surface, coefs) / 8,
                  O_RT = offsetof(rox_surface, rt) / 8, O_T = offsetof(rox_surface, t) / 8,
                  O_ZDIR = offsetof(rox_surface, z_dir) / 8;

#if ROX_XCD_SWIZZLE
    const unsigned per_xcd = (gridDim.x + 7u) / 8u;
    const unsigned vblock = (blockIdx.x % 8u) * per_xcd + blockIdx.x / 8u;
#else
    const unsigned vblock = blockIdx.x;
#endif
    for (int64_t r = (int64_t)vblock * kBlock + threadIdx.x; r < a.n_rays;
         r += (int64_t)gridDim.x * kBlock) {
        SegOut so;
        so.base = reinterpret_cast<char *>(a.out.seg);
        so.row_bytes = ld * 8;
        so.voff = (uint32_t)r * 8u;

        // ---- ray start -------------------------------------------------------
        v3 pt0, dir0;
        if (GEN == GEN_PUPIL) {
            double px, py;
            const int64_t rg = a.ray_base + r;
            if (a.axis_kind == AXIS_PRODUCT) {
                px = a.px[a.row_begin + rg / a.axis_num];
                py = a.py[rg % a.axis_num];
            } else {
                px = a.px[rg];
                py = a.py[rg];
            }
            if (flags & ROX_APPLY_VIGNETTING) {         // opticalspec.py:1339-1353
                if (px < 0.0) { if (a.fld.vlx != 0.0) px *= (1.0 - a.fld.vlx); }
                else          { if (a.fld.vux != 0.0) px *= (1.0 - a.fld.vux); }
                if (py < 0.0) { if (a.fld.vly != 0.0) py *= (1.0 - a.fld.vly); }
                else          { if (a.fld.vuy != 0.0) py *= (1.0 - a.fld.vuy); }
            }
            if (a.out.pupil) {
                a.out.pupil[r] = px;
                a.out.pupil[ld + r] = py;
            }
            // opticalspec.py:358-366
            const v3 pt1{a.fld.eprad * px + a.fld.aim[0], a.fld.eprad * py + a.fld.aim[1],
                         a.fld.z_enp};
            pt0 = v3{a.fld.pt0[0], a.fld.pt0[1], a.fld.pt0[2]};
            dir0 = unit(v3{pt1.x - pt0.x, pt1.y - pt0.y, pt1.z - pt0.z});
            if (dir0.z * a.fld.z_dir0 < 0)              // trace.py:307-308
                dir0 = v3{-dir0.x, -dir0.y, -dir0.z};
        } else {
            const int64_t rg = a.ray_base + r;
            pt0 = v3{a.pt0[rg], a.pt0[a.in_ld + rg], a.pt0[2 * a.in_ld + rg]};
            dir0 = v3{a.dir0[rg], a.dir0[a.in_ld + rg], a.dir0[2 * a.in_ld + rg]};
        }
        // per-ray wavelengths index the table per lane (a gather: plain pointer)
        const double *nwl = nullptr;
        if (PER_RAY_WVL)
            nwl = (ROX_TABLE_SCALAR ? a.n_table : (const double *)ntab) +
                  (size_t)a.wvl_idx[a.ray_base + r] * N;
#define NW(i) (PER_RAY_WVL ? nwl[i] : ntab[i])

        // ---- object surface, raytrace.py:145-158 -----------------------------
        int status = ROX_OK, fail_surf = -1;
        v3 bp, bn, bd = dir0;               // before_pt, before_normal, before_dir
        int b4_mode = ROX_DUMMY;
        {
            tblp row = tbl;
            if (intersect_obj) {
                const int2 mp{((tbli)row)[0], ((tbli)row)[1]};      // mode, profile
                b4_mode = mp.x;
                double s_;
                v3 df;
                bool ok;
                if (!(FEAT & F_POLY) || mp.y <= ROX_CONIC) {
                    ok = quadric_hit(mp.y == ROX_CONIC, row[O_CV], row[O_CC], row[O_EC], pt0, dir0,
                                     row[O_ZDIR], s_, bp);
                    const double k = (mp.y == ROX_CONIC) ? (row[O_CC] + 1.0) * row[O_CV] : row[O_CV];
                    df = v3{-row[O_CV] * bp.x, -row[O_CV] * bp.y, 1.0 - k * bp.z};
                } else {
                    ok = newton_hit(mp.y, row[O_CV], row[O_CC] + 1.0, row[O_EC], row[O_CR],
                                    ((tbli)row)[2], row + O_COEF, pt0, dir0, eps, s_, bp, df);
                }
                if (!ok) {              // raised outside the try block: no packet
                    status = ROX_MISSED_SURFACE;
                    fail_surf = 0;
                } else {
                    bn = unit(df);
                }
            } else {
                bp = pt0;
                bn = v3{0., 0., 1.};
            }
        }
        double z_dir_before = tbl[O_ZDIR];
        double opl = 0.0;
        double acc_dst = 0.0;           // dst of the most recently appended segment
        int acc_slot = 0;
        v3 inc{0, 0, 0}, nrm{0, 0, 0}, ad = dir0;
        v3 ray1_p{0, 0, 0}, rayk_p{0, 0, 0}, rayk_d{0, 0, 0};   // OPD mode: ray[1].p, ray[-2]
        if (OUT_MODE == ROX_OUT_FULL && status == ROX_OK)
            so.pdn(0, bp, bd, bn);

        // ---- remaining surfaces, raytrace.py:164-229 -------------------------
        for (int surf = 1; surf < N && status == ROX_OK; ++surf) {
            tblp prow = tbl + (size_t)(surf - 1) * kRowDoubles;     // `before`
            tblp row = tbl + (size_t)surf * kRowDoubles;             // `after`
            const int mode = ((tbli)row)[0], prof = ((tbli)row)[1];
            const double cv = row[O_CV];

            // :170-174 transform to the new vertex frame, closest approach
            const int rt_order = ((tbli)prow)[4];
            const v3 b4p = rotate(prow + O_RT, rt_order, v3{bp.x - prow[O_T], bp.y - prow[O_T + 1],
                                                            bp.z - prow[O_T + 2]});
            const v3 b4d = rotate(prow + O_RT, rt_order, bd);
            const double pp_dst = -dot3(b4p, b4d);
            const v3 pp{b4p.x + pp_dst * b4d.x, b4p.y + pp_dst * b4d.y, b4p.z + pp_dst * b4d.z};

            // :181-183 intersect
            double s;
            v3 df;
            bool ok;
            if (!(FEAT & F_POLY) || prof <= ROX_CONIC) {
                ok = quadric_hit(prof == ROX_CONIC, cv, row[O_CC], row[O_EC], pp, b4d,
                                 z_dir_before, s, inc);
            } else {
                ok = newton_hit(prof, cv, row[O_CC] + 1.0, row[O_EC], row[O_CR],
                                ((tbli)row)[2], row + O_COEF, pp, b4d, eps, s, inc, df);
            }
            const bool b4_filtered = filter_ph && (b4_mode == ROX_PHANTOM);
            if (!ok) {                                  // :231-237
                status = ROX_MISSED_SURFACE;
                fail_surf = surf;
                if (OUT_MODE == ROX_OUT_FULL) {
                    const int sl = b4_filtered ? NSLOTS_BEFORE(surf - 1) : SLOT(surf - 1);
                    if (b4_filtered)
                        so.pdn(sl, bp, bd, bn);
                    so.dst(sl, pp_dst);
                }
                break;
            }
            const double dst_b4 = pp_dst + s;
            // :185-191 the *previous* segment is completed only now
            if (b4_filtered) {
                acc_dst += dst_b4;
            } else {
                acc_dst = dst_b4;
                acc_slot = SLOT(surf - 1);
            }
            if (OUT_MODE == ROX_OUT_FULL)
                so.dst(acc_slot, acc_dst);

            // :193-194 (in_gap_range, :123-132)
            {
                const int g = surf - 1;
                const bool in_gap = !(last_surf >= 0 && first_surf == last_surf) && g >= first_surf &&
                                    (last_surf < 0 || g < last_surf);
                if (in_gap)
                    opl += NW(surf - 1) * dst_b4;
            }

            // :196 normal = normalize(df(inc_pt))
            if (!(FEAT & F_POLY) || prof <= ROX_CONIC) {
                const double k = (prof == ROX_CONIC) ? (row[O_CC] + 1.0) * cv : cv;
                df = v3{-cv * inc.x, -cv * inc.y, 1.0 - k * inc.z};
            }
            nrm = unit(df);

            // :198-202 aperture test (in_surface_range, :134-142)
            if (check_ap && surf >= first_surf && (last_surf < 0 || surf <= last_surf) &&
                mode != ROX_PHANTOM) {
                const bool in = (FEAT & F_APLIST)
                    ? inside_aperture(row, ((tbli)row)[3], inc.x, inc.y, fuzz)
                    : slim_sqrt(inc.x * inc.x + inc.y * inc.y) <=
                          row[offsetof(rox_surface, max_aperture) / 8] + fuzz;
                if (!in)
                    status = ROX_BLOCKED;               // :247-251
            }

            // :211-221 refract / reflect / pass through
            if (status == ROX_OK) {
                if (mode == ROX_REFLECT) {
                    ad = mirror(b4d, nrm);
                } else if (mode == ROX_TRANSMIT) {
                    if (!refract(b4d, nrm, NW(surf - 1), NW(surf), ad))
                        status = ROX_TIR;               // :239-245
                } else {
                    ad = b4d;
                }
            }
            if (status != ROX_OK) {
                // partial packet: [inc_pt, before_dir, 0.0, normal] in the next slot
                fail_surf = surf;
                if (OUT_MODE == ROX_OUT_FULL) {
                    const int sl = NSLOTS_BEFORE(surf);
                    so.pdn(sl, inc, bd, nrm);
                    so.dst(sl, 0.0);
                }
                break;
            }

            if (OUT_MODE == ROX_OUT_OPD) {
                if (surf == 1)
                    ray1_p = inc;
                if (surf == N - 2) {
                    rayk_p = inc;
                    rayk_d = ad;
                }
            }
            // :223-229 roll
            bp = inc; bd = ad;
            if (FEAT & F_PHFILT)
                bn = nrm;           // only a filtered phantom's late append needs it
            z_dir_before = row[O_ZDIR];
            b4_mode = mode;
            if (OUT_MODE == ROX_OUT_FULL) {
                const bool cur_filtered = filter_ph && (mode == ROX_PHANTOM) && surf < N - 1;
                if (!cur_filtered)
                    so.pdn(SLOT(surf), inc, ad, nrm);
            }
        }

        // ---- epilogue ---------------------------------------------------------
        if (status == ROX_OK) {                         // :259-262
            if (OUT_MODE == ROX_OUT_FULL) {
                so.dst(SLOT(N - 1), 0.0);
            } else if (OUT_MODE == ROX_OUT_LAST) {      // trace.py:214-217
                so.pdn(0, inc, ad, nrm);
                so.dst(0, 0.0);
            } else if (OUT_MODE == ROX_OUT_OPD) {
                so.put(0, 0, wave_abr_finite_pup(a.opts.wf, ray1_p, dir0, rayk_p, rayk_d, opl));
            } else {                                    // axisarrayfigure.py:229-238
                const double dist = a.opts.foc / ad.z;
                const double dx = inc.x + dist * ad.x;
                const double dy = inc.y + dist * ad.y;
                so.put(0, 0, dx - a.opts.image_pt[0]);
                so.put(0, 1, dy - a.opts.image_pt[1]);
            }
        }
        if (a.out.op)
            a.out.op[r] = opl;          // op_delta = 0 + opl on success; opl on failure (:236)
        if (a.out.status)
            a.out.status[r] = (uint8_t)status;
        if (a.out.fail_surf)
            a.out.fail_surf[r] = (int16_t)fail_surf;
    }
}
#undef NW
#undef SLOT
#undef NSLOTS_BEFORE

// trace.py:563-605 / 537-560: pupil coordinates by repeated `+=` of the step.
// lane 0 walks the x axis, lane 1 the y axis (both are sequential by
// definition: k-th value = start after k separately rounded additions).
__global__ void pupil_axes_kernel(double x0, double y0, double sx, double sy, int num,
                                  double *px, double *py)
{
    if (threadIdx.x > 1)
        return;
    double v = threadIdx.x == 0 ? x0 : y0;
    const double step = threadIdx.x == 0 ? sx : sy;
    double *out = threadIdx.x == 0 ? px : py;
    for (int k = 0; k < num; ++k) {
        out[k] = v;
        v += step;
    }
}

// Diagnostic: the slim fp64 paths against the plain operators on pseudo-random
// operands spanning the whole exponent range (zeros, denormals, band edges,
// inf and nan included).  counts[0] = sqrt mismatches, counts[1] = division
// mismatches, counts[2] = lanes that took a slim path.
__device__ __forceinline__ uint64_t mix64(uint64_t z)
{
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}

__device__ __forceinline__ double test_operand(uint64_t bits, int kind)
{
    // kind 0: any bit pattern; 1: in/near the band (the common case); 2: specials
    if (kind == 0)
        return __longlong_as_double((long long)bits);
    if (kind == 1) {
        const uint64_t e = 1023 - 40 + (bits >> 52) % 80;           // 2^-40 .. 2^40
        return __longlong_as_double((long long)((bits & 0x800fffffffffffffull) | (e << 52)));
    }
    const double sp[] = {0.0, -0.0, 1.0, -1.0, 4.9e-324, 2.2250738585072014e-308,
                         0x1p-383, 0x1.fffffffffffffp-384, 0x1p385, 0x1.fffffffffffffp384,
                         1.7976931348623157e308, __builtin_inf(), -__builtin_inf(),
                         __builtin_nan(""), 0x1p-767, 3.0};
    return sp[bits % 16];
}

__device__ __forceinline__ bool same_bits(double a, double b)
{
    return __double_as_longlong(a) == __double_as_longlong(b) || (a != a && b != b);
}

__global__ void __launch_bounds__(kBlock) selftest_kernel(uint64_t n, uint64_t seed,
                                                           unsigned long long *counts)
{
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n;
         i += (uint64_t)gridDim.x * kBlock) {
        // one operand class per wave so that the wave-uniform slim branch is taken
        const int kind = (int)((i / 64) % 4 == 3 ? (i / 256) % 3 : 1);
        const uint64_t k = i * 4 + seed * 0x9e3779b97f4a7c15ull;
        const double a0 = test_operand(mix64(k), kind), a1 = test_operand(mix64(k + 1), kind);
        const double a2 = test_operand(mix64(k + 2), kind), b = test_operand(mix64(k + 3), kind);
        const double x = fabs(a0);
        if (!same_bits(slim_sqrt(x), sqrt(x)))
            atomicAdd(&counts[0], 1ull);
        const v3 q = slim_div3(v3{a0, a1, a2}, b);
        if (!same_bits(q.x, a0 / b) || !same_bits(q.y, a1 / b) || !same_bits(q.z, a2 / b))
            atomicAdd(&counts[1], 1ull);
        if (__all(in_band(b) && in_band_or_zero(a0) && in_band_or_zero(a1) && in_band_or_zero(a2)))
            atomicAdd(&counts[2], 1ull);
    }
}

// ------------------------------------------------------------------ host side
thread_local char g_err[512] = "";

int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(expr)                                                                    \
    do {                                                                                 \
        hipError_t e_ = (expr);                                                          \
        if (e_ != hipSuccess)                                                            \
            return fail(ROX_E_HIP, "%s: %s", #expr, hipGetErrorString(e_));              \
    } while (0)

}  // namespace

struct rox_system {
    int device = 0;
    int32_t n_ifcs = 0, n_wvls = 0;
    std::vector<rox_surface> rows;      // host copy (immutable)
    double *d_rows = nullptr;
    double *d_ntab = nullptr;
    int32_t *d_slots[2] = {nullptr, nullptr};   // [0]: no phantom filtering, [1]: filtered
    int32_t n_seg[2] = {0, 0};
    double *d_axes = nullptr;           // pupil axes [2][axes_cap]
    int32_t axes_cap = 0;
    // the axes currently held in d_axes (spot diagrams reuse one grid definition
    // for every field and wavelength): skip the serial accumulate when unchanged
    double axes_key[4] = {0, 0, 0, 0};
    int32_t axes_num = 0;
    hipStream_t axes_stream = nullptr;
    int num_cus = 256;
    int features = 0;                   // F_POLY | F_APLIST of the table
};

namespace {

void slot_map(const rox_system *s, bool filter, std::vector<int32_t> &m, int32_t &n_seg)
{
    const int N = s->n_ifcs;
    m.assign(2 * N, 0);
    int next = 0;
    for (int i = 0; i < N; ++i) {
        // a phantom's segment is dropped when the *following* surface sees
        // b4_interact_mode == 'phantom' (raytrace.py:185-188); the last
        // interface has no follower, and the object is 'dummy' unless
        // intersect_obj reads its own mode
        const bool filtered = filter && s->rows[i].mode == ROX_PHANTOM && i > 0 && i < N - 1;
        m[N + i] = next;                // nslots_before[i]
        m[i] = filtered ? -1 : next++;
    }
    n_seg = next;
}

size_t lds_bytes(const rox_system *s, bool per_ray_wvl)
{
    const size_t N = s->n_ifcs;
    size_t b = N * sizeof(dev_surface) + (per_ray_wvl ? (size_t)s->n_wvls * N : N) * sizeof(double) +
               2 * N * sizeof(int32_t);
    return (b + 15) & ~size_t(15);
}

int check_opts(const rox_system *sys, const rox_opts *o, const rox_out *out, int64_t n_rays)
{
    if (!sys || !o || !out)
        return fail(ROX_E_ARG, "null argument");
    if (o->out_mode < ROX_OUT_FULL || o->out_mode > ROX_OUT_OPD)
        return fail(ROX_E_ARG, "bad out_mode %d", o->out_mode);
    if (o->out_mode == ROX_OUT_OPD) {
        if (sys->n_ifcs < 3)
            return fail(ROX_E_ARG, "OPD output needs at least 3 interfaces");
        if ((o->flags & ROX_FILTER_PHANTOMS) && sys->n_seg[1] != sys->n_seg[0])
            return fail(ROX_E_UNSUPPORTED, "OPD output with filter_out_phantoms");
        if (!(o->wf.ref_radius != 0.0))
            return fail(ROX_E_ARG, "OPD output needs rox_opts.wf (ref_radius is 0)");
    }
    if (out->ld < n_rays)
        return fail(ROX_E_ARG, "out.ld (%lld) < n_rays (%lld)", (long long)out->ld, (long long)n_rays);
    if (!out->seg && n_rays > 0)
        return fail(ROX_E_ARG, "out.seg is null");
    if ((o->flags & ROX_FILTER_PHANTOMS) && sys->rows[0].mode == ROX_PHANTOM)
        return fail(ROX_E_UNSUPPORTED, "phantom object surface with filter_out_phantoms");
    return 0;
}

// workgroups per CU of a launch (the rest of the batch is grid-strided).
// ROX_BLOCKS_PER_CU overrides it for experiments (tools/ab_bench.py).
int blocks_per_cu()
{
    static const int v = [] {
        const char *e = getenv("ROX_BLOCKS_PER_CU");
        const int n = e ? atoi(e) : 0;
        return n > 0 ? n : 32 * 256 / kBlock;      // measured: FULL 238 us at 8/CU, 228 us at 32/CU
    }();
    return v;
}

// rays per kernel launch (lane byte offsets are 32-bit: at most 2^28).
// ROX_RAYS_PER_LAUNCH overrides it for experiments.
int64_t rays_per_launch()
{
    static const int64_t v = [] {
        const char *e = getenv("ROX_RAYS_PER_LAUNCH");
        const long long n = e ? atoll(e) : 0;
        return (n > 0 && n <= (1LL << 28)) ? (int64_t)n : (int64_t(1) << 28);
    }();
    return v;
}

template <int GEN, bool PRW, int FEAT>
void launch_mode(int out_mode, dim3 grid, size_t lds, hipStream_t st, const TraceArgs &a)
{
    switch (out_mode) {
    case ROX_OUT_FULL:
        hipLaunchKernelGGL((trace_kernel<ROX_OUT_FULL, GEN, PRW, FEAT>), grid, dim3(kBlock), lds, st, a);
        break;
    case ROX_OUT_LAST:
        hipLaunchKernelGGL((trace_kernel<ROX_OUT_LAST, GEN, PRW, FEAT>), grid, dim3(kBlock), lds, st, a);
        break;
    case ROX_OUT_OPD:
        hipLaunchKernelGGL((trace_kernel<ROX_OUT_OPD, GEN, PRW, FEAT>), grid, dim3(kBlock), lds, st, a);
        break;
    default:
        hipLaunchKernelGGL((trace_kernel<ROX_OUT_HITS, GEN, PRW, FEAT>), grid, dim3(kBlock), lds, st, a);
        break;
    }
}

template <int FEAT>
void launch_gen(int gen, bool prw, int out_mode, dim3 grid, size_t lds, hipStream_t st,
                const TraceArgs &a)
{
    if (gen == GEN_PUPIL)
        launch_mode<GEN_PUPIL, false, FEAT>(out_mode, grid, lds, st, a);
    else if (prw)
        launch_mode<GEN_RAYS, true, FEAT>(out_mode, grid, lds, st, a);
    else
        launch_mode<GEN_RAYS, false, FEAT>(out_mode, grid, lds, st, a);
}

int launch(const rox_system *sys, TraceArgs &a, int gen, hipStream_t st)
{
    if (a.n_rays == 0)
        return 0;
    const bool prw = a.wvl_idx != nullptr;
    a.rows = sys->d_rows;
    a.n_table = sys->d_ntab;
    a.slots = sys->d_slots[(a.opts.flags & ROX_FILTER_PHANTOMS) ? 1 : 0];
    a.n_ifcs = sys->n_ifcs;
    a.n_wvls = sys->n_wvls;
    const size_t lds = ROX_TABLE_SCALAR ? 0 : lds_bytes(sys, prw);
    if (lds > 160 * 1024)
        return fail(ROX_E_UNSUPPORTED, "surface table needs %zu B of LDS (max 163840)", lds);
    // the leanest kernel instance that covers this system and these options
    int feat = sys->features;
    if ((a.opts.flags & ROX_FILTER_PHANTOMS) && sys->n_seg[1] != sys->n_seg[0])
        feat |= F_PHFILT;
    // lane byte offsets are 32-bit: at most 2^28 rays per launch
    const int64_t total = a.n_rays, chunk_max = rays_per_launch();
    const rox_out out0 = a.out;
    a.in_ld = total;
    for (int64_t base = 0; base < total; base += chunk_max) {
        a.ray_base = base;
        a.n_rays = total - base < chunk_max ? total - base : chunk_max;
        a.out.seg = out0.seg + base;
        a.out.op = out0.op ? out0.op + base : nullptr;
        a.out.status = out0.status ? out0.status + base : nullptr;
        a.out.fail_surf = out0.fail_surf ? out0.fail_surf + base : nullptr;
        a.out.pupil = out0.pupil ? out0.pupil + base : nullptr;
        // enough workgroups to fill 256 CUs several times over, grid-stride the rest
        int64_t blocks = (a.n_rays + kBlock - 1) / kBlock;
        const int64_t cap = (int64_t)sys->num_cus * blocks_per_cu();
        if (blocks > cap)
            blocks = cap;
        const dim3 grid((unsigned)blocks);
        if (feat == 0)
            launch_gen<0>(gen, prw, a.opts.out_mode, grid, lds, st, a);
        else
            launch_gen<F_ALL>(gen, prw, a.opts.out_mode, grid, lds, st, a);
    }
    a.n_rays = total;
    a.out = out0;
    HIP_TRY(hipGetLastError());
    return 0;
}

int64_t seg_rows(const rox_system *sys, const rox_opts *o)
{
    if (o->out_mode == ROX_OUT_FULL)
        return (int64_t)sys->n_seg[(o->flags & ROX_FILTER_PHANTOMS) ? 1 : 0] * ROX_SEG_DOUBLES;
    if (o->out_mode == ROX_OUT_OPD)
        return 1;
    return o->out_mode == ROX_OUT_LAST ? ROX_SEG_DOUBLES : 2;
}

// ROX_HOST_POINTERS: stage outputs through HBM
struct Staged {
    rox_out dev{};          // device-side buffers
    rox_out host{};         // caller's host buffers
    int64_t n = 0, rows = 0;
    bool want_pupil = false;
    ~Staged()
    {
        (void)hipFree(dev.seg); (void)hipFree(dev.op); (void)hipFree(dev.status); (void)hipFree(dev.fail_surf);
        (void)hipFree(dev.pupil);
    }
};

int stage_out(Staged &s, const rox_system *sys, const rox_opts *o, const rox_out *out, int64_t n)
{
    s.host = *out;
    s.n = n;
    s.rows = seg_rows(sys, o);
    s.dev.ld = n;
    HIP_TRY(hipMalloc(&s.dev.seg, sizeof(double) * s.rows * n));
    // untouched slots keep the caller's bytes
    HIP_TRY(hipMemcpy2D(s.dev.seg, sizeof(double) * n, out->seg, sizeof(double) * out->ld,
                        sizeof(double) * n, s.rows, hipMemcpyHostToDevice));
    if (out->op) {
        HIP_TRY(hipMalloc(&s.dev.op, sizeof(double) * n));
    }
    if (out->status) {
        HIP_TRY(hipMalloc(&s.dev.status, n));
    }
    if (out->fail_surf) {
        HIP_TRY(hipMalloc(&s.dev.fail_surf, sizeof(int16_t) * n));
    }
    if (out->pupil) {
        HIP_TRY(hipMalloc(&s.dev.pupil, sizeof(double) * 2 * n));
    }
    return 0;
}

int unstage_out(Staged &s, hipStream_t st)
{
    HIP_TRY(hipStreamSynchronize(st));
    HIP_TRY(hipMemcpy2D(s.host.seg, sizeof(double) * s.host.ld, s.dev.seg, sizeof(double) * s.n,
                        sizeof(double) * s.n, s.rows, hipMemcpyDeviceToHost));
    if (s.host.op)
        HIP_TRY(hipMemcpy(s.host.op, s.dev.op, sizeof(double) * s.n, hipMemcpyDeviceToHost));
    if (s.host.status)
        HIP_TRY(hipMemcpy(s.host.status, s.dev.status, s.n, hipMemcpyDeviceToHost));
    if (s.host.fail_surf)
        HIP_TRY(hipMemcpy(s.host.fail_surf, s.dev.fail_surf, sizeof(int16_t) * s.n,
                          hipMemcpyDeviceToHost));
    if (s.host.pupil)
        HIP_TRY(hipMemcpy2D(s.host.pupil, sizeof(double) * s.host.ld, s.dev.pupil,
                            sizeof(double) * s.n, sizeof(double) * s.n, 2, hipMemcpyDeviceToHost));
    return 0;
}

int ensure_axes(rox_system *sys, int32_t num)
{
    if (num <= sys->axes_cap)
        return 0;
    if (sys->d_axes)
        HIP_TRY(hipFree(sys->d_axes));
    sys->d_axes = nullptr;
    sys->axes_cap = 0;
    sys->axes_num = 0;
    HIP_TRY(hipMalloc(&sys->d_axes, sizeof(double) * 2 * (size_t)num));
    sys->axes_cap = num;
    return 0;
}

int prepare_grid(rox_system *sys, const rox_field *fld, const rox_grid *grid, int32_t wvl_idx,
                 const rox_opts *opts, const rox_out *out, hipStream_t st, TraceArgs &a)
{
    if (!fld || !grid)
        return fail(ROX_E_ARG, "null argument");
    if (grid->num < 1)
        return fail(ROX_E_ARG, "grid.num must be >= 1");
    if (wvl_idx < 0 || wvl_idx >= sys->n_wvls)
        return fail(ROX_E_ARG, "wvl_idx %d out of range", wvl_idx);
    int32_t rows = grid->num, row0 = 0;
    if (grid->kind != ROX_GRID_FAN && grid->row_count > 0) {
        rows = grid->row_count;
        row0 = grid->row_begin;
    }
    if (row0 < 0 || row0 + rows > grid->num)
        return fail(ROX_E_ARG, "grid row block [%d, %d) outside [0, %d)", row0, row0 + rows, grid->num);
    const int64_t R = grid->kind == ROX_GRID_FAN ? grid->num : (int64_t)rows * grid->num;
    int rc = check_opts(sys, opts, out, R);
    if (rc)
        return rc;
    rc = ensure_axes(sys, grid->num);
    if (rc)
        return rc;
    // trace.py:566-570 step = (stop - start)/(num - 1)
    const double sx = (grid->stop[0] - grid->start[0]) / (grid->num - 1);
    const double sy = (grid->stop[1] - grid->start[1]) / (grid->num - 1);
    double *px = sys->d_axes, *py = sys->d_axes + sys->axes_cap;
    const double key[4] = {grid->start[0], grid->start[1], sx, sy};
    if (sys->axes_num != grid->num || sys->axes_stream != st ||
        memcmp(key, sys->axes_key, sizeof key) != 0) {
        hipLaunchKernelGGL(pupil_axes_kernel, dim3(1), dim3(64), 0, st, grid->start[0],
                           grid->start[1], sx, sy, grid->num, px, py);
        HIP_TRY(hipGetLastError());
        memcpy(sys->axes_key, key, sizeof key);
        sys->axes_num = grid->num;
        sys->axes_stream = st;
    }
    a = TraceArgs{};
    a.n_rays = R;
    a.px = px;
    a.py = py;
    a.axis_kind = grid->kind == ROX_GRID_FAN ? AXIS_LIST : AXIS_PRODUCT;
    a.axis_num = grid->num;
    a.row_begin = row0;
    a.wvl_idx_all = wvl_idx;
    a.fld = *fld;
    a.opts = *opts;
    a.out = *out;
    return 0;
}

}  // namespace

// ---------------------------------------------------------------------- C ABI
extern "C" {

int rox_abi_version(void) { return ROX_ABI_VERSION; }

const char *rox_last_error(void) { return g_err; }

int rox_device_count(int *count)
{
    if (!count)
        return fail(ROX_E_ARG, "null argument");
    *count = 0;
    hipError_t e = hipGetDeviceCount(count);
    if (e != hipSuccess) {
        *count = 0;
        return fail(ROX_E_NO_DEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e));
    }
    return 0;
}

int rox_set_device(int device)
{
    HIP_TRY(hipSetDevice(device));
    return 0;
}

int rox_system_create(const rox_surface *rows, int32_t n_ifcs, const double *n_table,
                      int32_t n_wvls, rox_system **out_sys)
{
    if (!rows || !n_table || !out_sys || n_ifcs < 2 || n_wvls < 1)
        return fail(ROX_E_ARG, "rox_system_create: bad argument");
    for (int i = 0; i < n_ifcs; ++i) {
        const rox_surface &s = rows[i];
        if (s.mode < ROX_TRANSMIT || s.mode > ROX_PHANTOM || s.profile < ROX_SPHERICAL ||
            s.profile > ROX_XTOROID || s.ncoef < 0 || s.ncoef > ROX_MAX_COEF || s.n_ap < 0 ||
            s.n_ap > ROX_MAX_AP)
            return fail(ROX_E_ARG, "rox_system_create: row %d is malformed", i);
    }
    rox_system *sys = new (std::nothrow) rox_system;
    if (!sys)
        return fail(ROX_E_NOMEM, "out of host memory");
    sys->n_ifcs = n_ifcs;
    sys->n_wvls = n_wvls;
    sys->rows.assign(rows, rows + n_ifcs);
    for (int i = 0; i < n_ifcs; ++i) {
        if (rows[i].profile > ROX_CONIC)
            sys->features |= F_POLY;
        if (rows[i].n_ap > 0)
            sys->features |= F_APLIST;
    }
    hipError_t e = hipGetDevice(&sys->device);
    if (e != hipSuccess) {
        delete sys;
        return fail(ROX_E_NO_DEVICE, "hipGetDevice: %s", hipGetErrorString(e));
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, sys->device) == hipSuccess && prop.multiProcessorCount > 0)
        sys->num_cus = prop.multiProcessorCount;
    int rc = 0;
    do {
        std::vector<dev_surface> drows(n_ifcs);
        for (int i = 0; i < n_ifcs; ++i) {
            drows[i].pub = rows[i];
            const double c0 = rows[i].profile == ROX_RADIALPOLY ? 1.0 : 2.0;
            double c_coef = c0;                 // profiles.py:877-882, 1104-1109, 1364-1369
            for (int k = 0; k < ROX_MAX_COEF; ++k) {
                drows[i].dcoefs[k] = c_coef * rows[i].coefs[k];
                c_coef += c0;
            }
        }
        const size_t rb = sizeof(dev_surface) * n_ifcs, nb = sizeof(double) * n_wvls * n_ifcs;
        if (hipMalloc(&sys->d_rows, rb) != hipSuccess || hipMalloc(&sys->d_ntab, nb) != hipSuccess) {
            rc = fail(ROX_E_HIP, "hipMalloc failed for the surface table");
            break;
        }
        if (hipMemcpy(sys->d_rows, drows.data(), rb, hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(sys->d_ntab, n_table, nb, hipMemcpyHostToDevice) != hipSuccess) {
            rc = fail(ROX_E_HIP, "hipMemcpy failed for the surface table");
            break;
        }
        for (int f = 0; f < 2 && !rc; ++f) {
            std::vector<int32_t> m;
            slot_map(sys, f == 1, m, sys->n_seg[f]);
            if (hipMalloc(&sys->d_slots[f], sizeof(int32_t) * m.size()) != hipSuccess ||
                hipMemcpy(sys->d_slots[f], m.data(), sizeof(int32_t) * m.size(),
                          hipMemcpyHostToDevice) != hipSuccess)
                rc = fail(ROX_E_HIP, "slot map upload failed");
        }
    } while (0);
    if (rc) {
        rox_system_destroy(sys);
        return rc;
    }
    *out_sys = sys;
    return 0;
}

int rox_system_destroy(rox_system *sys)
{
    if (!sys)
        return 0;
    (void)hipFree(sys->d_rows);
    (void)hipFree(sys->d_ntab);
    (void)hipFree(sys->d_slots[0]);
    (void)hipFree(sys->d_slots[1]);
    (void)hipFree(sys->d_axes);
    delete sys;
    return 0;
}

int rox_system_num_segments(const rox_system *sys, uint32_t flags, int32_t *n_seg)
{
    if (!sys || !n_seg)
        return fail(ROX_E_ARG, "null argument");
    *n_seg = sys->n_seg[(flags & ROX_FILTER_PHANTOMS) ? 1 : 0];
    return 0;
}

int rox_trace_rays(rox_system *sys, int64_t n_rays, const double *pt0, const double *dir0,
                   const int32_t *wvl_idx, int32_t wvl_idx_all, const rox_opts *opts,
                   const rox_out *out, void *stream)
{
    int rc = check_opts(sys, opts, out, n_rays);
    if (rc)
        return rc;
    if (n_rays < 0 || (n_rays > 0 && (!pt0 || !dir0)))
        return fail(ROX_E_ARG, "rox_trace_rays: bad ray buffers");
    if (!wvl_idx && (wvl_idx_all < 0 || wvl_idx_all >= sys->n_wvls))
        return fail(ROX_E_ARG, "wvl_idx %d out of range", wvl_idx_all);
    hipStream_t st = (hipStream_t)stream;
    TraceArgs a{};
    a.n_rays = n_rays;
    a.wvl_idx_all = wvl_idx_all;
    a.opts = *opts;
    if (!(opts->flags & ROX_HOST_POINTERS)) {
        a.pt0 = pt0; a.dir0 = dir0; a.wvl_idx = wvl_idx; a.out = *out;
        return launch(sys, a, GEN_RAYS, st);
    }
    if (wvl_idx)
        for (int64_t i = 0; i < n_rays; ++i)
            if (wvl_idx[i] < 0 || wvl_idx[i] >= sys->n_wvls)
                return fail(ROX_E_ARG, "wvl_idx[%lld] = %d out of range", (long long)i, wvl_idx[i]);
    Staged s;
    rc = stage_out(s, sys, opts, out, n_rays);
    if (rc)
        return rc;
    double *d_in = nullptr;
    int32_t *d_w = nullptr;
    const size_t vb = sizeof(double) * 3 * (size_t)n_rays;
    hipError_t e = hipMalloc(&d_in, 2 * vb + 16);
    if (e == hipSuccess && wvl_idx)
        e = hipMalloc(&d_w, sizeof(int32_t) * (size_t)n_rays + 16);
    if (e == hipSuccess)
        e = hipMemcpy(d_in, pt0, vb, hipMemcpyHostToDevice);
    if (e == hipSuccess)
        e = hipMemcpy(d_in + 3 * n_rays, dir0, vb, hipMemcpyHostToDevice);
    if (e == hipSuccess && wvl_idx)
        e = hipMemcpy(d_w, wvl_idx, sizeof(int32_t) * (size_t)n_rays, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        a.pt0 = d_in; a.dir0 = d_in + 3 * n_rays; a.wvl_idx = d_w; a.out = s.dev;
        rc = launch(sys, a, GEN_RAYS, st);
        if (!rc)
            rc = unstage_out(s, st);
    } else {
        rc = fail(ROX_E_HIP, "staging inputs: %s", hipGetErrorString(e));
    }
    (void)hipFree(d_in);
    (void)hipFree(d_w);
    return rc;
}

int rox_trace_pupil_grid(rox_system *sys, const rox_field *fld, const rox_grid *grid,
                         int32_t wvl_idx, const rox_opts *opts, const rox_out *out, void *stream)
{
    if (!sys)
        return fail(ROX_E_ARG, "null system");
    hipStream_t st = (hipStream_t)stream;
    TraceArgs a;
    int rc = prepare_grid(sys, fld, grid, wvl_idx, opts, out, st, a);
    if (rc)
        return rc;
    if (!(opts->flags & ROX_HOST_POINTERS))
        return launch(sys, a, GEN_PUPIL, st);
    Staged s;
    rc = stage_out(s, sys, opts, out, a.n_rays);
    if (rc)
        return rc;
    a.out = s.dev;
    rc = launch(sys, a, GEN_PUPIL, st);
    return rc ? rc : unstage_out(s, st);
}

int rox_trace_pupil_list(rox_system *sys, const rox_field *fld, int64_t n_rays, const double *px,
                         const double *py, int32_t wvl_idx, const rox_opts *opts,
                         const rox_out *out, void *stream)
{
    int rc = check_opts(sys, opts, out, n_rays);
    if (rc)
        return rc;
    if (!fld || n_rays < 0 || (n_rays > 0 && (!px || !py)))
        return fail(ROX_E_ARG, "rox_trace_pupil_list: bad argument");
    if (wvl_idx < 0 || wvl_idx >= sys->n_wvls)
        return fail(ROX_E_ARG, "wvl_idx %d out of range", wvl_idx);
    hipStream_t st = (hipStream_t)stream;
    TraceArgs a{};
    a.n_rays = n_rays;
    a.axis_kind = AXIS_LIST;
    a.axis_num = 1;
    a.wvl_idx_all = wvl_idx;
    a.fld = *fld;
    a.opts = *opts;
    if (!(opts->flags & ROX_HOST_POINTERS)) {
        a.px = px; a.py = py; a.out = *out;
        return launch(sys, a, GEN_PUPIL, st);
    }
    Staged s;
    rc = stage_out(s, sys, opts, out, n_rays);
    if (rc)
        return rc;
    double *d_p = nullptr;
    hipError_t e = hipMalloc(&d_p, sizeof(double) * 2 * (size_t)n_rays + 16);
    if (e == hipSuccess)
        e = hipMemcpy(d_p, px, sizeof(double) * n_rays, hipMemcpyHostToDevice);
    if (e == hipSuccess)
        e = hipMemcpy(d_p + n_rays, py, sizeof(double) * n_rays, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        a.px = d_p; a.py = d_p + n_rays; a.out = s.dev;
        rc = launch(sys, a, GEN_PUPIL, st);
        if (!rc)
            rc = unstage_out(s, st);
    } else {
        rc = fail(ROX_E_HIP, "staging pupil coordinates: %s", hipGetErrorString(e));
    }
    (void)hipFree(d_p);
    return rc;
}

int rox_selftest_fp64(uint64_t n, uint64_t seed, uint64_t counts[3])
{
    if (!counts)
        return fail(ROX_E_ARG, "null argument");
    unsigned long long *d = nullptr;
    HIP_TRY(hipMalloc(&d, 3 * sizeof(unsigned long long)));
    HIP_TRY(hipMemset(d, 0, 3 * sizeof(unsigned long long)));
    hipLaunchKernelGGL(selftest_kernel, dim3(2048), dim3(kBlock), 0, nullptr, n, seed, d);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess)
        e = hipMemcpy(counts, d, 3 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    (void)hipFree(d);
    if (e != hipSuccess)
        return fail(ROX_E_HIP, "selftest: %s", hipGetErrorString(e));
    return 0;
}

int rox_time_pupil_grid(rox_system *sys, const rox_field *fld, const rox_grid *grid,
                        int32_t wvl_idx, const rox_opts *opts, const rox_out *out, void *stream,
                        int32_t launches, double *mean_ms)
{
    if (!sys || !mean_ms || launches < 1)
        return fail(ROX_E_ARG, "rox_time_pupil_grid: bad argument");
    if (opts && (opts->flags & ROX_HOST_POINTERS))
        return fail(ROX_E_ARG, "rox_time_pupil_grid needs device buffers");
    hipStream_t st = (hipStream_t)stream;
    TraceArgs a;
    int rc = prepare_grid(sys, fld, grid, wvl_idx, opts, out, st, a);
    if (rc)
        return rc;
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0));
    HIP_TRY(hipEventCreate(&e1));
    HIP_TRY(hipEventRecord(e0, st));
    for (int i = 0; i < launches && !rc; ++i)
        rc = launch(sys, a, GEN_PUPIL, st);
    HIP_TRY(hipEventRecord(e1, st));
    HIP_TRY(hipEventSynchronize(e1));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    *mean_ms = (double)ms / launches;
    return rc;
}

}  // extern "C"
