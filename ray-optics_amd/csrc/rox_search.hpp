// rox_search.hpp -- the search kernels (many short dependent traces per problem), templates
// over the feature instance of their trial-ray trace; csrc/search_*.hip instantiate them, one
// translation unit per instance like the trace kernels (a lean system's trial ray is then the
// lean trace: double Gauss aiming 0.43 -> 0.35 ms against the F_ALL trace it used to inline).
//
// Batched chief-ray aiming: the 1-D branch of trace.iterate_ray
// (rayoptics/raytr/trace.py:313-415) with scipy.optimize.newton's secant
// iteration (scipy/optimize/_zeros_py.py, `fprime is None` branch; x0 = 0,
// tol = 1.48e-8, rtol = 0, maxiter = 50, disp = False) restated per lane: one
// lane = one (field, wavelength) problem, every trial ray traced through the
// whole system with raytrace.trace's defaults (raytrace.py:51-80).
#pragma once
#include "rox_device.hpp"
#include "rox_hybrd.hpp"

#ifndef ROX_ENP_SPEC        // samples of the wide-angle walk traced at once, per direction (<= 32)
#define ROX_ENP_SPEC 4
#endif

namespace rox {

// The table of a search kernel's workgroup (64 threads): staged in LDS -- every wavelength's
// indices and phase constants, the trial rays carry their own wavelength index -- or, for the
// F_GTAB instance (tables beyond the LDS), left in global memory and read with scalar loads,
// with only the indices, the slot map and the aperture thresholds of the checked trace in LDS.
// ap_fuzz < 0: the kernel's trial rays never test apertures (no thresholds staged).
template <int FEAT, class ARGS>
__device__ __forceinline__ auto search_ctx(const ARGS &a, double *lds, double ap_fuzz)
{
    constexpr bool kGtab = (FEAT & F_GTAB) != 0;
    typedef typename std::conditional<kGtab, ctblp, tblp>::type TP;
    const int N = a.n_ifcs, W = a.n_wvls;
    double *tbl_w = lds;
    double *ntab_w = tbl_w + (kGtab ? 0 : (size_t)N * kRowDoubles);
    double *phc_w = ntab_w + (size_t)W * N;
    double *wvls_w = phc_w + (kGtab ? 0 : (size_t)W * N * kPhaseConsts);
    int32_t *slot_w = reinterpret_cast<int32_t *>(wvls_w + W);
    double *apthr_w = reinterpret_cast<double *>(slot_w + 2 * N);      // [N]
    double *aplthr_w = apthr_w + N;                                     // [N][ROX_MAX_AP] (F_GTAB)
    if (!kGtab) {
        for (int i = threadIdx.x; i < N * kRowDoubles; i += 64)
            tbl_w[i] = a.rows[i];
        for (int i = threadIdx.x; i < W * N * kPhaseConsts; i += 64)
            phc_w[i] = a.ph_consts[i];
    }
    for (int i = threadIdx.x; i < W * N; i += 64)
        ntab_w[i] = a.n_table[i];
    for (int i = threadIdx.x; i < W; i += 64)
        wvls_w[i] = a.wvls[i];
    for (int i = threadIdx.x; i < 2 * N; i += 64)
        slot_w[i] = a.slots[i];
    if (ap_fuzz >= 0.0) {
        // the sqrt-free aperture thresholds of the checked trace, behind the slot map
        for (int i = threadIdx.x; i < N; i += 64)
            apthr_w[i] = sqrt_le_threshold(
                a.rows[(size_t)i * kRowDoubles + offsetof(rox_surface, max_aperture) / 8] + ap_fuzz);
        if (kGtab && (FEAT & F_APLIST))
            for (int i = threadIdx.x; i < N * ROX_MAX_AP; i += 64) {
                const double *row = a.rows + (size_t)(i / ROX_MAX_AP) * kRowDoubles;
                const int k = i % ROX_MAX_AP;
                double t = 0.0;
                if (k < reinterpret_cast<const int32_t *>(row)[3]) {
                    const double *ap = row + offsetof(rox_surface, ap) / sizeof(double) +
                                       (size_t)k * (sizeof(rox_aperture) / sizeof(double));
                    if (reinterpret_cast<const int32_t *>(ap)[0] == ROX_AP_CIRCULAR)
                        t = sqrt_le_threshold(ap[3] + ap_fuzz);
                }
                aplthr_w[i] = t;
            }
    }
    __syncthreads();
    if (!kGtab && (FEAT & F_APLIST) && ap_fuzz >= 0.0) {
        stage_aperture_thresholds<FEAT>(tbl_w, N, ap_fuzz, threadIdx.x, 64);
        __syncthreads();
    }
    CtxT<TP> c;
    if constexpr (kGtab) {
        c.tbl = (ctblp)a.rows;
        c.phc = (ctblp)a.ph_consts;
        c.aplthr = ((FEAT & F_APLIST) && ap_fuzz >= 0.0) ? aplthr_w : nullptr;
    } else {
        c.tbl = tbl_w;
        c.phc = phc_w;
        c.aplthr = nullptr;
    }
    c.ntab = ntab_w; c.wvls = wvls_w;
    c.slot = slot_w; c.nslots_before = slot_w + N;
    c.apthr = ap_fuzz >= 0.0 ? apthr_w : nullptr;
    c.mu = nullptr;
    c.N = N;
    return c;
}

template <int FEAT>
__global__ void __launch_bounds__(64) aim_kernel(const AimArgs a)
{
    const int N = a.n_ifcs;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    auto c = search_ctx<FEAT>(a, lds, -1.0);     // (the trial rays of the aiming never test apertures)
    // a.wave_per_problem (small batches: the fields of a model): every lane of the wave follows
    // the same problem -- problems in different phases of their iterations (the call sites of
    // the trace inside MINPACK's hybrd) then run side by side on different CUs instead of
    // one after the other in one wave; the 64 lanes store identical results
    const int i = a.wave_per_problem ? (int)blockIdx.x : (int)(blockIdx.x * 64 + threadIdx.x);
    if (i >= a.n)
        return;
    const rox_aim pb = a.probs[i];

    c.check_ap = false; c.intersect_obj = true; c.filter_ph = false;    // raytrace.py:51-80, 83-99
    c.first_surf = 1; c.last_surf = N - 2;
    c.eps = a.eps; c.fuzz = 1e-5;
    c.probe_surf = pb.surf;
    SegOut so{nullptr, 0, 0};
    const v3 pt0{pb.pt0[0], pb.pt0[1], pb.pt0[2]};

    // the last trial ray evaluated -- the reference's `rr` (trace.py:334-346, 884-905), which
    // iterate_ray_raw's caller reads the reverse chief ray from (wideangle.py:646-651)
    double last_x = 0., last_y = 0.;
    int last_st = -1;
    auto keep_last = [&]() {
        if (a.last_xy) {
            a.last_xy[2 * i] = last_x;
            a.last_xy[2 * i + 1] = last_y;
            a.last_status[i] = last_st;
        }
    };
    // With a wave per problem the lanes are idle copies of each other: the evaluations that do
    // not depend on one another -- the secant iteration's two starting values, hybrd's value at
    // the starting point together with the two forward-difference points of its Jacobian --
    // are traced in ONE pass, lane k tracing the k-th point, and then taken in the reference's
    // order (the same values, and the same `last trial ray' if one of them stops the search).
    const bool wave = a.wave_per_problem != 0;
    const int lane = threadIdx.x;
    // one trial ray through (x1, y1) of the entrance pupil plane: status, whether it failed
    // before surf (the reference's objective raises), its coordinates at surf
    auto trial = [&](double x1, double y1, int &st, int &rs, double &xr, double &yr) {
        const v3 pt1{x1, y1, pb.z_enp};
        v3 dir0 = unit(v3{pt1.x - pt0.x, pt1.y - pt0.y, pt1.z - pt0.z});
        if (pb.flip && dir0.z * pb.z_dir0 < 0)
            dir0 = v3{-dir0.x, -dir0.y, -dir0.z};
        RayEnd e;
        trace_ray<MODE_PROBE, true, FEAT>(c, so, pt0, dir0, pb.wvl_idx, true, e);
        st = e.status;
        rs = 0;
        xr = yr = 0.;                           // final_coord = [0, 0, 0]
        if (e.status != ROX_OK)
            rs = e.fail_surf < pb.surf;
        else {
            xr = e.probe_p.x;
            yr = e.probe_p.y;
        }
    };
    // trace.py:322-349 y_stop_coordinate; `raised`: a trial ray failed before surf
    bool raised = false;
    auto f = [&](double y1) -> double {
        int st, rs;
        double xr, yr;
        trial(0., y1, st, rs, xr, yr);
        last_x = 0.; last_y = y1; last_st = st;
        if (rs)
            raised = true;
        return yr - pb.y_target;
    };
    // f(ya) and then, unless it raised, f(yb)
    auto f_two = [&](double ya, double yb, double &qa, double &qb) {
        if (!wave) {
            qa = f(ya);
            qb = raised ? 0.0 : f(yb);
            return;
        }
        int st, rs;
        double xr, yr;
        trial(0., lane == 1 ? yb : ya, st, rs, xr, yr);
        last_x = 0.; last_y = ya; last_st = __shfl(st, 0);
        qa = __shfl(yr, 0) - pb.y_target;
        qb = 0.0;
        if (__shfl(rs, 0)) {
            raised = true;
            return;
        }
        last_y = yb; last_st = __shfl(st, 1);
        qb = __shfl(yr, 1) - pb.y_target;
        if (__shfl(rs, 1))
            raised = true;
    };

    if (pb.two_d) {
        // trace.py:351-372 surface_coordinate under fsolve (MINPACK hybrd, rox_hybrd.hpp)
        auto f2 = [&](const double *coord, double *fv) -> bool {
            int st, rs;
            double xr, yr;
            trial(coord[0], coord[1], st, rs, xr, yr);
            last_x = coord[0]; last_y = coord[1]; last_st = st;
            if (rs)
                return false;                           // raise ray_error
            fv[0] = xr - pb.x_target;
            fv[1] = yr - pb.y_target;
            return true;
        };
        // f(x) if need_f, f(x + h0 e0), f(x + h1 e1) (MINPACK's fdjac1), in that order.
        // fsolve evaluates f(x0) itself before MINPACK evaluates it again: the same ray twice,
        // traced once here -- a failure stops the search the same way in either call
        auto f2_points = [&](const double *x, const double *h, bool need_f, double *fv,
                             double *cols) -> bool {
            if (!wave) {
                if (need_f && !f2(x, fv))
                    return false;
                for (int j = 0; j < 2; ++j) {
                    double xx[2] = {x[0], x[1]};
                    xx[j] = x[j] + h[j];
                    if (!f2(xx, cols + 2 * j))
                        return false;
                }
                return true;
            }
            // lane 0 (and the lanes above 2): x, when it is asked for; lane 1: column 0; lane 2: column 1
            const int v = lane == 2 ? 2 : ((lane == 1 || !need_f) ? 1 : 0);
            const double cx = v == 1 ? x[0] + h[0] : x[0];
            const double cy = v == 2 ? x[1] + h[1] : x[1];
            int st, rs;
            double xr, yr;
            trial(cx, cy, st, rs, xr, yr);
            for (int k = need_f ? 0 : 1; k < 3; ++k) {
                last_x = k == 1 ? x[0] + h[0] : x[0];
                last_y = k == 2 ? x[1] + h[1] : x[1];
                last_st = __shfl(st, k);
                if (__shfl(rs, k))
                    return false;
                double *out = k == 0 ? fv : cols + 2 * (k - 1);
                out[0] = __shfl(xr, k) - pb.x_target;
                out[1] = __shfl(yr, k) - pb.y_target;
            }
            return true;
        };
        double x[2] = {0., 0.};
        int nfev = 0;
        const int info = hybrd::solve<2>(f2, f2_points, x, 1.49012e-8, 600, pb.epsfcn, 100.0, nfev);
        if (info < 0) {                                 // except TraceError: start_coords = [0, 0]
            x[0] = x[1] = 0.0;
            a.result[i] = ROX_AIM_TRACE_ERROR;
        } else {
            a.result[i] = info == 1 ? ROX_AIM_CONVERGED : ROX_AIM_NOT_CONVERGED;
        }
        a.aim_xy[2 * i] = x[0];
        a.aim_xy[2 * i + 1] = x[1];
        keep_last();
        return;
    }
    const double tol = 1.48e-8;
    double p0 = 0.0, p = 0.0;
    int result = ROX_AIM_NOT_CONVERGED;
    const double eps = 1e-4;
    double p1 = p0 * (1 + eps);
    p1 += (p1 >= 0 ? eps : -eps);
    double q0, q1;
    f_two(p0, p1, q0, q1);
    if (!raised) {
        if (fabs(q1) < fabs(q0)) {
            double t = p0; p0 = p1; p1 = t;
            t = q0; q0 = q1; q1 = t;
        }
        for (int itr = 0; itr < 50; ++itr) {
            if (q1 == q0) {
                p = (p1 + p0) / 2.0;
                break;                          // _ECONVERR, but the root is still used
            }
            if (fabs(q1) > fabs(q0))
                p = (-q0 / q1 * p1 + p0) / (1 - q0 / q1);
            else
                p = (-q1 / q0 * p0 + p1) / (1 - q1 / q0);
            // np.isclose(p, p1, rtol=0, atol=tol)
            const bool close = (isfinite(p) && isfinite(p1)) ? (fabs(p - p1) <= tol) : (p == p1);
            if (close) {
                result = ROX_AIM_CONVERGED;
                break;
            }
            p0 = p1; q0 = q1;
            p1 = p;
            q1 = f(p1);
            if (raised)
                break;
        }
    }
    if (raised) {                               // trace.py:395-396: start_y = 0.0
        p = 0.0;
        result = ROX_AIM_TRACE_ERROR;
    }
    a.aim_xy[2 * i] = 0.0;
    a.aim_xy[2 * i + 1] = p;
    a.result[i] = result;
    keep_last();
}


// scipy.optimize.newton, secant branch (scipy/optimize/_zeros_py.py), disp=False,
// rtol=0, maxiter=50: returns the root estimate; `converged` as scipy flags it;
// `raised` is set by f when the reference's objective would raise -- the
// iteration stops at once.
// `f_two(pa, pb, qa, qb)`: f(pa) and then, unless it raised, f(pb) -- the two starting values,
// which a caller with idle lanes traces in one pass.
template <class F, class F2>
__device__ __forceinline__ double secant(F &f, F2 &f_two, double x0, double tol, bool &raised,
                                         bool &converged)
{
    converged = false;
    double p0 = x0, p = x0;
    const double eps = 1e-4;
    double p1 = x0 * (1 + eps);
    p1 += (p1 >= 0 ? eps : -eps);
    double q0, q1;
    f_two(p0, p1, q0, q1);
    if (raised)
        return p;
    if (fabs(q1) < fabs(q0)) {
        double t = p0; p0 = p1; p1 = t;
        t = q0; q0 = q1; q1 = t;
    }
    for (int itr = 0; itr < 50; ++itr) {
        if (q1 == q0) {
            p = (p1 + p0) / 2.0;
            return p;                           // _ECONVERR, but the root is still used
        }
        if (fabs(q1) > fabs(q0))
            p = (-q0 / q1 * p1 + p0) / (1 - q0 / q1);
        else
            p = (-q1 / q0 * p0 + p1) / (1 - q1 / q0);
        // np.isclose(p, p1, rtol=0, atol=tol)
        const bool close = (isfinite(p) && isfinite(p1)) ? (fabs(p - p1) <= tol) : (p == p1);
        if (close) {
            converged = true;
            return p;
        }
        p0 = p1; q0 = q1;
        p1 = p;
        q1 = f(p1);
        if (raised)
            return p;
    }
    return p;
}

// rayoptics/raytr/vigcalc.py:259-340 calc_vignetted_ray + :396-461 iterate_pupil_ray,
// one lane per (field, pupil direction)
template <int FEAT>
__global__ void __launch_bounds__(64) vig_kernel(const VigArgs a)
{
    const int N = a.n_ifcs;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    // (only the checked trace, pt_inside_fuzz = 1e-4, tests apertures here)
    auto c = search_ctx<FEAT>(a, lds, 1e-4);
    const int i = a.wave_per_problem ? (int)blockIdx.x : (int)(blockIdx.x * 64 + threadIdx.x);
    if (i >= a.n)
        return;
    const bool wave = a.wave_per_problem != 0;
    const int lane = threadIdx.x;
    rox_vig pb;
    if (a.iters) {              // rox_iterate_pupil_rays: the field, the axis, the wavelength
        const rox_pupil_iter it = a.iters[i];
        pb = rox_vig{};
        pb.fld = it.fld;
        pb.xy = it.xy;
        pb.wvl_idx = it.wvl_idx;
    } else {
        pb = a.probs[i];
    }
    const int xy = pb.xy;
    // (selected once: indexing the problem record with a run-time xy would keep it in scratch)
    const double unit_xy = xy == 0 ? pb.unit_dir[0] : pb.unit_dir[1];
    const double start_xy = xy == 0 ? pb.start_dir[0] : pb.start_dir[1];

    c.filter_ph = false;
    c.intersect_obj = pb.fld.kind != ROX_FLD_EPD_WIDE && pb.fld.z_dir0 != 0.0;     // trace.py:302-303
    c.first_surf = 1; c.last_surf = N - 2;
    c.eps = a.eps;
    SegOut so{nullptr, 0, 0};

    // trace_base(opm, (px, py), fld, wvl, apply_vignetting=False, check_apertures=check)
    auto trace = [&](double px, double py, bool check, int probe, RayEnd &e) {
        c.check_ap = check;
        c.fuzz = check ? 1e-4 : 1e-5;           // pt_inside_fuzz=1e-4 on the checked trace
        c.probe_surf = probe;
        v3 pt0, dir0;
        ray_start(pb.fld, 0u, px, py, pt0, dir0);
        trace_ray<MODE_PROBE, true, FEAT>(c, so, pt0, dir0, pb.wvl_idx, true, e);
    };
    // ifcs[s].edge_pt_target(start_dir)[xy]: surface.py:210-218, 422-427, 459-464,
    // interface.py:94-111 (the first clear aperture that is not an obscuration)
    // (read from the table in global memory: the staged copy's circular radii have become the
    // aperture test's thresholds, stage_aperture_thresholds())
    auto edge = [&](int s) -> double {
        tblp row = a.rows + (size_t)s * kRowDoubles;
        const int n_ap = ints_of(row)[3];
        tblp ap = row + (offsetof(rox_surface, ap) / sizeof(double));
        for (int k = 0; k < n_ap; ++k, ap += sizeof(rox_aperture) / sizeof(double)) {
            if (ints_of(ap)[1])
                continue;
            if (ints_of(ap)[0] == ROX_AP_CIRCULAR)
                return ap[3] * unit_xy;
            return (xy == 0 ? ap[3] : ap[4]) * unit_xy;
        }
        return row[offsetof(rox_surface, max_aperture) / sizeof(double)] * unit_xy;
    };
    // iterate_pupil_ray(opm, indx, xy, start_r0, r_target, fld, wvl)
    auto iterate = [&](int indx, double start_r0, double r_target) -> double {
        bool raised = false, conv;
        double raised_x = 0.0;
        // r_pupil_coordinate, :418-446: whether the objective raises for this ray, its value
        auto rpc = [&](double x, int &stop, double &val) {
            RayEnd e;
            trace(xy == 0 ? x : 0., xy == 1 ? x : 0., false, indx, e);
            stop = 0;
            val = 0.0;
            if (e.status != ROX_OK) {
                stop = (e.status == ROX_MISSED_SURFACE) ? (e.fail_surf <= indx)
                                                        : (e.fail_surf < indx);
                if (stop)
                    return;
            }
            // ray_pkg[mc.ray][indx][mc.p]: the partial packet ends with inc_pt at the
            // failing surface
            const v3 p = (e.status != ROX_OK && e.fail_surf == indx) ? e.inc : e.probe_p;
            const double r_ray = copysign(sqrt(p.x * p.x + p.y * p.y), r_target);
            val = r_ray - r_target;
        };
        auto f = [&](double x) -> double {
            int stop;
            double val;
            rpc(x, stop, val);
            if (stop) {
                raised = true;
                raised_x = x;
            }
            return val;
        };
        // the secant iteration's two starting values: with a wave per problem lane 1 traces
        // the second while the others trace the first
        auto f_two = [&](double xa, double xb, double &qa, double &qb) {
            if (!wave) {
                qa = f(xa);
                qb = raised ? 0.0 : f(xb);
                return;
            }
            int stop;
            double val;
            rpc(lane == 1 ? xb : xa, stop, val);
            qa = __shfl(val, 0);
            qb = 0.0;
            if (__shfl(stop, 0)) {
                raised = true;
                raised_x = xa;
                return;
            }
            qb = __shfl(val, 1);
            if (__shfl(stop, 1)) {
                raised = true;
                raised_x = xb;
            }
        };
        const double root = secant(f, f_two, start_r0, 1e-6, raised, conv);
        return raised ? 0.9 * raised_x : root;  // :456-459
    };

    if (a.iters) {              // vigcalc.iterate_pupil_ray alone (set_pupil, vigcalc.py:141-143)
        a.vig[i] = iterate(a.iters[i].indx, a.iters[i].start_r0, a.iters[i].r_target);
        return;
    }
    double rel[2] = {pb.start_dir[0], pb.start_dir[1]};
    int clip = -1;                              // None
    bool iterating = true;
    for (int it = 0; iterating && it < pb.max_iter; ++it) {
        RayEnd e;
        trace(rel[0], rel[1], true, -1, e);
        int indx;
        if (e.status != ROX_OK) {
            indx = e.fail_surf;
            if (indx == clip) {
                iterating = false;
                continue;
            }
        } else {
            if (clip >= 0 || pb.stop_surf < 0) {
                iterating = false;
                continue;
            }
            indx = pb.stop_surf;                // first pass: go to the edge of the stop
        }
        const double r = iterate(indx, xy == 0 ? rel[0] : rel[1], edge(indx));
        rel[0] = xy == 0 ? r : 0.0;
        rel[1] = xy == 0 ? 0.0 : r;
        clip = indx;
    }
    a.vig[i] = 1.0 - ((xy == 0 ? rel[0] : rel[1]) / start_xy);
    a.clip[i] = clip;
}

// scipy.optimize.newton's secant branch with rtol (disp = False); see secant() above for
// the rtol = 0 form the vignetting search uses.  `f` never raises here.
// `f_two(pa, pb, qa, qb)` = f(pa), then f(pb): the two starting values, in one pass of the wave.
template <class F, class F2>
__device__ __forceinline__ double secant_rtol(F &f, F2 &f_two, double x0, double tol, double rtol,
                                              bool &converged)
{
    converged = false;
    double p0 = x0, p = x0;
    const double eps = 1e-4;
    double p1 = x0 * (1 + eps);
    p1 += (p1 >= 0 ? eps : -eps);
    double q0, q1;
    f_two(p0, p1, q0, q1);
    if (fabs(q1) < fabs(q0)) {
        double t = p0; p0 = p1; p1 = t;
        t = q0; q0 = q1; q1 = t;
    }
    for (int itr = 0; itr < 50; ++itr) {
        if (q1 == q0)
            return (p1 + p0) / 2.0;
        if (fabs(q1) > fabs(q0))
            p = (-q0 / q1 * p1 + p0) / (1 - q0 / q1);
        else
            p = (-q1 / q0 * p0 + p1) / (1 - q1 / q0);
        const bool close = (isfinite(p) && isfinite(p1)) ? (fabs(p - p1) <= tol + rtol * fabs(p1))
                                                         : (p == p1);
        if (close) {
            converged = true;
            return p;
        }
        p0 = p1; q0 = q1;
        p1 = p;
        q1 = f(p1);
    }
    return p;
}

// scipy.optimize.brentq's core (scipy/optimize/Zeros/brentq.c): err = -1 when f(a) and f(b)
// have the same sign (scipy raises ValueError)
template <class F>
__device__ __forceinline__ double brentq(F &f, double xa, double xb, double xtol, double rtol,
                                         int iter, int &err)
{
    double xpre = xa, xcur = xb;
    double xblk = 0., fpre, fcur, fblk = 0., spre = 0., scur = 0., sbis;
    double delta, stry, dpre, dblk;
    fpre = f(xpre);
    fcur = f(xcur);
    err = 0;
    if (fpre == 0)
        return xpre;
    if (fcur == 0)
        return xcur;
    if (signbit(fpre) == signbit(fcur)) {
        err = -1;
        return 0.;
    }
    for (int i = 0; i < iter; ++i) {
        if (fpre != 0 && fcur != 0 && (signbit(fpre) != signbit(fcur))) {
            xblk = xpre;
            fblk = fpre;
            spre = scur = xcur - xpre;
        }
        if (fabs(fblk) < fabs(fcur)) {
            xpre = xcur; xcur = xblk; xblk = xpre;
            fpre = fcur; fcur = fblk; fblk = fpre;
        }
        delta = (xtol + rtol * fabs(xcur)) / 2;
        sbis = (xblk - xcur) / 2;
        if (fcur == 0 || fabs(sbis) < delta)
            return xcur;
        if (fabs(spre) > delta && fabs(fcur) < fabs(fpre)) {
            if (xpre == xblk) {
                stry = -fcur * (xcur - xpre) / (fcur - fpre);
            } else {
                dpre = (fpre - fcur) / (xpre - xcur);
                dblk = (fblk - fcur) / (xblk - xcur);
                stry = -fcur * (fblk * dblk - fpre * dpre) / (dblk * dpre * (fblk - fpre));
            }
            const double lim = fmin(fabs(spre), 3 * fabs(sbis) - delta);
            if (2 * fabs(stry) < lim) {
                spre = scur; scur = stry;
            } else {
                spre = sbis; scur = sbis;
            }
        } else {
            spre = sbis; scur = sbis;
        }
        xpre = xcur; fpre = fcur;
        if (fabs(scur) > delta)
            xcur += scur;
        else
            xcur += (sbis > 0 ? delta : -delta);
        fcur = f(xcur);
    }
    err = -2;
    return xcur;
}

// rayoptics/raytr/wideangle.py:96-292 find_real_enp_rev1 + :295-315 find_edge + :317-427
// find_z_enp_on_interval, one lane per (field, wavelength); every trial ray is
// enp_z_coordinate (:46-83).  tuples-or-None of the reference are (value, have_*) pairs here.
template <int FEAT>
__global__ void __launch_bounds__(64) enp_kernel(const EnpArgs a)
{
    const int N = a.n_ifcs;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    auto c = search_ctx<FEAT>(a, lds, -1.0);
    // One WAVE per problem (round 4).  With one lane per problem the lanes of a wave sit in
    // different phases of the search -- the walk, find_edge, the secant iteration, brentq are
    // different call sites of the trace -- and the wave executes their union one after the
    // other (48 Nikkor fields: 5.9 ms for searches of 3-25 trial rays each).  Here every lane of
    // the wave follows the same problem: the sequential decisions are wave-uniform, and the
    // one phase whose trial rays do not depend on each other -- the sampled walk from the
    // paraxial pupil -- is traced by the 64 lanes at once and then REPLAYED in the reference's
    // order, so the outcome (every result code included) is unchanged.
    const int i = blockIdx.x;
    const int lane = threadIdx.x;
    const rox_enp pb = a.probs[i];
    __shared__ double s_z[64], s_h[64];
    __shared__ int s_st[64], s_fs[64];

    c.check_ap = false; c.intersect_obj = false; c.filter_ph = false;   // intersect_obj=False
    c.first_surf = 1; c.last_surf = N - 2;
    c.eps = a.eps; c.fuzz = 1e-5;
    c.probe_surf = pb.surf;
    SegOut so{nullptr, 0, 0};
    const v3 dir0{pb.dir0[0], pb.dir0[1], pb.dir0[2]};

    // the last trial ray (the reference's `rr`)
    double z_last = 0.;
    RayEnd last;
    last.status = ROX_OK;
    last.fail_surf = -1;
    // enp_z_coordinate: true if the ray got through; ht = final_coord[1]
    auto trial = [&](double z_enp, double &ht) -> bool {
        const double obj2enp_dist = (pb.obj_dist + z_enp);
        const v3 pt1{0., 0., obj2enp_dist};
        const v3 m = rotate(pb.rot, pb.rot_order, v3{-pt1.x, -pt1.y, -pt1.z});
        const v3 pt0{m.x + pt1.x, m.y + pt1.y, m.z + pt1.z};
        trace_ray<MODE_PROBE, true, FEAT>(c, so, pt0, dir0, pb.wvl_idx, true, last);
        z_last = z_enp;
        if (last.status != ROX_OK) {
            ht = 0.;
            return false;
        }
        ht = last.probe_p.y;
        return true;
    };
    auto eval = [&](double z) -> double {
        double ht;
        (void)trial(z, ht);
        return ht - 0.;
    };
    // trial rays that do not depend on one another, traced in one pass -- lane k traces the
    // k-th -- and then taken in the reference's order: take(k, z) makes lane k's ray the
    // `last trial ray' of every lane (what last_ht and the walk read of it)
    auto take = [&](int k, double z) {
        last.status = __shfl(last.status, k);
        last.fail_surf = __shfl(last.fail_surf, k);
        last.probe_p.y = __shfl(last.probe_p.y, k);
        last.inc.y = __shfl(last.inc.y, k);
        z_last = z;
    };
    auto eval_two = [&](double za, double zb, double &qa, double &qb) {
        double ht;
        (void)trial(lane == 1 ? zb : za, ht);
        qa = __shfl(ht, 0) - 0.;
        qb = __shfl(ht, 1) - 0.;
        take(1, zb);
    };
    // rr.pkg.ray[stop_idx][mc.p][mc.y]: a failed ray's partial packet reaches the stop only if
    // it failed behind it, or at it with an incident point (anything but a missed surface)
    auto last_ht = [&](bool &raises) -> double {
        if (last.status == ROX_OK || last.fail_surf > pb.surf)
            return last.probe_p.y;
        if (last.fail_surf == pb.surf && last.status != ROX_MISSED_SURFACE)
            return last.inc.y;
        raises = true;
        return 0.;
    };
    auto find_edge = [&](double ea, double eb, int max_iter, double &z_edge, double &h_edge) {
        double fa, fb, fc, h_mine;
        const bool ok_mine = trial(lane == 1 ? eb : ea, h_mine);    // both ends in one pass
        fa = __shfl(h_mine, 0);
        fb = __shfl(h_mine, 1);
        bool okb = __shfl((int)ok_mine, 1);
        take(1, eb);
        if (!(FEAT & F_POLY) && max_iter == 6) {
            // Without Newton code no sample is slow to trace: all 63 midpoints the six
            // bisections can visit are traced at once -- lane L (heap order: 1 the first
            // midpoint, 2 L the next one after a failed ray, 2 L + 1 after a good one) forms
            // its interval by the bisection's own recurrence -- and the decisions replayed.
            double la = ea, lb = eb, lc = ea;
            int depth = 0;
            for (int t = lane; t > 1; t >>= 1)
                ++depth;
            for (int d = depth; d >= 0; --d) {
                lc = la + (lb - la) / 2;
                if (d > 0) {
                    if ((lane >> (d - 1)) & 1)
                        la = lc;
                    else
                        lb = lc;
                }
            }
            const bool ok_l = trial(lane == 0 ? ea : lc, h_mine);   // (lane 0: an idle copy)
            int node = 1;
            for (int k = 0; k < 6; ++k) {
                const double cc = ea + (eb - ea) / 2;
                fc = __shfl(h_mine, node);
                const bool ok = __shfl((int)ok_l, node);
                if (k == 5)
                    take(node, cc);
                if (!ok) {
                    eb = cc; okb = false; fb = fc;
                    node = 2 * node;
                } else {
                    ea = cc; fa = fc;
                    node = 2 * node + 1;
                }
            }
            max_iter = 0;
        }
        for (int k = 0; k < max_iter; ++k) {
            const double cc = ea + (eb - ea) / 2;
            if (!trial(cc, fc)) {
                eb = cc; okb = false; fb = fc;
            } else {
                ea = cc; fa = fc;
            }
        }
        if (!okb) {
            z_edge = ea; h_edge = fa;
        } else {
            z_edge = eb; h_edge = fb;
        }
    };
    auto fuzzy_zero = [](double x) { return fabs(x) < 1e-14; };

    double z_out = 0.;
    int code = ROX_ENP_FOUND;
    double ht;
    bool done = false;
    if (!isnan(pb.aim_info)) {                      // :128-134
        (void)trial(pb.aim_info, ht);
        if (fabs(ht) < 1.48e-08) {
            z_out = pb.aim_info;
            done = true;
        }
    }
    const double z_enp_0 = pb.z_enp_0;
    if (!done && pb.dir0[2] == 1) {                 // :138-141
        (void)trial(z_enp_0, ht);
        z_out = z_enp_0;
        done = true;
    }
    if (!done) {
        bool have_start = false, have_prev = false, have_end = false;
        double start_z = 0, start_h = 0, prev_z = 0, prev_h = 0, end_z = 0, end_h = 0;
        double del_z = -z_enp_0 / 16;
        double z_enp = z_enp_0;
        bool keep_going = true, first = true;
        int first_surf_misses = 0, trials = 0, successes = 0;
        // ---- the walk's samples, all at once.  Every z the walk can visit is an element of
        // one of two chains from z_enp_0 -- steps of +del_z or of -del_z, each formed by the
        // walk's own recurrence (`z += del_z`, a z that is fuzzily zero replaced by del_z / 10)
        // -- because every change of direction restarts from z_enp_0.  Lane L < kSpec traces
        // element L of the forward chain, lane kSpec <= L < 2 kSpec element L - kSpec + 1 of the
        // backward one.  kSpec = 4 (ROX_ENP_SPEC): samples far from the pupil, which the walk
        // itself rarely reaches, can be very slow to trace -- an asphere intersection there may
        // run the Spencer-Murty iteration to the reference's cap of 1000 steps -- and the pass
        // waits for its slowest lane.  Measured, ms for 9 / 37 double-Gauss and 9 / 48 Nikkor
        // problems (profiles/r04_wide_angle_latency.jsonl): round 3 (a lane per problem) 0.65 /
        // 1.27 / 0.55 / 5.87; a wave per problem without the pass 0.51 / 0.90 / 0.42 / 3.13;
        // kSpec 4: 0.47 / 0.80 / 0.32 / 1.79; 8: 0.48 / 0.73 / 1.07 / 2.48; 32: 0.48 / 0.56 /
        // 1.99 / 2.51.  (Dropping speculative samples that exceed a step budget and tracing them
        // again on demand keeps the nine Nikkor fields at 0.34 ms for every kSpec but costs the
        // 48: 3.8 ms -- one of them has such a sample ON its walk, and then pays for it alone
        // instead of beside the others.  That one 1000-step trace, ~1.7 ms, is the floor.)
        // (instances without Newton code have no slow samples: they speculate the whole chain a
        // wave holds -- double Gauss, 37 problems: 0.80 ms with 4, 0.56 ms with 32)
        constexpr int kSpec = (FEAT & F_POLY) ? ROX_ENP_SPEC : 32;
        const double d0 = del_z;
        if (lane < 2 * kSpec) {
            const double dz = lane < kSpec ? d0 : -d0;
            const int kk = lane < kSpec ? lane : lane - kSpec + 1;
            double z = z_enp_0;
            for (int k = 0; k < kk; ++k) {
                z += dz;
                if (fuzzy_zero(z))
                    z = dz / 10;
            }
            double h;
            (void)trial(z, h);
            s_z[lane] = z;
            s_h[lane] = h;
            s_st[lane] = last.status;
            s_fs[lane] = last.fail_surf;
        }
            __syncthreads();
        int chain_k = 0;                            // z_enp is element chain_k of the chain of del_z
        // the walk's trial at the current z_enp: read back when it was traced above
        auto sample = [&](double &h) -> bool {
            const int slot = (del_z == d0) ? (chain_k < kSpec ? chain_k : -1)
                                           : (chain_k == 0 ? 0 : (chain_k <= kSpec ? kSpec - 1 + chain_k : -1));
            if (slot < 0 || s_z[slot] != z_enp)
                return trial(z_enp, h);             // beyond the speculated samples: trace it now
            h = s_h[slot];
            last.status = s_st[slot];
            last.fail_surf = s_fs[slot];
            z_last = z_enp;
            return s_st[slot] == ROX_OK;
        };
        while (keep_going && trials < 64 && first_surf_misses < 2) {
            if (sample(ht)) {
                ++successes;
                if (!have_start) {
                    have_start = true; start_z = z_enp; start_h = ht;
                }
                have_prev = have_end; prev_z = end_z; prev_h = end_h;
                have_end = true; end_z = z_enp; end_h = ht;
                if (successes > 1 && prev_h * end_h < 0)
                    keep_going = false;
                if (successes == 2 && pb.check_direction) {
                    if (fabs(start_h) < fabs(end_h) && first) {
                        del_z = -del_z;
                        z_enp = z_enp_0; chain_k = 0;
                        first = false;
                        double t = end_z; end_z = start_z; start_z = t;
                        t = end_h; end_h = start_h; start_h = t;
                    }
                }
            } else {
                if (last.status == ROX_MISSED_SURFACE && last.fail_surf == 1) {
                    del_z = -del_z;
                    z_enp = z_enp_0; chain_k = 0;
                    ++first_surf_misses;
                }
                if (have_start) {
                    if (first) {
                        del_z = -del_z;
                        z_enp = z_enp_0; chain_k = 0;
                        first = false;
                        double t = end_z; end_z = start_z; start_z = t;
                        t = end_h; end_h = start_h; start_h = t;
                    } else {
                        keep_going = false;
                    }
                }
            }
            z_enp += del_z;
            if (fuzzy_zero(z_enp))
                z_enp = del_z / 10;
            ++chain_k;
            ++trials;
        }
        double ia = 0., ib = 0.;
        if (!have_start) {
            code = ROX_ENP_REFERENCE_RAISES;
            done = true;
        } else {
            const double z_a = start_z, h_a = start_h, z_b = end_z, h_b = end_h;
            if (z_a == z_b) {                       // :208-223
                const double start_new = z_a - del_z, end_new = z_b + del_z;
                have_start = have_end = false;
                const double step = (end_new - start_new) / 7;      // np.linspace(num=8)
                auto zk = [&](int k) { return k == 7 ? end_new : (double)k * step + start_new; };
                double h_mine;
                const bool ok_mine = trial(zk(lane & 7), h_mine);     // the eight samples at once
                for (int k = 0; k < 8; ++k) {
                    const double z = zk(k);
                    ht = __shfl(h_mine, k);
                    if (__shfl((int)ok_mine, k)) {
                        if (!have_start) {
                            have_start = true; start_z = z; start_h = ht;
                        }
                        have_end = true; end_z = z; end_h = ht;
                    }
                }
                take(7, end_new);
                if (!have_start) {
                    code = ROX_ENP_REFERENCE_RAISES;
                    done = true;
                }
                ia = start_z; ib = end_z;
            } else if (h_a * h_b < 0) {
                ia = z_a; ib = z_b;
                if (have_prev && prev_h * h_b < 0) {
                    start_z = prev_z; start_h = prev_h;
                    ia = prev_z; ib = z_b;
                }
            } else {
                double z_eb, h_eb, z_ea, h_ea;
                find_edge(z_b, z_b + del_z, 6, z_eb, h_eb);
                if (h_eb * h_b < 0) {
                    start_z = z_b; start_h = h_b;
                    end_z = z_eb; end_h = h_eb;
                    ia = z_b; ib = z_eb;
                } else {
                    find_edge(z_a, z_a - del_z, 6, z_ea, h_ea);
                    if (h_ea * h_a < 0) {
                        start_z = z_a; start_h = h_a;
                        end_z = z_ea; end_h = h_ea;
                        ia = z_a; ib = z_ea;
                    } else {
                        const double z_cntr = z_ea + (z_eb - z_ea) / 2;
                        (void)trial(z_cntr, ht);
                        z_out = z_b;
                        code = ROX_ENP_NO_CHIEF_RAY;
                        done = true;
                    }
                }
            }
        }
        if (!done) {
            double z_estimate;
            if (fuzzy_zero(end_h - start_h))
                z_estimate = start_z;
            else
                z_estimate = start_z - ((end_z - start_z) / (end_h - start_h)) * start_h;
            bool converged, raises = false;
            double z = secant_rtol(eval, eval_two, z_estimate, 1.48e-8, 1e-7, converged);
            const double ht_at_stop = last_ht(raises);
            if (!raises && fabs(ht_at_stop - 0.) < 1e-6)
                converged = true;
            if (!raises && !converged) {
                int err;
                z = brentq(eval, ia, ib, 2e-12, 1e-7, 100, err);
                if (err == -1)
                    raises = true;
            }
            if (!raises)
                (void)last_ht(raises);
            if (raises)
                code = ROX_ENP_REFERENCE_RAISES;
            else
                z_out = z;
        }
    }
    if (lane == 0) {
        a.z_out[2 * i] = z_out;
        a.z_out[2 * i + 1] = z_last;
        a.result[i] = code;
    }
}

template <int FEAT>
void launch_enp_instance(const EnpArgs &a, size_t lds, hipStream_t st)
{
    if (lds > kDefaultDynLds)
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(enp_kernel<FEAT>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(enp_kernel<FEAT>, dim3(a.n), dim3(64), lds, st, a);     // one wave per problem
}

template <int FEAT>
void launch_vig_instance(const VigArgs &a, size_t lds, hipStream_t st)
{
    if (lds > kDefaultDynLds)
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(vig_kernel<FEAT>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(vig_kernel<FEAT>, dim3(a.wave_per_problem ? a.n : (a.n + 63) / 64), dim3(64), lds, st, a);
}

template <int FEAT>
void launch_aim_instance(const AimArgs &a, size_t lds, hipStream_t st)
{
    if (lds > kDefaultDynLds)
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(aim_kernel<FEAT>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(aim_kernel<FEAT>, dim3(a.wave_per_problem ? a.n : (a.n + 63) / 64), dim3(64), lds, st, a);
}

// what a translation unit csrc/search_<name>.hip defines for its instance
#define ROX_SEARCH_INSTANCE(name, FEAT)                                                          \
    void launch_aim_##name(const AimArgs &a, size_t lds, hipStream_t st) { launch_aim_instance<FEAT>(a, lds, st); } \
    void launch_enp_##name(const EnpArgs &a, size_t lds, hipStream_t st) { launch_enp_instance<FEAT>(a, lds, st); } \
    void launch_vig_##name(const VigArgs &a, size_t lds, hipStream_t st) { launch_vig_instance<FEAT>(a, lds, st); }

}  // namespace rox
