// rox_device.hpp -- device code of the sequential real-ray trace for gfx950
// (MI355X, CDNA4): arithmetic, surface intersection, phase elements, packet
// stores and the trace kernel template.  Included by every kernel-instance
// translation unit (inst_*.hip) and by the host side (roxtrace.hip).
//
// Hot path of mjhoptics/ray-optics restated as hand-written HIP (reference
// paths relative to /root/reference/src/):
//   rayoptics/raytr/raytrace.py:83-264   trace_raw   -> trace_ray() (the loop)
//   rayoptics/raytr/raytrace.py:19-38    bend/reflect -> refract(), mirror()
//   rayoptics/raytr/raytrace.py:41-48, 205-210 phase  -> apply_phase()
//   rayoptics/oprops/doe.py:124-175      DiffractionGrating.phase_ludwig
//   rayoptics/oprops/doe.py:28-54, 272-323 DiffractiveElement.phase + radial_phase_fct
//   rayoptics/oprops/doe.py:375-397      HolographicElement.phase (every ThinLens)
//   rayoptics/oprops/thinlens.py:128-136 ThinLens.normal/intersect
//   rayoptics/elem/profiles.py:310-336   Spherical.intersect  \  quadric_hit()
//   rayoptics/elem/profiles.py:569-593   Conic.intersect      /
//   rayoptics/elem/profiles.py:155-186   intersect_spencer    -> newton_hit()
//   rayoptics/elem/profiles.py:849-885   EvenPolynomial.sag/df \ poly_eval()
//   rayoptics/elem/profiles.py:1070-1113 RadialPolynomial.sag/df/
//   rayoptics/elem/profiles.py:1317-1437 Y/XToroid            /
//   rayoptics/elem/surface.py:198-208, 416-457 point_inside    -> inside_aperture_list()
//   rayoptics/raytr/opticalspec.py:289-400, 1339-1353; trace.py:298-308
//                                         pupil -> (pt0, dir0) -> ray_start()
//
// Execution model: one wavefront lane = one ray; 512-thread workgroups take
// tiles of 512 consecutive rays.  Every per-surface parameter is wave-uniform:
// the surface table is staged once per workgroup in LDS and read with
// same-address (broadcast, conflict-free) ds_reads.  Ray packets are SoA
// [segment][component][ray]: each store is 64 lanes x 8 B = 512 B contiguous.
// All arithmetic is IEEE binary64 with the reference's operation order: this
// code is compiled with -ffp-contract=off, NumPy's BLAS dot sites are spelled
// as explicit fma chains (dot3), division and sqrt are the correctly rounded
// ones.  No MFMA: this is 3-vector arithmetic, not a contraction.
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>
#include <cstring>
#include <type_traits>

#include "../../include/roxtrace.h"

#pragma clang fp contract(off)

// ---- build-time knobs (defaults = the shipped configuration; the others are
// kept for A/B measurement with tools/ab_bench.py) -------------
#ifndef ROX_MIN_WAVES        // __launch_bounds__ second argument (waves per SIMD)
#define ROX_MIN_WAVES 4       // 128 VGPRs
#endif
#ifndef ROX_STORE_NT         // 1: non-temporal packet stores (measured: FULL 236 us vs 257 us)
#define ROX_STORE_NT 1
#endif
#ifndef ROX_SLIM_FP64        // 1: range-guarded slim sqrt / shared-reciprocal division triples
#define ROX_SLIM_FP64 1       //    (bit-identical to sqrt() and `/`; see slim_* below)
#endif
#ifndef ROX_IDENT_RT         // 1: interfaces whose rotation is exactly the identity skip the dgemv chains
#define ROX_IDENT_RT 1        //    (HITS 117.0 -> 111.2 us; FULL keeps the chains: 190.8 vs 194.5 us)
#endif
#ifndef ROX_UNIT_FUSED       // 1: unit() decides once for its sqrt and its three quotients
#define ROX_UNIT_FUSED 1      //    (6 fewer band-test instructions per surface: HITS 120.5 -> 117.1 us)
#endif
#ifndef ROX_BLOCK            // workgroup size of the reduced-output modes (HITS sustained: 512 -> 124 us,
#define ROX_BLOCK 512         // 1024 -> 131 us)
#endif
#ifndef ROX_BLOCK_POLY       // ... of the instances that carry Newton code
#define ROX_BLOCK_POLY ROX_BLOCK
#endif
#ifndef ROX_MIN_WAVES_POLY   // waves per SIMD the reduced-output modes of those instances are compiled for
#define ROX_MIN_WAVES_POLY ROX_MIN_WAVES
#endif
#ifndef ROX_KERNEL_ALIGN     // alignment of the trace kernels' code (0: the compiler's 256 bytes)
#define ROX_KERNEL_ALIGN 0
#endif
#if ROX_KERNEL_ALIGN > 0
#define ROX_KERNEL_ALIGNED __attribute__((aligned(ROX_KERNEL_ALIGN)))
#else
#define ROX_KERNEL_ALIGNED
#endif
#ifndef ROX_MIN_WAVES_APLIST_REDUCED  // waves per SIMD the HITS kernels of the aperture-list instance
#define ROX_MIN_WAVES_APLIST_REDUCED 8 // (spherical tables with clear-aperture lists: every Zemax import without
#endif                                 // aspheres -- BASELINE configs[3], configs[4]) are compiled for.  At the
                                       // default 4 hipcc takes 79 VGPRs; held to 64 (56 B of scratch per lane) or 72
                                       // the same source is 8 % faster: lithography lens HITS 421 -> 386 us per 2^20
                                       // rays, RC telescope 46.6 -> 42.9 (7: 389 / 42.9); the lean instance gains
                                       // nothing from it (111.0 -> 113.6) and FULL loses 7-12 %.  profiles/r05_min_waves.txt
#ifndef ROX_MIN_WAVES_FAST   // waves per SIMD the tolerance-mode (F_FAST) instances are compiled for
#define ROX_MIN_WAVES_FAST ROX_MIN_WAVES
#endif
#ifndef ROX_BLOCK_SMALL      // workgroup size of launches that do not fill the chip (block_of() below)
#define ROX_BLOCK_SMALL 256
#endif
#ifndef ROX_BLOCK_COMPACT    // workgroup = tile size of HITS_COMPACT: fewer, larger tiles make the
#define ROX_BLOCK_COMPACT 1024 // look-back cheaper (into pinned memory 128 -> 501, 256 -> 371,
#endif                        // 512 -> 337, 1024 -> 316 us)
#ifndef ROX_COMPACT_DEFER    // 1: HITS_COMPACT tiles after a workgroup's first look back and copy one tile
#define ROX_COMPACT_DEFER 0   //    late (measured: nothing at 1024-ray tiles -- 156 vs 159 us into HBM -- and the
#endif                        //    output leaves later: 332 vs 301 us into pinned memory)
#ifndef ROX_MIN_WAVES_COMPACT_LEAN   // waves per SIMD the lean / aplist HITS_COMPACT instances are compiled for
#define ROX_MIN_WAVES_COMPACT_LEAN ROX_MIN_WAVES
#endif
#ifndef ROX_BLOCK_FULL       // workgroup size of FULL mode: with the per-surface barrier the whole
#define ROX_BLOCK_FULL 1024   // workgroup writes its packet rows together (sustained 256 -> 215 us,
#endif                        // 512 -> 198 us, 1024 -> 192.5 us; without the barrier 217 us)
#ifndef ROX_FULL_EARLY_STORES    // 1: FULL packets: a segment's point and normal are stored as soon as they
#define ROX_FULL_EARLY_STORES 1  //    are known (after the intersection / the normalisation) instead of with the
#endif                           //    direction at the end of the iteration: the workgroup's ten row pieces per
                                 //    surface leave in three bursts instead of one -- the store-bound double Gauss
                                 //    192.0 -> 188.2 us per 2^20 rays (5.24 -> 5.34 TB/s), .zmx zoom and phone lens
                                 //    unchanged, lithography lens 669 -> 666 (gpurun_out ab_full, two rounds)
#ifndef ROX_WG_SYNC          // 1: FULL mode: a workgroup barrier per surface keeps the waves of a
#define ROX_WG_SYNC 1         //    workgroup on the same packet rows (218 -> 202 us)
#endif
// FULL mode of the instances that carry Newton code (aspheres, toroids): the iteration counts
// differ per wave, so a 1024-thread workgroup (the only one its CU holds at 99-125 VGPRs) waits
// at every surface for its slowest wave with nothing else to run; four 256-thread workgroups
// per CU, out of step with each other, avoid that (round 3, when these instances always ran
// 256: phone lens 319 -> 290 us).  Since round 5 the choice is made per SYSTEM at launch
// (roxtrace.hip want_small(): by the share of Newton interfaces): the large workgroup is the
// regular one here too and the ROX_BLOCK_SMALL kernels are the alternative.
#ifndef ROX_BLOCK_FULL_POLY
#define ROX_BLOCK_FULL_POLY 1024
#endif
#ifndef ROX_WG_SYNC_POLY     // barrier per surface also there (off: 286 / 483 / 195 -- within the noise)
#define ROX_WG_SYNC_POLY 1
#endif
#ifndef ROX_MIN_WAVES_FULL_POLY  // 5 caps the VGPRs at 96 (five workgroups per CU) at the price of 12-28 B
#define ROX_MIN_WAVES_FULL_POLY ROX_MIN_WAVES  // of scratch per lane: phone lens 273, but the .zmx zoom 386 us
#endif
#ifndef ROX_POLY_SAME_BRANCH  // 1: sqrt(rad_e) of an EvenPolynomial only when (cc + 1.0) != ec -- a real
#define ROX_POLY_SAME_BRANCH 1 //    wave-uniform branch.  Written as a select, hipcc evaluated the second (full,
#endif                         //    ~23 VALU incl. a quarter-rate v_rsq_f64) square root in EVERY evaluation
#ifndef ROX_POLY_SHARED_POW   // 1: the sag and slope series of EvenPolynomial / toroid profiles share one
#define ROX_POLY_SHARED_POW 1  //    power chain (r2^i of the slope loop IS r2^(i-1) * r2 of the sag loop, bit for bit)
#endif
#ifndef ROX_NEWTON_SLIM_DIV   // 1: the Spencer-Murty quotient f / dot(d, df) through slim_div()
#define ROX_NEWTON_SLIM_DIV 1
#endif
#ifndef ROX_NEWTON_UNROLL     // Spencer-Murty steps spelled straight-line before the residual loop.  Rounds 1-4
#define ROX_NEWTON_UNROLL 0    // shipped 4 (six inlined evaluations per asphere site); the plain loop -- two --
#endif                         // is a third of the code and faster: Nikkor HITS 329 -> 314 us, .zmx 143.2 -> 141.5
#ifndef ROX_IDENT_RT_FULL_POLY   // 1: the identity-rotation short cut also in the FULL mode of those
#define ROX_IDENT_RT_FULL_POLY 1 //    instances, which are closer to their VALU bound (283 / 473 / 187 us)
#endif

namespace rox {

constexpr int kBlock = ROX_BLOCK;
constexpr int kWaves = kBlock / 64;
// threads per workgroup (= rays per tile) of an output mode of a feature instance
// (feat & kFeatNewton: the instance carries Newton code -- F_EVEN | F_RADIAL | F_TOROID)
constexpr int kFeatNewton = 1 | 2 | 4;
constexpr int kFeatFast = 64;       // F_FAST: a tolerance-mode instance (ROX_FAST_FP64; reduced-output modes only)
// `small`: the launch does not fill the chip several times over (roxtrace.hip want_small()): it
// runs in workgroups of ROX_BLOCK_SMALL threads, which spread over the CUs wave by wave instead
// of sixteen (FULL) or eight waves at a time.  A CU holds 5 waves per SIMD of the lean FULL
// instance (84 VGPRs) but only ONE 1024-thread workgroup (4 per SIMD); BASELINE configs[3]
// (5 x 256^2 rays = 5120 waves) is 320 such workgroups on 256 CUs -- two rounds -- and 1280
// workgroups of 256 threads on 1280 slots: one.
constexpr int block_of(int out_mode, int feat, bool small = false)
{
    return out_mode == ROX_OUT_HITS_COMPACT ? ROX_BLOCK_COMPACT
         : small ? ROX_BLOCK_SMALL
         : out_mode == ROX_OUT_FULL ? ((feat & kFeatNewton) ? ROX_BLOCK_FULL_POLY : ROX_BLOCK_FULL)
         : ((feat & kFeatNewton) ? ROX_BLOCK_POLY : ROX_BLOCK);
}
// whether a distinct small-workgroup kernel exists for (mode, instance)
constexpr bool has_small(int out_mode, int feat)
{
    return block_of(out_mode, feat, true) != block_of(out_mode, feat, false);
}
constexpr int min_waves_of(int out_mode, int feat, bool small = false)
{
    // (the small-workgroup kernels serve launches that fit the chip once: latency, where the
    // scratch of the tighter budget costs -- configs[3] HITS 18 -> 20 us -- instead of paying)
    return ((feat & kFeatFast) && out_mode != ROX_OUT_FULL) ? ROX_MIN_WAVES_FAST
         : (feat == 8 /* F_APLIST */ && out_mode == ROX_OUT_HITS && !small) ? ROX_MIN_WAVES_APLIST_REDUCED
         : (out_mode == ROX_OUT_FULL && (feat & kFeatNewton)) ? ROX_MIN_WAVES_FULL_POLY
         : (out_mode == ROX_OUT_HITS_COMPACT && !(feat & ~8)) ? ROX_MIN_WAVES_COMPACT_LEAN
         : (out_mode != ROX_OUT_FULL && out_mode != ROX_OUT_HITS_COMPACT && (feat & kFeatNewton))
               ? ROX_MIN_WAVES_POLY
         : ROX_MIN_WAVES;
}
static_assert(sizeof(rox_aperture) == 40, "rox_aperture layout");
static_assert(sizeof(rox_phase) == 168, "rox_phase layout");
static_assert(sizeof(rox_surface) == 576, "rox_surface layout");
static_assert(sizeof(rox_field) == 192, "rox_field layout");
static_assert(sizeof(rox_enp) == 136, "rox_enp layout");
static_assert(sizeof(rox_pupil_iter) == 224, "rox_pupil_iter layout");

// Device-side row = the public rox_surface + per-surface values that are the
// same for every ray and are therefore computed once at rox_system_create:
// dcoefs[i] = c_coef_i * coefs[i], the product the df() loops of the polynomial
// profiles form per evaluation (c_coef_i = 2(i+1), or i+1 for RadialPolynomial;
// exact small integers, so the host product has the reference's rounding).
// cd[] interleaves (coefs[i], dcoefs[i]): the sag and slope series read one
// 16-byte LDS word per term.
struct dev_surface {
    rox_surface pub;
    double cd[2 * ROX_MAX_COEF];
};
constexpr int kRowDoubles = sizeof(dev_surface) / sizeof(double);   // 92
static_assert(offsetof(dev_surface, cd) % 16 == 0 && sizeof(dev_surface) % 16 == 0, "16-byte LDS reads");
// Per (wavelength, interface) constants of a DiffractionGrating, formed on the
// host with libm pow() exactly as the reference's `mu**2` / `T**2` are:
//   [0] mu = n_in/n_out  [1] mu**2  [2] T  [3] T**2          (doe.py:138-143)
constexpr int kPhaseConsts = 4;

enum { GEN_RAYS = 0, GEN_PUPIL = 1 };
enum { AXIS_LIST = 0, AXIS_PRODUCT = 1 };
// internal output mode of the aiming kernel: no packet stores, the intercept
// at TraceArgs.probe_surf is kept
constexpr int MODE_PROBE = 100;

// system features a launch needs; the host picks the leanest instance
enum { F_EVEN = 1,      // some interface is an EvenPolynomial (Newton code)
       F_RADIAL = 2,    // ... a RadialPolynomial
       F_TOROID = 4,    // ... a Y/XToroid
       F_APLIST = 8,    // some interface carries clear_apertures
       F_PHFILT = 16,   // filter_out_phantoms with phantoms present
       F_PHASE = 32 };  // phase elements / thin lenses
constexpr int F_POLY = F_EVEN | F_RADIAL | F_TOROID;
constexpr int F_ALL = F_POLY | F_APLIST | F_PHFILT | F_PHASE;
// instance flavours beside the feature bits: F_FAST = 64 (tolerance mode, declared with its
// arithmetic below), F_GTAB = 128: the table stays in global memory and is read with scalar
// loads (ctblp above) -- the instance of tables beyond the LDS (and, ROX_FAST_GTAB, of the
// tolerance-mode kernels)
constexpr int F_GTAB = 128;
// Which kernels read their table that way (beside the instance for tables beyond the LDS):
// ROX_GTAB_EXACT / ROX_GTAB_FAST, bit masks over {1: the reduced-output modes of the instances
// that carry Newton code, 2: ... of the other instances, 4: FULL packets of the Newton
// instances, 8: FULL of the others}.  A value read by a scalar load sits in SGPRs: the Newton
// instances, whose asphere coefficients and row constants otherwise occupy vector registers
// for the whole evaluation, drop from 91-101 to 72-80 VGPRs -- one or two more resident waves
// per SIMD.  Measured per 2^20 rays (gpurun r06t): .zmx zoom HITS 134 -> 124 us, tolerance mode
// 87 -> 76; Nikkor 291 -> 276, 154 -> 143; phone lens 227 -> 226, 151 -> 142; FULL packets gain
// nothing (bound by their stores; phone lens 267 -> 277): shipped for the reduced-output modes
// of the Newton instances, exact and tolerance mode.  Bit-identical either way.
#ifndef ROX_GTAB_EXACT
#define ROX_GTAB_EXACT 1
#endif
#ifndef ROX_GTAB_FAST
#define ROX_GTAB_FAST 1
#endif
// F_GTAB or 0 for (feature set, tolerance mode, output mode): the kernel the host launches
constexpr int gtab_of(int feat, bool fast, int out_mode)
{
    const int mask = fast ? ROX_GTAB_FAST : ROX_GTAB_EXACT;
    const bool newton = (feat & (1 | 2 | 4)) != 0;
    const int bit = (out_mode == ROX_OUT_FULL ? 4 : 1) << (newton ? 0 : 1);
    return ((feat & F_GTAB) || (mask & bit)) ? F_GTAB : 0;
}

struct v3 { double x, y, z; };
typedef double d2 __attribute__((ext_vector_type(2)));

typedef const double *tblp;     // LDS (generic pointer into __shared__)
typedef const int32_t *tbli;
// The same table left in GLOBAL memory and read through the constant address space (the F_GTAB
// instances): a wave-uniform address in that space is a scalar load (s_load_dwordx2..x16 through
// the scalar cache), the value lands in SGPRs and no LDS is involved -- so a table has no size
// limit (SequentialModel has none: rayoptics/seq/sequential.py:167-202) and the tolerance-mode
// kernels, which issue one LDS broadcast per five VALU instructions otherwise, keep the LDS
// pipe out of their way.  Every function below that reads the table is a template over the
// pointer type; ints_of() / pairs_of() reinterpret within the pointer's own address space.
typedef const __attribute__((address_space(4))) double *ctblp;
typedef const __attribute__((address_space(4))) int32_t *ctbli;
typedef const __attribute__((address_space(4))) d2 *ctbl2;
__device__ __forceinline__ tbli ints_of(tblp p) { return reinterpret_cast<tbli>(p); }
__device__ __forceinline__ ctbli ints_of(ctblp p) { return (ctbli)p; }
__device__ __forceinline__ const d2 *pairs_of(tblp p) { return reinterpret_cast<const d2 *>(p); }
__device__ __forceinline__ ctbl2 pairs_of(ctblp p) { return (ctbl2)p; }

// ---------------------------------------------------------------- kernel args
struct TraceArgs {
    const double *rows;        // [N][kRowDoubles] dev_surface rows
    const double *n_table;     // [W][N]
    const double *ph_consts;   // [W][N][kPhaseConsts] or nullptr
    const double *wvls;        // [W] nm
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
    // HITS_COMPACT: decoupled look-back state of this launch
    uint64_t *tile_state;      // [tiles] (epoch << 32 | flag << 30 | count)
    uint32_t *ticket;          // [0] next tile, [1] workgroups done
    // pairs already in out.seg when this launch starts (earlier launches of a chunked
    // call; earlier calls with ROX_HITS_APPEND) -- nullptr = none -- and where the running
    // total goes.  Never the same word: a tile may still be reading the base while the
    // last tile already knows the total (the host ping-pongs two slots between launches).
    const int64_t *hits_base_in;
    int64_t *hits_total_out;
    uint32_t epoch;
    // the first `small_tiles` tickets are tiles of kSmallTile rays (see compact_tiles())
    int32_t small_tiles;
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

// np.cross on float64[3]: each component is multiply, multiply, subtract
__device__ __forceinline__ v3 cross3(const v3 &a, const v3 &b)
{
    return v3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
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
// every active lane of the wave passes `p`.  __all() goes through an integer (v_cndmask 0/1,
// v_cmp_ne, s_cmp vcc == exec); the ballot of the failing lanes is the compare mask itself
// (inactive lanes read 0): two VALU instructions fewer per test site.
#ifndef ROX_WAVE_ALL_BALLOT
#define ROX_WAVE_ALL_BALLOT 1
#endif
__device__ __forceinline__ bool wave_all(bool p)
{
#if ROX_WAVE_ALL_BALLOT
    return __builtin_amdgcn_ballot_w64(!p) == 0;
#else
    return __all(p);
#endif
}

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
    if (wave_all(in_band(x)))
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

// a / b
__device__ __forceinline__ double slim_div(double a, double b)
{
#if ROX_SLIM_FP64
    if (wave_all(in_band(b) && in_band_or_zero(a)))
        return div_band(a, b, rcp_band(b));
#endif
    return a / b;
}

// (a.x / b, a.y / b, a.z / b)
__device__ __forceinline__ v3 slim_div3(const v3 &a, double b)
{
#if ROX_SLIM_FP64
    if (wave_all(in_band(b) && in_band_or_zero(a.x) && in_band_or_zero(a.y) && in_band_or_zero(a.z))) {
        const double r = rcp_band(b);
        return v3{div_band(a.x, b, r), div_band(a.y, b, r), div_band(a.z, b, r)};
    }
#endif
    return v3{a.x / b, a.y / b, a.z / b};
}

// Largest s with sqrt(s) <= t.  The aperture test of the reference is
// `sqrt(x*x + y*y) <= max_aperture + fuzz` (interface.py:113-122); the correctly
// rounded sqrt is monotonic, so that is the same predicate as `x*x + y*y <= u(t)`
// with u = this threshold -- computed once per surface per workgroup, it takes the
// square root out of the per-ray path without changing a single outcome (NaN and
// infinities included).
__device__ __forceinline__ double sqrt_le_threshold(double t)
{
    if (!(t >= 0.0))
        return (t != t) ? t : -1.0;             // NaN: never true; negative: never true
    if (t == __builtin_inf())
        return t;
    double s = t * t;
    if (s == __builtin_inf())
        s = 1.7976931348623157e308;
    while (sqrt(s) > t)                         // at most a few steps either way
        s = __longlong_as_double(__double_as_longlong(s) - 1);
    for (;;) {
        if (s >= 1.7976931348623157e308)
            break;
        const double n = __longlong_as_double(__double_as_longlong(s) + 1);
        if (sqrt(n) > t)
            break;
        s = n;
    }
    return s;
}

// numerator test of unit(): zero, or not below the band (the upper edge is implied there)
__device__ __forceinline__ bool zero_or_not_tiny(double x)
{
    const uint32_t h = (uint32_t)__double2hiint(x) & 0x7fffffffu;
    return h >= 0x28000000u || x == 0.0;
}

// misc_math.py:48-54 normalize
__device__ __forceinline__ v3 unit(const v3 &v)
{
    const double l2 = dot3(v, v);
#if ROX_SLIM_FP64 && ROX_UNIT_FUSED
    // One decision for the sqrt and the three quotients.  l2 in the band puts
    // len = sqrt(l2) in [2^-192, 2^193) -- inside the band and non-zero, no test needed --
    // and every |v_i| <= len (1 + 2^-51) < 2^385, so only the lower edge of the numerators
    // is tested.  The instruction sequences are those of slim_sqrt / slim_div3.
    if (wave_all(in_band(l2) && zero_or_not_tiny(v.x) && zero_or_not_tiny(v.y) && zero_or_not_tiny(v.z))) {
        const double len = sqrt_band(l2);
        const double r = rcp_band(len);
        return v3{div_band(v.x, len, r), div_band(v.y, len, r), div_band(v.z, len, r)};
    }
    const double len = sqrt(l2);
    if (len == 0.0)
        return v;
    return v3{v.x / len, v.y / len, v.z / len};
#else
    const double len = slim_sqrt(l2);
    if (len == 0.0)
        return v;
    return slim_div3(v, len);
#endif
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
    const v3 num{n_in * d.x + alpha * nrm.x, n_in * d.y + alpha * nrm.y, n_in * d.z + alpha * nrm.z};
    out = slim_div3(num, n_out);
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
// kind = ROX_EVENPOLY | ROX_RADIALPOLY | ROX_YTOROID | ROX_XTOROID (wave-uniform);
// The sag and slope series of the polynomial profiles in one pass: two independent chains,
//   z_asp += coef_i * z_pow;  z_pow *= m;     e_asp += (c_i * coef_i) * e_pow;  e_pow *= m
// -- the same operations in the same order as the reference's two loops (forward power
// accumulation, every product and sum separately rounded).  cd = the row's interleaved
// (coef_i, c_i * coef_i) pairs, one 16-byte LDS word per term.
//
// A full-length series in the RadialPolynomial instance (all eight aspheres of the phone lens
// carry ten coefficients) is straight-line code: the ten LDS words are requested before the
// first product needs one, where the generic loop -- unrolled by eight plus a one-term
// remainder loop by the compiler -- waits for most words one at a time.  Phone lens HITS 249
// -> 232 us, FULL 284.5 -> 264 us per 2^20 rays, bit-identical.  Measured and not kept
// (gpurun_out/r04e, r04f): the same for every instance (the EvenPolynomial instance pays for
// the extra code: Nikkor HITS 319 -> 330 us), straight-line for every length with a scalar
// branch per term (Nikkor 330, .zmx zoom 161 -> 171, phone lens 242), and prefetching the next
// term's word in the loop (phone lens 253).
//
// SHARED (the EvenPolynomial and toroid call sites: z_pow = m, e_pow = 1): the slope loop's power
// 1 * m * m ... and the sag loop's m * m ... are the same numbers one term apart (1 * m is m
// exactly, and from there both chains multiply the same value by m), so one chain serves both --
// five operations per term instead of six, every result the one the two loops produce.
template <int FEAT, bool SHARED = false, class CD>
__device__ __forceinline__ void series2(CD cd, int ncoef, double m, double z_pow, double e_pow,
                                        double &z_asp, double &e_asp)
{
    z_asp = 0.0;
    e_asp = 0.0;
    if (SHARED && ROX_POLY_SHARED_POW) {
        double pw = 1.0;                    // e_pow of term i; pw * m = z_pow of term i
        for (int i = 0; i < ncoef; ++i) {
            const d2 c = cd[i];
            e_asp += c.y * pw;
            pw *= m;
            z_asp += c.x * pw;
        }
        return;
    }
    if ((FEAT & F_RADIAL) && !(FEAT & F_EVEN) && ncoef == ROX_MAX_COEF) {
        d2 c[ROX_MAX_COEF];
#pragma unroll
        for (int i = 0; i < ROX_MAX_COEF; ++i)
            c[i] = cd[i];
#pragma unroll
        for (int i = 0; i < ROX_MAX_COEF; ++i) {
            z_asp += c[i].x * z_pow;
            z_pow *= m;
            e_asp += c[i].y * e_pow;
            e_pow *= m;
        }
        return;
    }
    for (int i = 0; i < ncoef; ++i) {
        const d2 c = cd[i];
        z_asp += c.x * z_pow;
        z_pow *= m;
        e_asp += c.y * e_pow;           // (c_coef*coefs[i])*r_pow
        e_pow *= m;
    }
}

// FEAT says which of the three families this kernel instance carries code for.
template <int FEAT, bool WANT_F, class TP>
__device__ __forceinline__ bool poly_eval(int kind, double cv, double cc1, double ec, double cR,
                                          int ncoef, TP coefs,
                                          const v3 &p, double &f, v3 &df)
{
    // (coefs[i], dcoefs[i]) pairs of the device row
    const auto cd = pairs_of(coefs + (offsetof(dev_surface, cd) - offsetof(rox_surface, coefs)) / 8);
    if ((FEAT & F_TOROID) && (!(FEAT & (F_EVEN | F_RADIAL)) || kind >= ROX_YTOROID)) {
        // profiles.py:1337-1377 YToroid.fY/f/df; XToroid swaps x and y (:1429-1434)
        const bool xt = (kind == ROX_XTOROID);
        const double px = xt ? p.y : p.x, py = xt ? p.x : p.y;
        const double y2 = py * py;
        const double rad = 1. - cc1 * cv * cv * y2;
        if (rad < 0.0)
            return false;
        const double srad = slim_sqrt(rad);
        double z_asp, e_asp;
        series2<FEAT, true>(cd, ncoef, y2, y2, 1, z_asp, e_asp);
        const double fY = slim_div(cv * y2, 1. + srad) + z_asp;
        if (WANT_F)
            f = p.z - fY - cR * (px * px + p.z * p.z - fY * fY) / 2;
        const double dfdY = slim_div(cv, srad) + e_asp;
        const double Fx = -cR * px;
        const double Fy = (cR * fY - 1) * (dfdY) * py;
        df = xt ? v3{Fy, Fx, 1 - cR * p.z} : v3{Fx, Fy, 1 - cR * p.z};
        return true;
    }
    const bool radial = (FEAT & F_RADIAL) && (!(FEAT & F_EVEN) || kind == ROX_RADIALPOLY);
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
            const double srad = slim_sqrt(rad);
#if ROX_POLY_SAME_BRANCH
            srad_e = srad;
            if (!same) {
                // (the empty asm keeps this a branch: as a select the compiler evaluated the
                // second square root -- the full expansion -- in every evaluation)
                asm volatile("" ::: "memory");
                srad_e = sqrt(rad_e);
            }
#else
            srad_e = same ? srad : sqrt(rad_e);
#endif
            const double z = slim_div(cv * r2, 1. + srad);
            const double e = slim_div(cv, srad_e);
            double z_asp, e_asp;
            series2<FEAT, true>(cd, ncoef, r2, r2, 1, z_asp, e_asp);
            f = p.z - (z + z_asp);
            e_tot = e + e_asp;
        } else {
            srad_e = sqrt(rad_e);
            const double e = slim_div(cv, srad_e);
            double r_pow = 1, e_asp = 0.0;
            for (int i = 0; i < ncoef; ++i) {
                e_asp += cd[i].y * r_pow;       // (c_coef*coefs[i])*r_pow
                r_pow *= r2;
            }
            e_tot = e + e_asp;
        }
    } else {
        const double r = slim_sqrt(r2);
        if (WANT_F && rad_e < 0.0)
            return false;
        const double srad_e = WANT_F ? slim_sqrt(rad_e) : sqrt(rad_e);   // NaN when negative
        const double e = slim_div(cv, srad_e);
        double e_asp = 0.0;
        double e_pow = (r == 0.0) ? 1.0 : slim_div(1.0, r);
        if (WANT_F) {
            const double z = slim_div(cv * r2, 1. + srad_e);
            double z_asp;
            series2<FEAT>(cd, ncoef, r, r, e_pow, z_asp, e_asp);
            f = p.z - (z + z_asp);
        } else {
            for (int i = 0; i < ncoef; ++i) {
                e_asp += cd[i].y * e_pow;
                e_pow *= r;
            }
        }
        e_tot = e + e_asp;
    }
    df = v3{-e_tot * p.x, -e_tot * p.y, 1.0};
    return true;
}

// profiles.py:155-186 Spencer & Murty Newton iteration.  Returns the last
// *evaluated* iterate as the hit point (p0 itself when |s1| <= eps at once).
template <int FEAT, class TP>
__device__ __forceinline__ bool newton_hit(int kind, double cv, double cc1, double ec, double cR,
                                           int ncoef, TP coefs,
                                           const v3 &p0, const v3 &d, double eps,
                                           double &s, v3 &hit, v3 &df)
{
    v3 p = p0;
    double f;
    if (!poly_eval<FEAT, true>(kind, cv, cc1, ec, cR, ncoef, coefs, p, f, df))
        return false;
#if ROX_NEWTON_SLIM_DIV
    double s1 = slim_div(-f, dot3(d, df));
#else
    double s1 = -f / dot3(d, df);
#endif
    double delta = fabs(s1);
    int iter = 0;
    bool ok = true;
    // one Spencer-Murty step for the lanes that have not converged
    auto step = [&]() {
        p = v3{p0.x + s1 * d.x, p0.y + s1 * d.y, p0.z + s1 * d.z};
        if (!poly_eval<FEAT, true>(kind, cv, cc1, ec, cR, ncoef, coefs, p, f, df)) {
            ok = false;
            delta = 0.0;            // leave the iteration; the caller reports the miss
            return;
        }
#if ROX_NEWTON_SLIM_DIV
        const double s2 = s1 - slim_div(f, dot3(d, df));
#else
        const double s2 = s1 - f / dot3(d, df);
#endif
        delta = fabs(s2 - s1);
        s1 = s2;
        ++iter;
    };
    // measured on the reference's even-asphere zoom: 2 steps 20 %, 3 steps 73 %,
    // 4 steps 6 %, more < 1 %.  ROX_NEWTON_UNROLL steps straight-line and
    // predicated per lane (0 since round 5), then the loop (cap 1000 as in the reference)
#pragma unroll
    for (int u = 0; u < ROX_NEWTON_UNROLL; ++u)
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

// ================================================================ tolerance mode
// ROX_FAST_FP64 (rox_opts.flags, include/roxtrace.h): the caller accepts results within the
// tolerance BASELINE's north star states -- 1e-10 on ray intercepts, scaled by max(1, |ref|) --
// instead of the reference's bits.  The reduced-output modes (HITS, HITS_COMPACT, LAST, OPD, FAN)
// are bound by VALU issue, and 2.4 x their algorithmic flop count is the price of bit parity:
// every `/` and `sqrt` a 10-25-instruction correctly rounded expansion behind an exponent-band
// test, every product and sum separately rounded in NumPy's order, |n| re-measured where it is
// 1 by construction.  The F_FAST instances below are still IEEE binary64 throughout, with
//   * v_rcp_f64 / v_rsq_f64 (2^-23 .. 2^-24 relative) + two Newton-Raphson steps (<= ~1.5 ulp)
//     instead of the correctly rounded expansions, no band tests, no fall-back paths;
//   * fused multiply-adds wherever the formula has a*b + c;
//   * refraction through mu = n_in / n_out staged per (wavelength, interface): no division
//     per ray; the surface normal is taken as the unit vector it is (raytrace.py:19-30 divides
//     by its length again), and a sphere's gradient (-cv x, -cv y, 1 - cv z) is not normalised
//     at all: its squared length is 1 + cv F(hit) with F the surface function, i.e. 1 up to
//     the residual of the intersection;
//   * Horner evaluation of the asphere series (profiles.py:849-885, 1070-1113 accumulate
//     powers forward), 1/sqrt from the same iteration that gives the sqrt;
//   * the Spencer-Murty iteration itself as the reference runs it (same start, same
//     convergence test, the penultimate iterate returned): the iterates differ from the
//     reference's by rounding only, so both stop within eps of the same point.
// Ray-failure decisions (miss, TIR, aperture) are taken on values that differ from the
// reference's in the last bits: a ray whose radicand / aperture margin is within rounding of
// zero may be decided the other way (tests/test_gpu_fast.py counts such rays and shows each
// of them on its boundary).  Degenerate operands keep the reference's outcomes where those are
// defined (den == 0 in Spherical.intersect, r == 0 in RadialPolynomial.df, a zero-length
// gradient); NaN and infinity propagate as NaN.
constexpr int F_FAST = 64;
#ifndef ROX_FAST_NR          // Newton-Raphson steps after v_rcp_f64 / v_rsq_f64: 2 = ~1 ulp,
#define ROX_FAST_NR 2         // 1 = ~2^-45 relative (measured difference: see EXPERIMENTS.md)
#endif

// 1 / b, b != 0 and finite
__device__ __forceinline__ double rcp_f(double b)
{
    double r = __builtin_amdgcn_rcp(b);
    r = fma(r, fma(-b, r, 1.0), r);
#if ROX_FAST_NR >= 2
    r = fma(r, fma(-b, r, 1.0), r);
#endif
    return r;
}

// s = sqrt(x), h = 0.5 / sqrt(x) by the coupled iteration on (x y, y / 2), y = v_rsq_f64(x).
// The seed is taken of x + 2^-1000 (= x for every x >= 2^-947): x = 0 gives s = 0 (a finite y
// times 0) instead of the NaN of 0 * inf -- an on-axis ray has r^2 = 0 exactly -- and NaN stays
// NaN.  (One v_add; fmax() costs two v_max in IEEE mode, the first to quiet a signalling NaN.)
__device__ __forceinline__ void sqrt_half_rsqrt_f(double x, double &s, double &h)
{
    const double y = __builtin_amdgcn_rsq(x + 0x1p-1000);
    s = x * y;
    h = 0.5 * y;
    const double e = fma(-h, s, 0.5);
    s = fma(s, e, s);
    h = fma(h, e, h);
#if ROX_FAST_NR >= 2
    const double e1 = fma(-h, s, 0.5);
    s = fma(s, e1, s);
    h = fma(h, e1, h);
#endif
}

__device__ __forceinline__ double sqrt_f(double x)
{
    const double y = __builtin_amdgcn_rsq(x + 0x1p-1000);
    double s = x * y;
    const double h = 0.5 * y;
    s = fma(s, fma(-h, s, 0.5), s);
#if ROX_FAST_NR >= 2
    s = fma(fma(-s, s, x), h, s);       // (the residual form: h need not be refined for it)
#endif
    return s;
}

__device__ __forceinline__ double dot3_f(const v3 &a, const v3 &b)
{
    return fma(a.z, b.z, fma(a.y, b.y, a.x * b.x));
}

// v / |v|; the zero vector stays the zero vector (misc_math.py:48-54)
__device__ __forceinline__ v3 unit_f(const v3 &v)
{
    double s, h;
    sqrt_half_rsqrt_f(dot3_f(v, v), s, h);
    const double y = h + h;
    return v3{v.x * y, v.y * y, v.z * y};
}

template <class P>
__device__ __forceinline__ v3 rotate_f(P rt, const v3 &v)
{
    return v3{fma(rt[2], v.z, fma(rt[1], v.y, rt[0] * v.x)),
              fma(rt[5], v.z, fma(rt[4], v.y, rt[3] * v.x)),
              fma(rt[8], v.z, fma(rt[7], v.y, rt[6] * v.x))};
}

// raytrace.py:19-38 with a unit normal, ONE formula for every interact mode, straight-line:
//   d_out = mu d + (sg sign(cosI) sqrt(1 - mu^2 (1 - cosI^2)) - mu cosI) n
//   transmit  mu = n_in / n_out, sg = +1          (bend)
//   reflect   mu = 1, sg = -1: the root is |cosI|, d_out = d - 2 cosI n   (reflect)
//   (mu = 1, sg = +1 would pass the direction through -- the bracket is |cosI| sign(cosI) - cosI
//   = 0 -- but dummy and phantom interfaces copy it instead, as the reference does: a NaN normal
//   of a degenerate ray must not reach the direction)
// (mu, mu^2, sg) are staged per (wavelength, interface) by the workgroup.  Returns false where
// the radicand is negative (total internal reflection; out is NaN there).
__device__ __forceinline__ bool interact_f(const v3 &d, const v3 &n, double mu, double mu2, double sg,
                                           v3 &out)
{
    const double c = dot3_f(d, n);
    const double rad = fma(mu2, fma(c, c, -1.0), 1.0);
    const double alpha = fma(sg, copysign(sqrt_f(rad), c), -(mu * c));
    out = v3{fma(alpha, n.x, mu * d.x), fma(alpha, n.y, mu * d.y), fma(alpha, n.z, mu * d.z)};
    return !(rad < 0.0);
}

// profiles.py:321-336 / 580-593.  The reference's outcomes for a vanishing denominator
// (FloatingPointError -> s = 0 for a finite non-zero numerator, NaN for 0/0 unless all three
// coefficients vanish) are kept by one wave-uniform test.
// Straight-line: a negative radicand does not leave -- the lane's s and hit become NaN (the
// seed of sqrt_f is NaN for it) and the caller reads the returned flag.  A branch here costs a
// saved exec mask and, where the paths meet again, a register copy per live value.
__device__ __forceinline__ bool quadric_hit_f(bool conic, double cv, double cc, double ec,
                                              const v3 &p, const v3 &d, double z_dir,
                                              double &s, v3 &hit)
{
    double ax2, cx2, b;
    if (!conic) {
        ax2 = cv;
        cx2 = fma(cv, dot3_f(p, p), -(p.z + p.z));
        b = fma(cv, dot3_f(d, p), -d.z);
    } else {
        ax2 = cv * fma(cc * d.z, d.z, 1.0);
        const double ez = ec * p.z;
        cx2 = fma(cv, fma(ez, p.z, fma(p.y, p.y, p.x * p.x)), -(p.z + p.z));
        b = fma(cv, fma(ez, d.z, fma(d.y, p.y, d.x * p.x)), -d.z);
    }
    const double rad = fma(-ax2, cx2, b * b);
    const double den = fma(z_dir, sqrt_f(rad), -b);
    s = cx2 * rcp_f(den);
    if (__builtin_amdgcn_ballot_w64(den == 0.0) != 0) {
        if (den == 0.0) {
            if (cx2 != 0.0 && isfinite(cx2))
                s = 0.0;
            else if (b == 0.0 && cx2 == 0.0 && ax2 == 0.0)
                s = 0.0;
            else
                s = __builtin_nan("");
        }
    }
    hit = v3{fma(s, d.x, p.x), fma(s, d.y, p.y), fma(s, d.z, p.z)};
    return !(rad < 0.0);                        // false: TraceMissedSurfaceError
}

// f(p), df(p) of EvenPolynomial / RadialPolynomial (profiles.py:849-885, 1070-1113), Horner
// form.  Toroids keep the exact evaluation (poly_eval): no workload of BASELINE carries one.
template <int FEAT, class TP>
__device__ __forceinline__ bool poly_eval_f(int kind, double cv, double cc1, double ec, double cR,
                                            int ncoef, TP coefs, const v3 &p, double &f, v3 &df)
{
    if ((FEAT & F_TOROID) && kind >= ROX_YTOROID)
        return poly_eval<FEAT & ~F_FAST, true>(kind, cv, cc1, ec, cR, ncoef, coefs, p, f, df);
    const auto cd = pairs_of(coefs + (offsetof(dev_surface, cd) - offsetof(rox_surface, coefs)) / 8);
    const bool radial = (FEAT & F_RADIAL) && (!(FEAT & F_EVEN) || kind == ROX_RADIALPOLY);
    const double r2 = fma(p.y, p.y, p.x * p.x);
    const double cv2 = cv * cv;
    double z_asp = 0.0, e_asp = 0.0, e_tot;
    if (!radial) {
        // sag: z = cv r2 / (1 + sqrt(1 - (cc+1) cv^2 r2)) + sum coef_i r2^(i+1)
        // df:  e = cv / sqrt(1 - ec cv^2 r2)               + sum 2(i+1) coef_i r2^i
        const double rad = fma(-cc1 * cv2, r2, 1.0);
        if (rad < 0.0)
            return false;
        double srad, hrad;
        sqrt_half_rsqrt_f(rad, srad, hrad);
        double he = hrad;                       // 0.5 / sqrt(rad_e)
        if (cc1 != ec) {                        // (wave-uniform; equal unless a caller fills ec otherwise)
            double se;
            const double rad_e = fma(-ec * cv2, r2, 1.0);
            sqrt_half_rsqrt_f(rad_e, se, he);
            if (rad_e < 0.0)
                he = __builtin_nan("");         // np.sqrt of a negative: NaN, no exception
        }
        for (int i = ncoef - 1; i >= 0; --i) {
            const d2 c = cd[i];
            z_asp = fma(z_asp, r2, c.x);
            e_asp = fma(e_asp, r2, c.y);
        }
        const double z = (cv * r2) * rcp_f(1.0 + srad);
        f = p.z - fma(z_asp, r2, z);
        e_tot = fma(cv + cv, he, e_asp);
    } else {
        // sag: z = cv r2 / (1 + sqrt(1 - ec cv^2 r2)) + sum coef_i r^(i+1)
        // df:  e = cv / sqrt(.)                         + sum (i+1) coef_i r^(i-1)
        const double rad = fma(-ec * cv2, r2, 1.0);
        if (rad < 0.0)
            return false;
        double srad, hrad, r, hr;
        sqrt_half_rsqrt_f(rad, srad, hrad);
        sqrt_half_rsqrt_f(r2, r, hr);
        const double rinv = (r2 == 0.0) ? 1.0 : hr + hr;    // profiles.py:1104: r_pow = 1 at r = 0
        for (int i = ncoef - 1; i >= 1; --i) {
            const d2 c = cd[i];
            z_asp = fma(z_asp, r, c.x);
            e_asp = fma(e_asp, r, c.y);
        }
        const d2 c0 = cd[0];
        z_asp = fma(z_asp, r, ncoef > 0 ? c0.x : 0.0);
        e_asp = fma(ncoef > 0 ? c0.y : 0.0, rinv, e_asp);
        const double z = (cv * r2) * rcp_f(1.0 + srad);
        f = p.z - fma(z_asp, r, z);
        e_tot = fma(cv + cv, hrad, e_asp);
    }
    df = v3{-e_tot * p.x, -e_tot * p.y, 1.0};
    return true;
}

// profiles.py:155-186, as newton_hit() runs it
template <int FEAT, class TP>
__device__ __forceinline__ bool newton_hit_f(int kind, double cv, double cc1, double ec, double cR,
                                             int ncoef, TP coefs, const v3 &p0, const v3 &d,
                                             double eps, double &s, v3 &hit, v3 &df)
{
    v3 p = p0;
    double f;
    if (!poly_eval_f<FEAT>(kind, cv, cc1, ec, cR, ncoef, coefs, p, f, df))
        return false;
    double s1 = -f * rcp_f(dot3_f(d, df));
    double delta = fabs(s1);
    int iter = 0;
    bool ok = true;
    while (delta > eps && iter < 1000) {
        p = v3{fma(s1, d.x, p0.x), fma(s1, d.y, p0.y), fma(s1, d.z, p0.z)};
        if (!poly_eval_f<FEAT>(kind, cv, cc1, ec, cR, ncoef, coefs, p, f, df)) {
            ok = false;
            break;
        }
        const double s2 = fma(-f, rcp_f(dot3_f(d, df)), s1);
        delta = fabs(s2 - s1);
        s1 = s2;
        ++iter;
    }
    if (!ok)
        return false;
    s = s1;
    hit = p;
    return true;
}

// surface.py:198-208 (+ surface.py:416-419, 453-457): the AND over an interface's
// clear_apertures.  A circular aperture's radius slot of the STAGED row holds
// sqrt_le_threshold(radius + fuzz) (stage_aperture_thresholds() below), so that
// `sqrt(xx*xx + yy*yy) <= radius + fuzz` is decided without the square root, outcome for
// outcome; interfaces without a list take the max_aperture threshold (Ctx.apthr).
// thr != nullptr (the F_GTAB instances, whose table is read-only): the interface's circular
// thresholds, staged in LDS beside the table instead of inside it.
template <class TP>
__device__ __forceinline__ bool inside_aperture_list(TP row, int n_ap, double x, double y, double fuzz,
                                                     tblp thr = nullptr)
{
    auto ap = row + (offsetof(rox_surface, ap) / sizeof(double));
    for (int k = 0; k < n_ap; ++k, ap += sizeof(rox_aperture) / sizeof(double)) {
        const int2 ki{ints_of(ap)[0], ints_of(ap)[1]};            // kind, is_obscuration
        const double xx = x - ap[1];
        const double yy = y - ap[2];
        bool ans;
        if (ki.x == ROX_AP_CIRCULAR)
            ans = (xx * xx + yy * yy) <= (thr ? thr[k] : ap[3]);    // the staged threshold
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

// Post-pass over the table a workgroup has staged in LDS (instances with aperture lists; call
// between two workgroup barriers): the radius of every circular clear aperture becomes the
// largest s with sqrt(s) <= radius + fuzz.  `fuzz` is the launch's pt_inside_fuzz.
template <int FEAT>
__device__ __forceinline__ void stage_aperture_thresholds(double *tbl_w, int N, double fuzz, int tid,
                                                          int nthreads)
{
    if (!(FEAT & F_APLIST))
        return;
    for (int i = tid; i < N * ROX_MAX_AP; i += nthreads) {
        double *row = tbl_w + (size_t)(i / ROX_MAX_AP) * kRowDoubles;
        const int k = i % ROX_MAX_AP;
        if (k < reinterpret_cast<const int32_t *>(row)[3]) {
            double *ap = row + offsetof(rox_surface, ap) / sizeof(double) +
                         (size_t)k * (sizeof(rox_aperture) / sizeof(double));
            if (reinterpret_cast<const int32_t *>(ap)[0] == ROX_AP_CIRCULAR)
                ap[3] = sqrt_le_threshold(ap[3] + fuzz);
        }
    }
}

// ------------------------------------------------------------------ phase
// x**n as the reference's `r_sqr**(i+1)` evaluates it: libm pow(), which is
// correctly rounded except for ~1e-3 of its arguments.  Here: the product in
// double-double (error ~2^-100), rounded once -- the correctly rounded power.
__device__ __forceinline__ double pow_int(double x, int n)
{
    if (n == 0)
        return 1.0;
    double hi = x, lo = 0.0;
    for (int k = 1; k < n; ++k) {
        const double ph = hi * x;
        const double pl = fma(hi, x, -ph);
        const double t = fma(lo, x, pl);
        hi = ph + t;
        lo = t - (hi - ph);
    }
    return hi;
}

enum { PHASE_OK = 0, PHASE_EVANESCENT = 1, PHASE_TIR = 2 };

// raytrace.py:41-48 phase() over the phase element of the row.  ph points at
// rox_surface.ph of the row; pc at the (wavelength, interface) grating
// constants.  Returns PHASE_*; `out` = after_dir, `dW` = phs.
template <class TP>
__device__ __forceinline__ int apply_phase(TP ph, TP pc, const v3 &pt, const v3 &in_dir,
                                           const v3 &srf_nrml, double z_dir, double wvl,
                                           double n_in, double n_out, int mode, v3 &out,
                                           double &dW)
{
    constexpr int O_ORDER = offsetof(rox_phase, order) / 8, O_REFWL = offsetof(rox_phase, ref_wl) / 8,
                  O_SPACING = offsetof(rox_phase, spacing_nm) / 8, O_A = offsetof(rox_phase, a) / 8,
                  O_B = offsetof(rox_phase, b) / 8, O_COEF = offsetof(rox_phase, coefs) / 8;
    const int kind = ints_of(ph)[0];
    if (kind == ROX_PH_GRATING) {               // doe.py:124-175 phase_ludwig
        const double refl = (mode == ROX_REFLECT) ? -1.0 : 1.0;
        const v3 un = unit(srf_nrml);
        const v3 normal{z_dir * un.x, z_dir * un.y, z_dir * un.z};
        const v3 G{ph[O_A], ph[O_A + 1], ph[O_A + 2]};
        const v3 P = cross3(G, normal);
        const v3 D = unit(cross3(normal, P));
        const double mu = pc[0], mu2 = pc[1], T = pc[2], T2 = pc[3];
        const double in_cosI = dot3(in_dir, normal);
        const double V = mu * in_cosI;
        const double W = mu2 - 1 + T2 - 2 * mu * T * dot3(D, in_dir);
        const double result = sqrt(V * V - W);  // np.sqrt: NaN, not an exception
        const double Q1 = result - V;
        const double Q2 = -result - V;
        // Python's max(a, b) = b if b > a else a; min(a, b) = b if b < a else a
        double Q = Q1;
        if (mode == ROX_TRANSMIT)
            Q = (Q2 > Q1) ? Q2 : Q1;
        else if (mode == ROX_REFLECT)
            Q = (Q2 < Q1) ? Q2 : Q1;
        v3 o{mu * in_dir.x - T * D.x + Q * normal.x, mu * in_dir.y - T * D.y + Q * normal.y,
             mu * in_dir.z - T * D.z + Q * normal.z};
        const double rz = 1 - o.x * o.x - o.y * o.y;
        if (rz < 0.0)
            return PHASE_EVANESCENT;            // math.sqrt ValueError
        o.z = copysign(sqrt(rz), o.z);
        const double si = 1 - in_cosI * in_cosI;
        if (si < 0.0)
            return PHASE_EVANESCENT;
        const double in_sinI = sqrt(si);
        const double out_cosI = dot3(o, normal);
        const double so = 1 - out_cosI * out_cosI;
        if (so < 0.0)
            return PHASE_EVANESCENT;
        const double out_sinI = sqrt(so);
        dW = (ph[O_SPACING] / wvl) * (n_in * in_sinI + refl * n_out * out_sinI);
        out = o;
        return PHASE_OK;
    }
    if (kind == ROX_PH_DOE_RADIAL) {            // doe.py:272-323 + radial_phase_fct :28-54
        const double order = ph[O_ORDER];
        const v3 normal = unit(srf_nrml);
        v3 inc = in_dir;
        if (n_in != 1.0) {
            if (!refract(in_dir, srf_nrml, n_in, 1.0, inc))
                return PHASE_TIR;
        }
        const double in_cosI = dot3(inc, normal);
        const double mu = wvl / ph[O_REFWL];
        const double r_sqr = pt.x * pt.x + pt.y * pt.y;
        double w = 0, dWdX = 0, dWdY = 0;
        const int nc = ints_of(ph)[1];
        for (int i = 0; i < nc; ++i) {
            const double c = ph[O_COEF + i];
            w += c * pow_int(r_sqr, i + 1);
            const double r_exp = pow_int(r_sqr, i);
            const double fc = (double)(2 * (i + 1)) * c;
            dWdX += fc * pt.x * r_exp;
            dWdY += fc * pt.y * r_exp;
        }
        const double om = order * mu;
        const double b = in_cosI + om * (normal.x * dWdX + normal.y * dWdY);
        const double c = mu * (mu * (dWdX * dWdX + dWdY * dWdY) / 2 +
                               order * (inc.x * dWdX + inc.y * dWdY));
        const double rad = b * b - 2 * c;
        if (rad < 0.0)
            return PHASE_EVANESCENT;
        const double Q = -b + z_dir * sqrt(rad);
        v3 o{inc.x + om * dWdX + Q * normal.x, inc.y + om * dWdY + Q * normal.y,
             inc.z + om * 0.0 + Q * normal.z};
        dW = w * mu;
        if (n_in != 1.0) {
            v3 o2;
            if (!refract(o, srf_nrml, 1.0, n_out, o2))
                return PHASE_TIR;
            o = o2;
        }
        out = o;
        return PHASE_OK;
    }
    // ROX_PH_HOLOGRAM, doe.py:375-397
    const int flags = ints_of(ph)[2];
    const v3 normal = unit(srf_nrml);
    v3 ref_dir = unit(v3{pt.x - ph[O_A], pt.y - ph[O_A + 1], pt.z - ph[O_A + 2]});
    if (flags & 1)
        ref_dir = v3{-ref_dir.x, -ref_dir.y, -ref_dir.z};
    const double ref_cosI = dot3(ref_dir, normal);
    v3 obj_dir = unit(v3{pt.x - ph[O_B], pt.y - ph[O_B + 1], pt.z - ph[O_B + 2]});
    if (flags & 2)
        obj_dir = v3{-obj_dir.x, -obj_dir.y, -obj_dir.z};
    const double obj_cosI = dot3(obj_dir, normal);
    const double in_cosI = dot3(in_dir, normal);
    const double mu = wvl / ph[O_REFWL];
    const double b = in_cosI + mu * (obj_cosI - ref_cosI);
    const double refp_cosI = dot3(ref_dir, in_dir);
    const double objp_cosI = dot3(obj_dir, in_dir);
    const double ro_cosI = dot3(ref_dir, obj_dir);
    const double c = mu * (mu * (1.0 - ro_cosI) + (objp_cosI - refp_cosI));
    const double rad = b * b - 2 * c;
    if (rad < 0.0)
        return PHASE_EVANESCENT;
    const double Q = -b + z_dir * sqrt(rad);
    out = v3{in_dir.x + mu * (obj_dir.x - ref_dir.x) + Q * normal.x,
             in_dir.y + mu * (obj_dir.y - ref_dir.y) + Q * normal.y,
             in_dir.z + mu * (obj_dir.z - ref_dir.z) + Q * normal.z};
    dW = 0.;
    return PHASE_OK;
}

// ------------------------------------------------------------------ OPD
// FAST (the tolerance-mode instances): the quotients and the square root of the wave-aberration
// formulas through rcp_f / sqrt_f (~1 ulp) instead of the correctly rounded expansions; the
// operation order is otherwise the reference's.  An OPD is a difference of optical paths of
// O(100) system units: 1e-16 relative on its terms is 1e-14 absolute.
template <bool FAST>
__device__ __forceinline__ double wf_div(double a, double b) { return FAST ? a * rcp_f(b) : a / b; }
template <bool FAST>
__device__ __forceinline__ double wf_sqrt(double x) { return FAST ? sqrt_f(x) : sqrt(x); }

// waveabr.py:117-132 eic_distance
template <bool FAST = false, class P3>
__device__ __forceinline__ double eic_distance(const v3 &p, const v3 &d, P3 p0, P3 d0)
{
    const v3 sd{d.x + d0[0], d.y + d0[1], d.z + d0[2]};
    const v3 dp{p.x - p0[0], p.y - p0[1], p.z - p0[2]};
    return wf_div<FAST>(dot3(sd, dp), 1. + dot3(d, v3{d0[0], d0[1], d0[2]}));
}

// waveabr.py:256-307 wave_abr_full_calc_finite_pup (+ transform.py:234-258)
template <bool FAST = false, class WF>      // rox_wavefront, in whatever address space the launch arguments live
__device__ __forceinline__ double wave_abr_finite_pup(WF &w, const v3 &ray1_p,
                                                      const v3 &ray0_d, const v3 &rayk_p,
                                                      const v3 &rayk_d, double ray_op)
{
    const double e1 = eic_distance<FAST>(ray1_p, ray0_d, w.cr1_p, w.cr0_d);
    const double ekp = eic_distance<FAST>(rayk_p, rayk_d, w.crk_p, w.crk_d);
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
    // (R is a launch constant: the tolerance mode multiplies by its reciprocal, formed once)
    const double rR = FAST ? rcp_f(R) : 0.0;
    const double F = dot3(rd, b4d) - (FAST ? dot3(b4d, pc) * rR : dot3(b4d, pc) / R);
    const double J = (FAST ? dot3(pc, pc) * rR : dot3(pc, pc) / R) - 2.0 * dot3(rd, pc);
    const double denom = F + w.sign_soln * wf_sqrt<FAST>(F * F + (FAST ? J * rR : J / R));
    const double ep = (denom == 0) ? 0 : wf_div<FAST>(J, denom);
    return -w.n_obj * e1 - ray_op + w.n_img * ekp + w.cr_op - w.n_img * ep;
}

// waveabr.py:356-424 wave_abr_full_calc_inf_ref (ROX_WF_INF_FULL) and its pre-calc /
// calc split :427-488 (ROX_WF_INF_SPLIT); dist_to_shortest_join :178-196,
// ray_dist_to_perp_from_origin :166-175.  Chief-ray-only terms come from the host.
template <bool FAST = false, class WF>
__device__ __forceinline__ double wave_abr_inf_ref(WF &w, const v3 &ray1_p,
                                                   const v3 &ray0_d, const v3 &rayk_p,
                                                   const v3 &rayk_d, const v3 &rayl_p,
                                                   const v3 &rayl_d, double ray_op)
{
    const double e1 = eic_distance<FAST>(ray1_p, ray0_d, w.cr1_p, w.cr0_d);
    v3 p_b4 = rayk_p, d_b4 = rayk_d;
    if (w.last_kind) {
        p_b4 = rotate(w.last_rt, w.last_order, v3{rayk_p.x - w.last_t[0], rayk_p.y - w.last_t[1],
                                                  rayk_p.z - w.last_t[2]});
        d_b4 = rotate(w.last_rt, w.last_order, rayk_d);
    }
    const double op_b4 = dot3(d_b4, v3{-p_b4.x, -p_b4.y, -p_b4.z});
    const v3 p1{w.cr_last_p[0], w.cr_last_p[1], w.cr_last_p[2]};
    const v3 d1{w.cr_last_d[0], w.cr_last_d[1], w.cr_last_d[2]};
    const v3 del_p{rayl_p.x - p1.x, rayl_p.y - p1.y, rayl_p.z - p1.z};
    const v3 n = cross3(d1, rayl_d);
    const double nn = dot3(n, n);
    v3 P1, P2;
    if (nn == 0) {
        const double t2 = dot3(v3{p1.x - rayl_p.x, p1.y - rayl_p.y, p1.z - rayl_p.z}, d1) *
                          dot3(d1, rayl_d);
        P1 = p1;
        P2 = v3{rayl_p.x + t2 * rayl_d.x, rayl_p.y + t2 * rayl_d.y, rayl_p.z + t2 * rayl_d.z};
    } else {
        const double t1 = wf_div<FAST>(dot3(cross3(rayl_d, n), del_p), nn);
        const double t2 = wf_div<FAST>(dot3(cross3(d1, n), del_p), nn);
        P1 = v3{p1.x + t1 * d1.x, p1.y + t1 * d1.y, p1.z + t1 * d1.z};
        P2 = v3{rayl_p.x + t2 * rayl_d.x, rayl_p.y + t2 * rayl_d.y, rayl_p.z + t2 * rayl_d.z};
    }
    const v3 rF0{(P1.x + P2.x) / 2, (P1.y + P2.y) / 2, (P1.z + P2.z) / 2};
    const v3 dcr{w.d_cr_b4[0], w.d_cr_b4[1], w.d_cr_b4[2]};
    const double V_B = ray_op + op_b4;
    const double W0 = V_B - w.v_be +
                      w.n_img * dot3(v3{d_b4.x - dcr.x, d_b4.y - dcr.y, d_b4.z - dcr.z}, rF0);
    const v3 ta{rayl_p.x - w.image_pt[0], rayl_p.y - w.image_pt[1], rayl_p.z - w.image_pt[2]};
    const double dbc = dot3(d_b4, dcr);
    const double numer = dot3(v3{dcr.x - d_b4.x * dbc, dcr.y - d_b4.y * dbc, dcr.z - d_b4.z * dbc}, ta);
    const double denom = 1 + dot3(d_b4, dcr);
    if (w.kind == ROX_WF_INF_SPLIT) {
        const double pre_opd = -w.n_obj * e1 - W0;
        const double W_inf = wf_div<FAST>(w.n_img * numer, denom);
        return pre_opd - W_inf;
    }
    const double W_inf = W0 + wf_div<FAST>(w.n_img * numer, denom);
    return -w.n_obj * e1 - W_inf;
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

// the workgroup's view of the surface table (LDS) + the launch options
template <class TP>
struct CtxT {
    TP tbl;                 // [N][kRowDoubles]: LDS (tblp), or global through scalar loads (ctblp, F_GTAB)
    tblp ntab;              // [N] (one wavelength) or [W][N] (per-ray wavelengths)
    TP phc;                 // [N][kPhaseConsts] or [W][N][kPhaseConsts]; FEAT & F_PHASE only
    tblp aplthr;            // F_GTAB: [N][ROX_MAX_AP] circular clear-aperture thresholds; else nullptr
    tblp wvls;              // [W]
    tbli slot, nslots_before;
    tblp apthr;             // [N] sqrt_le_threshold(max_aperture + fuzz)
    tblp mu;                // F_FAST only: per wavelength row [N] mu, [N] mu^2, [N] sg (interact_f)
    int N;
    bool check_ap, intersect_obj, filter_ph;
    int first_surf, last_surf;
    double eps, fuzz;
    int probe_surf;         // MODE_PROBE
};
typedef CtxT<tblp> Ctx;

// what a traced ray leaves in registers for the epilogues
struct RayEnd {
    int status, fail_surf;
    double opl, phs;        // op_delta = phs + opl on success (raytrace.py:208, 261)
    v3 inc, ad, nrm;        // ray[-1]
    v3 ray1_p, rayk_p, rayk_d;   // OPD mode: ray[1].p, ray[-2].{p, d}
    v3 probe_p;             // MODE_PROBE: ray[probe_surf].p
};

// ------------------------------------------------------------------ one ray
// raytrace.py:83-264 trace_raw for one lane.  wi = wavelength index of the ray
// (wave-uniform unless PER_RAY_WVL).
// `live` = false for the lanes past the end of the batch: they trace nothing.
// kSync (FULL packets only): the waves of a workgroup meet at a barrier before
// every surface, so that the workgroup writes whole [segment] rows together;
// every wave must then reach every barrier: failed lanes idle instead of leaving.
#define ROX_LEAVE if (kSync) continue; else break
template <int OUT_MODE, bool PER_RAY_WVL, int FEAT, class CTX>
__device__ __forceinline__ void trace_ray(const CTX &c, const SegOut &so, const v3 &pt0,
                                          const v3 &dir0, int wi, bool live, RayEnd &e)
{
    constexpr int O_CV = offsetof(rox_surface, cv) / 8, O_CC = offsetof(rox_surface, cc) / 8,
                  O_EC = offsetof(rox_surface, ec) / 8, O_CR = offsetof(rox_surface, cR) / 8,
                  O_COEF = offsetof(rox_surface, coefs) / 8,
                  O_RT = offsetof(rox_surface, rt) / 8, O_T = offsetof(rox_surface, t) / 8,
                  O_ZDIR = offsetof(rox_surface, z_dir) / 8,
                  O_PH = offsetof(rox_surface, ph) / 8;
    constexpr bool kPoly = (FEAT & F_POLY) != 0;
    constexpr bool kSync = OUT_MODE == ROX_OUT_FULL && (kPoly ? ROX_WG_SYNC_POLY : ROX_WG_SYNC);
    constexpr bool kIdentRt = ROX_IDENT_RT && (OUT_MODE != ROX_OUT_FULL || (kPoly && ROX_IDENT_RT_FULL_POLY));
    const int N = c.N;
    const auto tbl = c.tbl;
    tblp nwl = PER_RAY_WVL ? c.ntab + (size_t)wi * N : c.ntab;
#define SLOT(s) ((FEAT & F_PHFILT) ? c.slot[s] : (s))
#define NSLOTS_BEFORE(s) ((FEAT & F_PHFILT) ? c.nslots_before[s] : (s))
    // without phantom filtering segment k of a packet is interface k

    // ---- object surface, raytrace.py:145-158 -----------------------------
    int status = live ? ROX_OK : 255, fail_surf = -1;
    v3 bp, bn, bd = dir0;               // before_pt, before_normal, before_dir
    int b4_mode = ROX_DUMMY;
    if (live) {
        const auto row = tbl;
        if (c.intersect_obj) {
            const int2 mp{ints_of(row)[0], ints_of(row)[1]};      // mode, profile
            b4_mode = mp.x;
            double s_;
            v3 df;
            bool ok;
            if ((FEAT & F_PHASE) && mp.y == ROX_THINLENS) {     // thinlens.py:131-134
                s_ = -pt0.z / dir0.z;
                bp = v3{pt0.x + s_ * dir0.x, pt0.y + s_ * dir0.y, pt0.z + s_ * dir0.z};
                ok = true;
            } else if (!kPoly || mp.y <= ROX_CONIC) {
                ok = quadric_hit(mp.y == ROX_CONIC, row[O_CV], row[O_CC], row[O_EC], pt0, dir0,
                                 row[O_ZDIR], s_, bp);
                const double k = (mp.y == ROX_CONIC) ? (row[O_CC] + 1.0) * row[O_CV] : row[O_CV];
                const double ncv = row[O_CR];       // -cv as df forms it (roxtrace.hip, device row)
                df = v3{ncv * bp.x, ncv * bp.y, 1.0 - k * bp.z};
            } else {
                ok = newton_hit<FEAT>(mp.y, row[O_CV], row[O_CC] + 1.0, row[O_EC], row[O_CR],
                                      ints_of(row)[2], row + O_COEF, pt0, dir0, c.eps, s_, bp, df);
            }
            if (!ok) {              // raised outside the try block: no packet
                status = ROX_MISSED_SURFACE;
                fail_surf = 0;
            } else if ((FEAT & F_PHASE) && mp.y == ROX_THINLENS) {
                bn = v3{0., 0., 1.};
            } else {
                bn = unit(df);
            }
        } else {
            bp = pt0;
            bn = v3{0., 0., 1.};
        }
    }
    double z_dir_before = tbl[O_ZDIR];
    double opl = 0.0, phs = 0.0;
    double acc_dst = 0.0;           // dst of the most recently appended segment
    int acc_slot = 0;
    v3 inc{0, 0, 0}, nrm{0, 0, 0}, ad = dir0;
    e.ray1_p = e.rayk_p = e.rayk_d = e.probe_p = v3{0, 0, 0};
    if (OUT_MODE == MODE_PROBE && c.probe_surf == 0)
        e.probe_p = bp;
    if (OUT_MODE == ROX_OUT_FULL && status == ROX_OK)
        so.pdn(0, bp, bd, bn);

    // ---- remaining surfaces, raytrace.py:164-229 -------------------------
    for (int surf = 1; surf < N && (kSync || status == ROX_OK); ++surf) {
        if (kSync) {
            __builtin_amdgcn_s_barrier();
            if (status != ROX_OK)
                continue;
        }
        const auto prow = tbl + (size_t)(surf - 1) * kRowDoubles;     // `before`
        const auto row = tbl + (size_t)surf * kRowDoubles;             // `after`
        const int mode = ints_of(row)[0], prof = ints_of(row)[1];
        const double cv = row[O_CV];
        const bool thin = (FEAT & F_PHASE) && prof == ROX_THINLENS;

        // :170-174 transform to the new vertex frame, closest approach
        const int rt_order = ints_of(prow)[4];
        const v3 dp{bp.x - prow[O_T], bp.y - prow[O_T + 1], bp.z - prow[O_T + 2]};
        v3 b4p, b4d;
        // rt exactly the identity (flagged on the device row when the system is created): each
        // dgemv chain fma(0, z, fma(0, y, fma(1, x, 0.0))) is x + 0.0 for finite operands in
        // either column order (-0 becomes +0 as in the chain).  A non-finite component would
        // turn the other rows into NaN through 0 * inf, so the short cut is taken only when
        // every active lane's six components are finite (their sum is: a conservative test).
        // (reduced-output modes only: in FULL mode, which is bound by its packet stores, the
        // extra branch measured 2 % slower)
        if (kIdentRt && (ints_of(prow)[5] & 2) != 0 &&
            wave_all(__builtin_isfinite(((dp.x + dp.y) + dp.z) + ((bd.x + bd.y) + bd.z)))) {
            b4p = v3{dp.x + 0.0, dp.y + 0.0, dp.z + 0.0};
            b4d = v3{bd.x + 0.0, bd.y + 0.0, bd.z + 0.0};
        } else {
            b4p = rotate(prow + O_RT, rt_order, dp);
            b4d = rotate(prow + O_RT, rt_order, bd);
        }
        const double pp_dst = -dot3(b4p, b4d);
        const v3 pp{b4p.x + pp_dst * b4d.x, b4p.y + pp_dst * b4d.y, b4p.z + pp_dst * b4d.z};

        // :181-183 intersect
        double s;
        v3 df;
        bool ok;
        if (thin) {
            s = -pp.z / b4d.z;
            inc = v3{pp.x + s * b4d.x, pp.y + s * b4d.y, pp.z + s * b4d.z};
            ok = true;
        } else if (!kPoly || prof <= ROX_CONIC) {
            ok = quadric_hit(prof == ROX_CONIC, cv, row[O_CC], row[O_EC], pp, b4d,
                             z_dir_before, s, inc);
        } else {
            ok = newton_hit<FEAT>(prof, cv, row[O_CC] + 1.0, row[O_EC], row[O_CR],
                                  ints_of(row)[2], row + O_COEF, pp, b4d, c.eps, s, inc, df);
        }
        const bool b4_filtered = c.filter_ph && (b4_mode == ROX_PHANTOM);
        if (!ok) {                                  // :231-237
            status = ROX_MISSED_SURFACE;
            fail_surf = surf;
            if (OUT_MODE == ROX_OUT_FULL) {
                const int sl = b4_filtered ? NSLOTS_BEFORE(surf - 1) : SLOT(surf - 1);
                if (b4_filtered)
                    so.pdn(sl, bp, bd, bn);
                so.dst(sl, pp_dst);
            }
            ROX_LEAVE;
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
        // (without phantom filtering the record of this interface -- complete or partial -- holds
        // inc_pt and the normal in slot `surf` whatever happens next: stored as they become known)
        constexpr bool kEarly = OUT_MODE == ROX_OUT_FULL && ROX_FULL_EARLY_STORES && !(FEAT & F_PHFILT);
        if (kEarly) {
            so.put(surf, 0, inc.x); so.put(surf, 1, inc.y); so.put(surf, 2, inc.z);
        }

        // :193-194 (in_gap_range, :123-132)
        {
            const int g = surf - 1;
            const bool in_gap = !(c.last_surf >= 0 && c.first_surf == c.last_surf) &&
                                g >= c.first_surf && (c.last_surf < 0 || g < c.last_surf);
            if (in_gap)
                opl += nwl[surf - 1] * dst_b4;
        }

        // :196 normal = normalize(df(inc_pt))
        if (thin) {
            nrm = v3{0., 0., 1.};
        } else {
            if (!kPoly || prof <= ROX_CONIC) {
                const double k = (prof == ROX_CONIC) ? (row[O_CC] + 1.0) * cv : cv;
                const double ncv = row[O_CR];       // -cv as df forms it (roxtrace.hip, device row)
                df = v3{ncv * inc.x, ncv * inc.y, 1.0 - k * inc.z};
            }
            nrm = unit(df);
        }
        if (kEarly) {
            so.put(surf, 7, nrm.x); so.put(surf, 8, nrm.y); so.put(surf, 9, nrm.z);
        }

        // :198-202 aperture test (in_surface_range, :134-142)
        if (c.check_ap && surf >= c.first_surf && (c.last_surf < 0 || surf <= c.last_surf) &&
            mode != ROX_PHANTOM) {
            const int n_ap = (FEAT & F_APLIST) ? ints_of(row)[3] : 0;
            const bool in = n_ap > 0
                ? inside_aperture_list(row, n_ap, inc.x, inc.y, c.fuzz,
                                       c.aplthr ? c.aplthr + (size_t)surf * ROX_MAX_AP : nullptr)
                : (inc.x * inc.x + inc.y * inc.y) <= c.apthr[surf];   // sqrt-free, see above
            if (!in)
                status = ROX_BLOCKED;               // :247-251
        }

        // :205-221 phase element, or refract / reflect / pass through
        if (status == ROX_OK) {
            if ((FEAT & F_PHASE) && ints_of(row + O_PH)[0] != ROX_PH_NONE) {
                double dW = 0.0;
                const auto pc = c.phc + ((PER_RAY_WVL ? (size_t)wi * N : 0) + surf) * kPhaseConsts;
                const int rc = apply_phase(row + O_PH, pc, inc, b4d, nrm, z_dir_before,
                                           c.wvls[wi], nwl[surf - 1], nwl[surf], mode, ad, dW);
                if (rc == PHASE_OK)
                    phs += dW;
                else
                    status = (rc == PHASE_TIR) ? ROX_TIR : ROX_EVANESCENT;    // :253-257
            } else if (mode == ROX_REFLECT) {
                ad = mirror(b4d, nrm);
            } else if (mode == ROX_TRANSMIT) {
                if (!refract(b4d, nrm, nwl[surf - 1], nwl[surf], ad))
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
                if (kEarly) {
                    so.put(sl, 3, bd.x); so.put(sl, 4, bd.y); so.put(sl, 5, bd.z);
                } else {
                    so.pdn(sl, inc, bd, nrm);
                }
                so.dst(sl, 0.0);
            }
            ROX_LEAVE;
        }

        if (OUT_MODE == ROX_OUT_OPD || OUT_MODE == ROX_OUT_FAN) {
            if (surf == 1)
                e.ray1_p = inc;
            if (surf == N - 2) {
                e.rayk_p = inc;
                e.rayk_d = ad;
            }
        }
        if (OUT_MODE == MODE_PROBE && surf == c.probe_surf)
            e.probe_p = inc;
        // :223-229 roll
        bp = inc; bd = ad;
        if (FEAT & F_PHFILT)
            bn = nrm;           // only a filtered phantom's late append needs it
        z_dir_before = row[O_ZDIR];
        b4_mode = mode;
        if (OUT_MODE == ROX_OUT_FULL) {
            const bool cur_filtered = c.filter_ph && (mode == ROX_PHANTOM) && surf < N - 1;
            if (kEarly) {
                so.put(surf, 3, ad.x); so.put(surf, 4, ad.y); so.put(surf, 5, ad.z);
            } else if (!cur_filtered) {
                so.pdn(SLOT(surf), inc, ad, nrm);
            }
        }
    }
    if (OUT_MODE == ROX_OUT_FULL && status == ROX_OK)   // :259-262
        so.dst(SLOT(N - 1), 0.0);
#undef SLOT
#undef NSLOTS_BEFORE
#undef ROX_LEAVE
    e.status = status;
    e.fail_surf = fail_surf;
    e.opl = opl;
    e.phs = phs;
    e.inc = inc; e.ad = ad; e.nrm = nrm;
}

// ------------------------------------------------------------------ one ray, reduced outputs, exact
// trace_ray() for the reduced-output modes (LAST, HITS, packed hits, OPD, FAN) in the loop shape
// the tolerance-mode kernels showed to pay (trace_ray_fast below): the arithmetic is trace_ray()'s,
// operation for operation -- so every bit of every output is the same -- but a missed surface, a
// blocked ray or a total internal reflection is a FLAG the iteration ends on, not a way out of
// the middle of it.  The operations behind a raised flag run on (on NaN where a radicand was
// negative) in lanes nobody reads any more: no saved exec masks inside the iteration, no
// register copies where paths meet, and the incidence point / direction are the loop-carried
// before_pt / before_dir themselves.  FULL packets (segment stores, the late append of
// raytrace.py:185-191, phantom filtering) and MODE_PROBE stay with trace_ray().
#ifndef ROX_REDUCED_STRAIGHT     // 0: the reduced-output modes of the exact instances keep trace_ray()
#define ROX_REDUCED_STRAIGHT 1
#endif

// profiles.py:321-336 / 580-593 as quadric_root() / quadric_hit() compute them, without the early exit
__device__ __forceinline__ bool quadric_hit_sl(bool conic, double cv, double cc, double ec,
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
    const double rad = b * b - ax2 * cx2;
    const double den = z_dir * slim_sqrt(rad) - b;      // (NaN where rad < 0: flagged below)
    double sq = cx2 / den;      // (through slim_div(): measured slower, EXPERIMENTS.md round 6)
    // np.errstate(divide='raise') -> FloatingPointError -> s = 0 only for a finite non-zero
    // numerator; 0/0 and nan/0 stay NaN; all three coefficients zero: s = 0 without dividing
    if (den == 0.0 && cx2 != 0.0 && isfinite(cx2))
        sq = 0.0;
    if (!((b != 0) || (cx2 != 0) || (ax2 != 0)))
        sq = 0.0;
    s = sq;
    hit = v3{p.x + s * d.x, p.y + s * d.y, p.z + s * d.z};
    return !(rad < 0.0) || !((b != 0) || (cx2 != 0) || (ax2 != 0));
}

// raytrace.py:19-30 as refract() computes it, without the early exit (false = TIR, out is NaN)
__device__ __forceinline__ bool refract_sl(const v3 &d, const v3 &nrm, double n_in, double n_out, v3 &out)
{
    const double nlen = slim_sqrt(dot3(nrm, nrm));
    const double cosI = dot3(d, nrm) / nlen;
    const double sin2 = 1.0 - cosI * cosI;
    const double rad = n_out * n_out - n_in * n_in * sin2;
    const double n_cosIp = copysign(slim_sqrt(rad), cosI);
    const double alpha = n_cosIp - n_in * cosI;
    const v3 num{n_in * d.x + alpha * nrm.x, n_in * d.y + alpha * nrm.y, n_in * d.z + alpha * nrm.z};
    out = slim_div3(num, n_out);
    return !(rad < 0.0);
}

template <int OUT_MODE, bool PER_RAY_WVL, int FEAT, class CTX>
__device__ __forceinline__ void trace_ray_reduced(const CTX &c, const v3 &pt0, const v3 &dir0, int wi,
                                                  bool live, RayEnd &e)
{
    static_assert(OUT_MODE != ROX_OUT_FULL && OUT_MODE != MODE_PROBE, "reduced-output modes only");
    constexpr int O_CV = offsetof(rox_surface, cv) / 8, O_CC = offsetof(rox_surface, cc) / 8,
                  O_EC = offsetof(rox_surface, ec) / 8, O_CR = offsetof(rox_surface, cR) / 8,
                  O_COEF = offsetof(rox_surface, coefs) / 8,
                  O_RT = offsetof(rox_surface, rt) / 8, O_T = offsetof(rox_surface, t) / 8,
                  O_ZDIR = offsetof(rox_surface, z_dir) / 8,
                  O_PH = offsetof(rox_surface, ph) / 8;
    constexpr bool kPoly = (FEAT & F_POLY) != 0;
    const int N = c.N;
    const auto tbl = c.tbl;
    tblp nwl = PER_RAY_WVL ? c.ntab + (size_t)wi * N : c.ntab;

    // ---- object surface, raytrace.py:145-158 (before_normal is only ever stored: not needed)
    int status = live ? ROX_OK : 255, fail_surf = -1;
    v3 inc = pt0, ad = dir0, nrm{0, 0, 0};
    if (live && c.intersect_obj) {
        const auto row = tbl;
        const int prof = ints_of(row)[1];
        double s_;
        v3 df, bp;
        bool ok;
        if ((FEAT & F_PHASE) && prof == ROX_THINLENS) {     // thinlens.py:131-134
            s_ = -pt0.z / dir0.z;
            bp = v3{pt0.x + s_ * dir0.x, pt0.y + s_ * dir0.y, pt0.z + s_ * dir0.z};
            ok = true;
        } else if (!kPoly || prof <= ROX_CONIC) {
            ok = quadric_hit(prof == ROX_CONIC, row[O_CV], row[O_CC], row[O_EC], pt0, dir0,
                             row[O_ZDIR], s_, bp);
        } else {
            ok = newton_hit<FEAT>(prof, row[O_CV], row[O_CC] + 1.0, row[O_EC], row[O_CR],
                                  ints_of(row)[2], row + O_COEF, pt0, dir0, c.eps, s_, bp, df);
        }
        if (!ok) {                          // raised outside the try block: no packet
            status = ROX_MISSED_SURFACE;
            fail_surf = 0;
        } else {
            inc = bp;
        }
    }
    double z_dir_before = tbl[O_ZDIR];
    double opl = 0.0, phs = 0.0;
    e.ray1_p = e.rayk_p = e.rayk_d = e.probe_p = v3{0, 0, 0};

    // ---- remaining surfaces, raytrace.py:164-229
    for (int surf = 1; surf < N && status == ROX_OK; ++surf) {
        const auto prow = tbl + (size_t)(surf - 1) * kRowDoubles;     // `before`
        const auto row = tbl + (size_t)surf * kRowDoubles;             // `after`
        const int mode = ints_of(row)[0], prof = ints_of(row)[1];
        const double cv = row[O_CV];
        const bool thin = (FEAT & F_PHASE) && prof == ROX_THINLENS;

        // :170-174 transform to the new vertex frame, closest approach (trace_ray(): the identity
        // short cut under the same finiteness test, the dgemv chains otherwise)
        const int rt_order = ints_of(prow)[4];
        const v3 dp{inc.x - prow[O_T], inc.y - prow[O_T + 1], inc.z - prow[O_T + 2]};
        v3 b4p, b4d;
        if (ROX_IDENT_RT && (ints_of(prow)[5] & 2) != 0 &&
            wave_all(__builtin_isfinite(((dp.x + dp.y) + dp.z) + ((ad.x + ad.y) + ad.z)))) {
            b4p = v3{dp.x + 0.0, dp.y + 0.0, dp.z + 0.0};
            b4d = v3{ad.x + 0.0, ad.y + 0.0, ad.z + 0.0};
        } else {
            b4p = rotate(prow + O_RT, rt_order, dp);
            b4d = rotate(prow + O_RT, rt_order, ad);
        }
        const double pp_dst = -dot3(b4p, b4d);
        const v3 pp{b4p.x + pp_dst * b4d.x, b4p.y + pp_dst * b4d.y, b4p.z + pp_dst * b4d.z};

        // :181-183 intersect
        double s;
        v3 df;
        bool hit;
        if (thin) {
            s = -pp.z / b4d.z;
            inc = v3{pp.x + s * b4d.x, pp.y + s * b4d.y, pp.z + s * b4d.z};
            hit = true;
        } else if (!kPoly || prof <= ROX_CONIC) {
            hit = quadric_hit_sl(prof == ROX_CONIC, cv, row[O_CC], row[O_EC], pp, b4d, z_dir_before, s, inc);
        } else {
            hit = newton_hit<FEAT>(prof, cv, row[O_CC] + 1.0, row[O_EC], row[O_CR],
                                   ints_of(row)[2], row + O_COEF, pp, b4d, c.eps, s, inc, df);
        }
        // :193-194 (in_gap_range, :123-132); a ray that missed keeps the path it had (:231-237)
        {
            const int g = surf - 1;
            const bool in_gap = !(c.last_surf >= 0 && c.first_surf == c.last_surf) &&
                                g >= c.first_surf && (c.last_surf < 0 || g < c.last_surf);
            if (in_gap) {
                const double dst_b4 = pp_dst + s;
                const double opl_new = opl + nwl[surf - 1] * dst_b4;
                opl = hit ? opl_new : opl;
            }
        }

        // :196 normal = normalize(df(inc_pt))
        if (thin) {
            nrm = v3{0., 0., 1.};
        } else {
            if (!kPoly || prof <= ROX_CONIC) {
                const double k = (prof == ROX_CONIC) ? (row[O_CC] + 1.0) * cv : cv;
                const double ncv = row[O_CR];       // -cv as df forms it (roxtrace.hip, device row)
                df = v3{ncv * inc.x, ncv * inc.y, 1.0 - k * inc.z};
            }
            nrm = unit(df);
        }

        // :198-202 aperture test (in_surface_range, :134-142)
        bool blocked = false;
        if (c.check_ap && surf >= c.first_surf && (c.last_surf < 0 || surf <= c.last_surf) &&
            mode != ROX_PHANTOM) {
            const int n_ap = (FEAT & F_APLIST) ? ints_of(row)[3] : 0;
            blocked = n_ap > 0
                ? !inside_aperture_list(row, n_ap, inc.x, inc.y, c.fuzz,
                                        c.aplthr ? c.aplthr + (size_t)surf * ROX_MAX_AP : nullptr)
                : !((inc.x * inc.x + inc.y * inc.y) <= c.apthr[surf]);
        }

        // :205-221 phase element, or refract / reflect / pass through (wave-uniform branches)
        bool tir = false;
        int ph_fail = ROX_OK;
        if ((FEAT & F_PHASE) && ints_of(row + O_PH)[0] != ROX_PH_NONE) {
            if (hit && !blocked) {
                double dW = 0.0;
                const auto pc = c.phc + ((PER_RAY_WVL ? (size_t)wi * N : 0) + surf) * kPhaseConsts;
                v3 out = ad;
                const int rc = apply_phase(row + O_PH, pc, inc, b4d, nrm, z_dir_before,
                                           c.wvls[wi], nwl[surf - 1], nwl[surf], mode, out, dW);
                ad = out;
                if (rc == PHASE_OK)
                    phs += dW;
                else
                    ph_fail = (rc == PHASE_TIR) ? ROX_TIR : ROX_EVANESCENT;       // :253-257
            }
        } else if (mode == ROX_REFLECT) {
            ad = mirror(b4d, nrm);
        } else if (mode == ROX_TRANSMIT) {
            tir = !refract_sl(b4d, nrm, nwl[surf - 1], nwl[surf], ad);              // :239-245
        } else {
            ad = b4d;
        }
        // :231-257 the first failure in the reference's order: miss, blocked, TIR / evanescent
        const int st = !hit ? (int)ROX_MISSED_SURFACE
                     : blocked ? (int)ROX_BLOCKED
                     : tir ? (int)ROX_TIR : ph_fail;
        if (st != ROX_OK) {
            status = st;
            fail_surf = surf;
            break;
        }
        if (OUT_MODE == ROX_OUT_OPD || OUT_MODE == ROX_OUT_FAN) {
            if (surf == 1)
                e.ray1_p = inc;
            if (surf == N - 2) {
                e.rayk_p = inc;
                e.rayk_d = ad;
            }
        }
        z_dir_before = row[O_ZDIR];
    }
    e.status = status;
    e.fail_surf = fail_surf;
    e.opl = opl;
    e.phs = phs;
    e.inc = inc; e.ad = ad; e.nrm = nrm;
}

// ------------------------------------------------------------------ one ray, tolerance mode
// trace_ray() for the F_FAST instances: the same loop (raytrace.py:83-264) over the same table,
// reduced-output modes only -- no packet stores, hence no segment bookkeeping -- with the
// arithmetic of the section "tolerance mode" above.  c.mu / c.mu2 = n_in / n_out and its square
// per interface, staged by the workgroup.
template <int OUT_MODE, bool PER_RAY_WVL, int FEAT, class CTX>
__device__ __forceinline__ void trace_ray_fast(const CTX &c, const SegOut &so, const v3 &pt0,
                                               const v3 &dir0, int wi, bool live, RayEnd &e)
{
    static_assert(OUT_MODE != MODE_PROBE, "the searches keep the exact arithmetic");
    // FULL packets (round 6): the segment stores of trace_ray() -- [p, d, dst, normal] per
    // interface, the distance of a segment known only at the next intersection, the partial
    // record [inc_pt, before_dir, 0.0, normal] where a ray is blocked or totally reflected -- in
    // this loop.  Without phantom filtering (the host sends ROX_FILTER_PHANTOMS launches to the
    // exact kernels): segment k is interface k.  The workgroup's waves meet at a barrier before
    // every interface as in trace_ray(), so every wave runs every iteration.
    constexpr bool kFull = OUT_MODE == ROX_OUT_FULL;
    constexpr int O_CV = offsetof(rox_surface, cv) / 8, O_CC = offsetof(rox_surface, cc) / 8,
                  O_EC = offsetof(rox_surface, ec) / 8, O_CR = offsetof(rox_surface, cR) / 8,
                  O_COEF = offsetof(rox_surface, coefs) / 8,
                  O_RT = offsetof(rox_surface, rt) / 8, O_T = offsetof(rox_surface, t) / 8,
                  O_ZDIR = offsetof(rox_surface, z_dir) / 8,
                  O_PH = offsetof(rox_surface, ph) / 8;
    constexpr bool kPoly = (FEAT & F_POLY) != 0;
    constexpr bool kSync = kFull && (kPoly ? ROX_WG_SYNC_POLY : ROX_WG_SYNC);
    const int N = c.N;
    const auto tbl = c.tbl;
    tblp nwl = PER_RAY_WVL ? c.ntab + (size_t)wi * N : c.ntab;
    tblp muw = PER_RAY_WVL ? c.mu + (size_t)wi * 3 * N : c.mu;      // [N] mu, [N] mu^2, [N] sg

    // The object surface and the transfer to interface 1 keep the REFERENCE's arithmetic
    // (trace_ray()'s operations in its order, correctly rounded): an object at infinity sits
    // 1e10 system units away (rayoptics' convention), so `before_pt - t`, `-dot(b4_pt, b4_dir)`
    // and `b4_pt + pp_dst * b4_dir` cancel numbers of that size down to the lens' own, and what
    // is left carries ~1e-6 of rounding that depends on the operation order.  That noise is part
    // of the reference's answer; any other order lands 1e-7 away from it.  Once per ray.
    int status = live ? ROX_OK : 255, fail_surf = -1;
    v3 bp = pt0, bd = dir0;
    v3 pp1{0, 0, 0}, b4d1 = dir0;
    double pp_dst1 = 0.0;
    if (live) {
        const auto row = tbl;
        v3 bn{0., 0., 1.};
        if (c.intersect_obj) {                  // raytrace.py:145-158 (the normal: FULL packets only)
            const int prof = ints_of(row)[1];
            double s_;
            v3 df{0., 0., 1.};
            bool ok;
            if ((FEAT & F_PHASE) && prof == ROX_THINLENS) {
                s_ = -pt0.z / dir0.z;
                bp = v3{pt0.x + s_ * dir0.x, pt0.y + s_ * dir0.y, pt0.z + s_ * dir0.z};
                ok = true;
            } else if (!kPoly || prof <= ROX_CONIC) {
                ok = quadric_hit(prof == ROX_CONIC, row[O_CV], row[O_CC], row[O_EC], pt0, dir0,
                                 row[O_ZDIR], s_, bp);
                if (kFull) {
                    const double k = (prof == ROX_CONIC) ? (row[O_CC] + 1.0) * row[O_CV] : row[O_CV];
                    df = v3{row[O_CR] * bp.x, row[O_CR] * bp.y, 1.0 - k * bp.z};
                }
            } else {
                // (an aspheric OBJECT surface -- never at infinity -- takes the tolerance-mode
                // iteration: the exact one would set the register budget of the whole kernel)
                ok = newton_hit_f<FEAT>(prof, row[O_CV], row[O_CC] + 1.0, row[O_EC], row[O_CR],
                                        ints_of(row)[2], row + O_COEF, pt0, dir0, c.eps, s_, bp, df);
            }
            if (!ok) {
                status = ROX_MISSED_SURFACE;
                fail_surf = 0;
            } else if (kFull && !((FEAT & F_PHASE) && prof == ROX_THINLENS)) {
                bn = unit_f(df);
            }
        }
        if (kFull && status == ROX_OK)
            so.pdn(0, bp, bd, bn);
        const v3 dp{bp.x - row[O_T], bp.y - row[O_T + 1], bp.z - row[O_T + 2]};
        const v3 b4p = rotate(row + O_RT, ints_of(row)[4], dp);
        b4d1 = rotate(row + O_RT, ints_of(row)[4], bd);
        pp_dst1 = -dot3(b4p, b4d1);
        pp1 = v3{b4p.x + pp_dst1 * b4d1.x, b4p.y + pp_dst1 * b4d1.y, b4p.z + pp_dst1 * b4d1.z};
    }
    double z_dir_before = tbl[O_ZDIR];
    double opl = 0.0, phs = 0.0;
    // The loop body is straight-line per lane: a missed surface, a blocked ray or a total
    // internal reflection is a FLAG -- the arithmetic behind it runs on (NaN for a negative
    // radicand), and the iteration ends with one status select and one exit test.  `inc` and
    // `ad` are the loop-carried point and direction themselves (the reference's before_pt /
    // before_dir): a lane that fails leaves with values nobody reads (every consumer asks its
    // status first), so nothing has to be copied aside for it.
    v3 inc = bp, nrm{0, 0, 0}, ad = dir0;
    e.ray1_p = e.rayk_p = e.rayk_d = e.probe_p = v3{0, 0, 0};

    // one interface, from the closest approach `pp` (at distance pp_dst along b4d from the previous
    // intersection) on; false = the ray ends here.  Inlined twice: for interface 1, whose transfer
    // was made above in the reference's arithmetic, and in the loop -- so that the three vectors of
    // that first transfer are not carried, 14 registers wide, through every later interface.
    auto interface = [&](const int surf, const v3 &pp, const v3 &b4d, const double pp_dst)
        __attribute__((always_inline)) -> bool {
        const auto row = tbl + (size_t)surf * kRowDoubles;
        const int mode = ints_of(row)[0], prof = ints_of(row)[1];
        const double cv = row[O_CV];
        const bool thin = (FEAT & F_PHASE) && prof == ROX_THINLENS;
        const v3 bd0 = ad;          // before_dir in the previous frame (FULL: a partial record stores it)

        // :181-183 intersect
        double s;
        v3 df;
        bool hit;
        if (thin) {
            s = -pp.z * rcp_f(b4d.z);
            inc = v3{fma(s, b4d.x, pp.x), fma(s, b4d.y, pp.y), fma(s, b4d.z, pp.z)};
            hit = true;
        } else if (!kPoly || prof <= ROX_CONIC) {
            hit = quadric_hit_f(prof == ROX_CONIC, cv, row[O_CC], row[O_EC], pp, b4d, z_dir_before, s, inc);
        } else {
            hit = newton_hit_f<FEAT>(prof, cv, row[O_CC] + 1.0, row[O_EC], row[O_CR],
                                     ints_of(row)[2], row + O_COEF, pp, b4d, c.eps, s, inc, df);
        }
        // :193-194 (in_gap_range, :123-132); a ray that missed keeps the path it had (:231-237)
        {
            const int g = surf - 1;
            const bool in_gap = !(c.last_surf >= 0 && c.first_surf == c.last_surf) &&
                                g >= c.first_surf && (c.last_surf < 0 || g < c.last_surf);
            if (in_gap) {
                const double opl_new = fma(nwl[surf - 1], pp_dst + s, opl);
                opl = hit ? opl_new : opl;
            }
        }
        // :185-191 the previous segment is completed only now (:231-237: up to the closest
        // approach where the surface was missed)
        constexpr bool kEarly = kFull && ROX_FULL_EARLY_STORES;
        if (kFull)
            so.dst(surf - 1, hit ? pp_dst + s : pp_dst);
        if (kEarly && hit) {
            so.put(surf, 0, inc.x); so.put(surf, 1, inc.y); so.put(surf, 2, inc.z);
        }

        // :196 normal = normalize(df(inc_pt)); a sphere's gradient has unit length on the sphere
        if (thin) {
            nrm = v3{0., 0., 1.};
        } else if (!kPoly || prof <= ROX_CONIC) {
            const double k = (prof == ROX_CONIC) ? (row[O_CC] + 1.0) * cv : cv;     // (wave-uniform)
            nrm = v3{-cv * inc.x, -cv * inc.y, fma(-k, inc.z, 1.0)};
            if (prof == ROX_CONIC)
                nrm = unit_f(nrm);
        } else {
            nrm = unit_f(df);
        }

        if (kEarly && hit) {
            so.put(surf, 7, nrm.x); so.put(surf, 8, nrm.y); so.put(surf, 9, nrm.z);
        }

        // :198-202 aperture test (in_surface_range, :134-142)
        bool blocked = false;
        if (c.check_ap && surf >= c.first_surf && (c.last_surf < 0 || surf <= c.last_surf) &&
            mode != ROX_PHANTOM) {
            const int n_ap = (FEAT & F_APLIST) ? ints_of(row)[3] : 0;
            blocked = n_ap > 0
                ? !inside_aperture_list(row, n_ap, inc.x, inc.y, c.fuzz,
                                        c.aplthr ? c.aplthr + (size_t)surf * ROX_MAX_AP : nullptr)
                : !(fma(inc.y, inc.y, inc.x * inc.x) <= c.apthr[surf]);
        }

        // :205-221 phase element (the exact code: rare), or refract / reflect / pass through
        bool tir = false;
        int ph_fail = ROX_OK;
        if ((FEAT & F_PHASE) && ints_of(row + O_PH)[0] != ROX_PH_NONE) {
            if (hit && !blocked) {
                double dW = 0.0;
                const auto pc = c.phc + ((PER_RAY_WVL ? (size_t)wi * N : 0) + surf) * kPhaseConsts;
                v3 out = ad;
                const int rc = apply_phase(row + O_PH, pc, inc, b4d, nrm, z_dir_before,
                                           c.wvls[wi], nwl[surf - 1], nwl[surf], mode, out, dW);
                ad = out;
                if (rc == PHASE_OK)
                    phs += dW;
                else
                    ph_fail = (rc == PHASE_TIR) ? ROX_TIR : ROX_EVANESCENT;       // :253-257
            }
        } else if (mode == ROX_TRANSMIT || mode == ROX_REFLECT) {                   // (wave-uniform)
            tir = !interact_f(b4d, nrm, muw[surf], muw[N + surf], muw[2 * N + surf], ad);   // :239-245
        } else {
            ad = b4d;       // dummy / phantom: the direction as it is, whatever the normal (:215-221)
        }
        // :231-257 the first failure in the reference's order: miss, blocked, TIR / evanescent
        const int st = !hit ? (int)ROX_MISSED_SURFACE
                     : blocked ? (int)ROX_BLOCKED
                     : tir ? (int)ROX_TIR : ph_fail;
        if (st != ROX_OK) {
            status = st;
            fail_surf = surf;
            if (kFull && hit) {         // :239-257 partial record: [inc_pt, before_dir, 0.0, normal]
                if (kEarly) {
                    so.put(surf, 3, bd0.x); so.put(surf, 4, bd0.y); so.put(surf, 5, bd0.z);
                } else {
                    so.pdn(surf, inc, bd0, nrm);
                }
                so.dst(surf, 0.0);
            }
            return false;
        }
        if (kEarly) {
            so.put(surf, 3, ad.x); so.put(surf, 4, ad.y); so.put(surf, 5, ad.z);
        } else if (kFull) {
            so.pdn(surf, inc, ad, nrm);
        }
        if (OUT_MODE == ROX_OUT_OPD || OUT_MODE == ROX_OUT_FAN) {
            if (surf == 1)
                e.ray1_p = inc;
            if (surf == N - 2) {
                e.rayk_p = inc;
                e.rayk_d = ad;
            }
        }
        z_dir_before = row[O_ZDIR];
        return true;
    };

    if constexpr (kSync) {
        // every wave of the workgroup reaches every barrier; lanes whose ray has ended idle
        bool alive = false;
        if (N > 1) {
            __builtin_amdgcn_s_barrier();
            if (status == ROX_OK)
                alive = interface(1, pp1, b4d1, pp_dst1);
        }
        for (int surf = 2; surf < N; ++surf) {
            __builtin_amdgcn_s_barrier();
            if (!alive)
                continue;
            const auto prow = tbl + (size_t)(surf - 1) * kRowDoubles;
            v3 b4p{inc.x - prow[O_T], inc.y - prow[O_T + 1], inc.z - prow[O_T + 2]};
            v3 b4d = ad;
            if ((ints_of(prow)[5] & 2) == 0) {
                b4p = rotate_f(prow + O_RT, b4p);
                b4d = rotate_f(prow + O_RT, ad);
            }
            const double pp_dst = -dot3_f(b4p, b4d);
            const v3 pp{fma(pp_dst, b4d.x, b4p.x), fma(pp_dst, b4d.y, b4p.y), fma(pp_dst, b4d.z, b4p.z)};
            alive = interface(surf, pp, b4d, pp_dst);
        }
    } else if (status == ROX_OK && N > 1 && interface(1, pp1, b4d1, pp_dst1)) {
        for (int surf = 2; surf < N; ++surf) {
            // :170-174 transform to the new vertex frame, closest approach to its origin
            const auto prow = tbl + (size_t)(surf - 1) * kRowDoubles;
            v3 b4p{inc.x - prow[O_T], inc.y - prow[O_T + 1], inc.z - prow[O_T + 2]};
            v3 b4d = ad;
            if ((ints_of(prow)[5] & 2) == 0) {  // (identity rotations are flagged on the device row)
                b4p = rotate_f(prow + O_RT, b4p);
                b4d = rotate_f(prow + O_RT, ad);
            }
            const double pp_dst = -dot3_f(b4p, b4d);
            const v3 pp{fma(pp_dst, b4d.x, b4p.x), fma(pp_dst, b4d.y, b4p.y), fma(pp_dst, b4d.z, b4p.z)};
            if (!interface(surf, pp, b4d, pp_dst))
                break;
        }
    }
    if (kFull && status == ROX_OK)      // :259-262
        so.dst(N - 1, 0.0);
    e.status = status;
    e.fail_surf = fail_surf;
    e.opl = opl;
    e.phs = phs;
    e.inc = inc; e.ad = ad; e.nrm = nrm;
}

// ------------------------------------------------------------------ ray start
// opticalspec.py:1339-1353 apply_vignetting + :289-400 ray_start_from_osp +
// trace.py:302-308.  pupil (px, py) is updated in place (the reference's quirk).
template <class FLD>     // rox_field (kernel argument, batch item in constant memory, aim problem)
__device__ __forceinline__ void ray_start(FLD &f, uint32_t flags, double &px,
                                          double &py, v3 &pt0, v3 &dir0)
{
    if (flags & ROX_APPLY_VIGNETTING) {         // opticalspec.py:1339-1353
        if (px < 0.0) { if (f.vlx != 0.0) px *= (1.0 - f.vlx); }
        else          { if (f.vux != 0.0) px *= (1.0 - f.vux); }
        if (py < 0.0) { if (f.vly != 0.0) py *= (1.0 - f.vly); }
        else          { if (f.vuy != 0.0) py *= (1.0 - f.vuy); }
    }
    pt0 = v3{f.pt0[0], f.pt0[1], f.pt0[2]};
    const int kind = f.kind;                    // wave-uniform
    if (kind <= ROX_FLD_AIM_PT) {
        v3 pt1;
        if (kind == ROX_FLD_EPD) {              // :358-366
            pt1 = v3{f.eprad * px + f.aim[0], f.eprad * py + f.aim[1], f.z_enp};
        } else if (kind == ROX_FLD_AIM_PT) {    // :334-337
            pt1 = v3{px, py, f.z_enp};
        } else {                                // :340-356 wide angle
            pt1 = rotate(f.rot, f.rot_order, v3{f.eprad * px, f.eprad * py, f.eprad * 0.});
            pt1.z -= f.z_enp;
        }
        dir0 = unit(v3{pt1.x - pt0.x, pt1.y - pt0.y, pt1.z - pt0.z});
    } else {                                    // :368-398 angular measures
        double dx, dy;
        if (kind == ROX_FLD_NA) {
            dx = f.eprad * px;
            dy = f.eprad * py;
        } else if (kind == ROX_FLD_FNO) {
            const double slope = f.eprad;
            const double ax = px * slope, ay = py * slope;
            const double hypt = sqrt(1 + ax * ax + ay * ay);
            dx = slope * px / hypt;
            dy = slope * py / hypt;
        } else {
            dx = px;
            dy = py;
        }
        if (kind != ROX_FLD_AIM_DIR) {
            dx += f.cr_dir[0];
            dy += f.cr_dir[1];
        }
        const double dd = fma(dy, dy, fma(dx, dx, 0.0));    // np.dot of 2-vectors
        dir0 = v3{dx, dy, sqrt(1 - dd)};
    }
    if (kind != ROX_FLD_EPD_WIDE && dir0.z * f.z_dir0 < 0)   // trace.py:304-308
        dir0 = v3{-dir0.x, -dir0.y, -dir0.z};
}

// ------------------------------------------------------------------ compaction
// tile_state word: epoch << 32 | flag << 30 | count  (count < 2^30)
enum : uint64_t { TS_AGG = 1, TS_PREFIX = 2 };
__device__ __forceinline__ uint64_t ts_pack(uint32_t epoch, uint64_t flag, uint32_t count)
{
    return ((uint64_t)epoch << 32) | (flag << 30) | count;
}

// Where a tile's `total` packed pairs go: dst[base + excl ...), base = the pairs already in the
// buffer (earlier launches of a chunked call, earlier ROX_HITS_APPEND calls).  `cap` (rox_out.ld)
// is the capacity of dst in pairs: nothing is ever stored at or beyond it, and a call that would
// have needed more room leaves the NEGATED pair count it needed in *total_out (the last tile
// writes it), which later appending launches keep negative -- the caller sees n_hits < 0.
// Called by all kB threads of the workgroup.
template <int kB>
__device__ __forceinline__ void store_tile_pairs(d2 *dst, int64_t cap, const int64_t *base_in,
                                                 int64_t excl, int total, const d2 *stash,
                                                 int64_t *total_out)
{
    const int64_t base = base_in ? *base_in : 0;
    const int64_t at = (base < 0 ? -base : base) + excl;
    int64_t room = base < 0 ? 0 : cap - at;
    if (room > total)
        room = total;
    for (int j = threadIdx.x; j < room; j += kB)
        __builtin_nontemporal_store(stash[j], dst + at + j);
    if (total_out && threadIdx.x == 0) {
        const int64_t all = at + total;
        *total_out = (base < 0 || all > cap) ? -all : all;
    }
}

// Wave 0 of a workgroup (all 64 lanes): the number of survivors of all tiles before `tile`, by
// decoupled look-back over the predecessors' published counts; publishes this tile's inclusive
// prefix.  The tile's own count must already be published (TS_AGG; tile 0: TS_PREFIX).
__device__ __forceinline__ uint32_t look_back(uint64_t *st, uint32_t epoch, int64_t tile, int total,
                                              int lane)
{
    uint32_t excl = 0;
    if (tile != 0) {
        // Look back over the predecessors, nearest first, 512 states per round trip (8
        // independent loads per lane): sum counts until a tile that already knows its
        // inclusive prefix is met; a tile that has published nothing yet is waited for,
        // keeping the partial sum.
        int64_t look = tile - 1;
        for (bool done = false; !done;) {
            uint64_t w[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int64_t idx = look - (k * 64 + lane);
                w[k] = ts_pack(epoch, TS_PREFIX, 0);      // before tile 0
                if (idx >= 0)
                    w[k] = __hip_atomic_load(&st[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            int consumed = 512;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const uint32_t flag = (uint32_t)(w[k] >> 30) & 3u;
                const bool ready = (uint32_t)(w[k] >> 32) == epoch && flag != 0;
                const uint64_t rmask = __ballot(ready);
                const uint64_t pmask = __ballot(ready && flag == TS_PREFIX);
                const int first_not = (~rmask) ? __builtin_ctzll(~rmask) : 64;
                const int first_pref = pmask ? __builtin_ctzll(pmask) : 64;
                const int take = first_pref < first_not ? first_pref + 1 : first_not;
                uint32_t v = lane < take ? (uint32_t)(w[k] & 0x3fffffffu) : 0u;
                for (int o = 32; o > 0; o >>= 1)
                    v += __shfl_xor(v, o);
                excl += v;
                if (first_pref < first_not) {
                    done = true;
                    break;
                }
                if (first_not < 64) {       // wait for that tile, resume from it
                    consumed = k * 64 + first_not;
                    __builtin_amdgcn_s_sleep(1);
                    break;
                }
            }
            look -= consumed;
        }
        if (lane == 0)
            __hip_atomic_store(&st[tile], ts_pack(epoch, TS_PREFIX, excl + (uint32_t)total),
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    return excl;
}

// A tile's survivors (packed in `stash`, `total` of them) go to their final place: wave 0 finds
// the number of survivors of all earlier tiles by look_back(), then every thread copies pairs
// (consecutive threads, consecutive pairs).  Called by all threads of the workgroup.
template <int kB, class ARGS>
__device__ __forceinline__ void finish_tile(ARGS &a, int64_t tile, int total,
                                            const d2 *stash, int64_t n_tiles, uint32_t *s_excl)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (wave == 0) {
        const uint32_t excl = look_back(a.tile_state, a.epoch, tile, total, lane);
        if (lane == 0)
            *s_excl = excl;
    }
    __syncthreads();
    const uint32_t excl = (uint32_t)__builtin_amdgcn_readfirstlane((int)*s_excl);
    store_tile_pairs<kB>(reinterpret_cast<d2 *>(a.out.seg), a.out.ld, a.hits_base_in, (int64_t)excl,
                         total, stash, tile == n_tiles - 1 ? a.hits_total_out : nullptr);
}

// HITS_COMPACT tile geometry.  The first `want` tickets of a launch take tiles of kSmallTile
// rays: a 1024-thread workgroup then runs four waves, one per SIMD, and is through its
// surfaces in a third of the time sixteen take.  The host asks for them when the whole launch
// fits one small tile per CU (a 256 x 256 grid spreads over 256 CUs instead of 64); later
// tickets take full tiles.
constexpr int kSmallTile = 256;
__host__ __device__ inline int64_t compact_small_tiles(int64_t n_rays, int32_t want)
{
    const int64_t fit = n_rays / kSmallTile;
    return want < fit ? (want < 0 ? 0 : want) : fit;
}
__host__ __device__ inline int64_t compact_tiles(int64_t n_rays, int32_t want_small, int kB)
{
    const int64_t s = compact_small_tiles(n_rays, want_small);
    return s + (n_rays - s * kSmallTile + kB - 1) / kB;
}

// ------------------------------------------------------------------ the kernel
// the work of one workgroup on one launch item: its share of the item's ray tiles
template <int OUT_MODE, int GEN, bool PER_RAY_WVL, int FEAT, bool SMALL, class ARGS>
__device__ __forceinline__ void trace_tiles(ARGS &a)
{
    constexpr bool kCompact = (OUT_MODE == ROX_OUT_HITS_COMPACT);
    constexpr int kB = block_of(OUT_MODE, FEAT, SMALL);     // threads per workgroup = rays per tile

    const int N = a.n_ifcs;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    // LDS of a workgroup: [the table: N rows]  the indices  [phase constants]  wavelengths
    // max-aperture thresholds  [clear-aperture thresholds]  slot map  [mu, mu^2]  [stash].
    // F_GTAB leaves the table and the phase constants in global memory and keeps the circular
    // clear-aperture thresholds, which the LDS instances write into their copy of the table,
    // in an array of their own.
    constexpr bool kGtab = (FEAT & F_GTAB) != 0;
    constexpr bool kFast = (FEAT & F_FAST) != 0;
    typedef typename std::conditional<kGtab, ctblp, tblp>::type TP;
    double *tbl_w = lds;                               // [N][kRowDoubles]
    const int nw_rows = PER_RAY_WVL ? a.n_wvls : 1;
    double *ntab_w = tbl_w + (kGtab ? 0 : (size_t)N * kRowDoubles);    // [W][N] (or [N] for one wavelength)
    double *phc_w = ntab_w + (size_t)nw_rows * N;      // [W][N][4] (F_PHASE only)
    double *wvls_w = phc_w + (((FEAT & F_PHASE) && !kGtab) ? (size_t)nw_rows * N * kPhaseConsts : 0);
    double *apthr_w = wvls_w + a.n_wvls;                // [N]
    double *aplthr_w = apthr_w + N;                     // [N][ROX_MAX_AP] (F_GTAB with F_APLIST)
    int32_t *slot_w = reinterpret_cast<int32_t *>(
        aplthr_w + ((kGtab && (FEAT & F_APLIST)) ? (size_t)N * ROX_MAX_AP : 0));
    // F_FAST: (mu, mu^2, sg) of interact_f per (wavelength row, interface), behind the slot map
    // (2 N int32 = 8 N bytes behind an 8-byte aligned start: aligned as it stands -- a pointer
    // rounded through an integer loses its LDS address space and is read with flat loads)
    double *mu_w = reinterpret_cast<double *>(slot_w + 2 * N);
    // HITS_COMPACT: two tiles' worth of packed (x, y) pairs behind them, 16-byte aligned: the
    // offset is rounded in doubles (lds is 16-byte aligned, every region before is whole doubles)
    double *end_w = mu_w + (kFast ? (size_t)nw_rows * 3 * N : 0);
    d2 *stash_w = reinterpret_cast<d2 *>(lds + (((size_t)(end_w - lds) + 1) & ~size_t(1)));

    // stage the surface table once per workgroup
    if (!kGtab)
        for (int i = threadIdx.x; i < N * kRowDoubles; i += kB)
            tbl_w[i] = a.rows[i];
    const size_t w0 = PER_RAY_WVL ? 0 : (size_t)a.wvl_idx_all * N;
    {
        for (int i = threadIdx.x; i < nw_rows * N; i += kB)
            ntab_w[i] = a.n_table[w0 + i];
        if (kFast)
            for (int i = threadIdx.x; i < nw_rows * N; i += kB) {
                const int w = i / N, sf = i - w * N;
                // (mode of the interface: rox_surface.mode is the row's first word)
                const int md = reinterpret_cast<const int32_t *>(a.rows + (size_t)sf * kRowDoubles)[0];
                const double m = (sf > 0 && md == ROX_TRANSMIT) ? a.n_table[w0 + i - 1] / a.n_table[w0 + i] : 1.0;
                mu_w[(size_t)w * 3 * N + sf] = m;
                mu_w[(size_t)w * 3 * N + N + sf] = m * m;
                mu_w[(size_t)w * 3 * N + 2 * N + sf] = (md == ROX_REFLECT) ? -1.0 : 1.0;
            }
        if ((FEAT & F_PHASE) && !kGtab)
            for (int i = threadIdx.x; i < nw_rows * N * kPhaseConsts; i += kB)
                phc_w[i] = a.ph_consts[w0 * kPhaseConsts + i];
    }
    for (int i = threadIdx.x; i < a.n_wvls; i += kB)
        wvls_w[i] = a.wvls[i];
    for (int i = threadIdx.x; i < 2 * N; i += kB)
        slot_w[i] = a.slots[i];
    for (int i = threadIdx.x; i < N; i += kB)
        apthr_w[i] = sqrt_le_threshold(
            a.rows[(size_t)i * kRowDoubles + offsetof(rox_surface, max_aperture) / 8] + a.opts.fuzz);
    if (kGtab && (FEAT & F_APLIST))
        for (int i = threadIdx.x; i < N * ROX_MAX_AP; i += kB) {
            const double *row = a.rows + (size_t)(i / ROX_MAX_AP) * kRowDoubles;
            const int k = i % ROX_MAX_AP;
            double t = 0.0;
            if (k < reinterpret_cast<const int32_t *>(row)[3]) {
                const double *ap = row + offsetof(rox_surface, ap) / sizeof(double) +
                                   (size_t)k * (sizeof(rox_aperture) / sizeof(double));
                if (reinterpret_cast<const int32_t *>(ap)[0] == ROX_AP_CIRCULAR)
                    t = sqrt_le_threshold(ap[3] + a.opts.fuzz);
            }
            aplthr_w[i] = t;
        }
    __syncthreads();
    if ((FEAT & F_APLIST) && !kGtab) {
        stage_aperture_thresholds<FEAT>(tbl_w, N, a.opts.fuzz, threadIdx.x, kB);
        __syncthreads();
    }

    CtxT<TP> c;
    if constexpr (kGtab) {
        c.tbl = (ctblp)a.rows;
        c.phc = (ctblp)a.ph_consts + w0 * kPhaseConsts;
        c.aplthr = (FEAT & F_APLIST) ? aplthr_w : nullptr;
    } else {
        c.tbl = tbl_w;
        c.phc = phc_w;
        c.aplthr = nullptr;
    }
    c.ntab = ntab_w; c.wvls = wvls_w; c.apthr = apthr_w;
    c.slot = slot_w; c.nslots_before = slot_w + N;
    c.mu = mu_w;
    c.N = N;
    const uint32_t flags = a.opts.flags;
    c.check_ap = flags & ROX_CHECK_APERTURES;
    c.intersect_obj = flags & ROX_INTERSECT_OBJ;
    c.filter_ph = (FEAT & F_PHFILT) && (flags & ROX_FILTER_PHANTOMS);
    c.first_surf = a.opts.first_surf; c.last_surf = a.opts.last_surf;
    c.eps = a.opts.eps; c.fuzz = a.opts.fuzz;
    c.probe_surf = -1;
    const int64_t ld = a.out.ld;
    const int64_t n_small = kCompact ? compact_small_tiles(a.n_rays, a.small_tiles) : 0;
    const int64_t n_tiles = kCompact ? compact_tiles(a.n_rays, a.small_tiles, kB)
                          : (a.n_rays + kB - 1) / kB;

    // HITS_COMPACT: tiles are handed out by ticket, so that the tile a workgroup
    // waits for in the look-back is always held by a running workgroup
    __shared__ int64_t s_tile;
    __shared__ int32_t s_wcnt[kB / 64];
    __shared__ uint32_t s_excl;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr bool kDefer = kCompact && ROX_COMPACT_DEFER;
    int64_t pend_tile = -1, pend_it = 0;    // HITS_COMPACT: the tile whose finish is deferred
    int pend_total = 0;

    for (int64_t it = 0;; ++it) {
        int64_t tile;
        if (kCompact) {
            if (threadIdx.x == 0)
                s_tile = (int64_t)atomicAdd(&a.ticket[0], 1u);
            __syncthreads();
            // (wave-uniform by construction: keep it in scalar registers across the trace)
            tile = (int64_t)__builtin_amdgcn_readfirstlane((int)s_tile);
        } else {
            tile = (int64_t)blockIdx.x + it * gridDim.x;
        }
        if (tile >= n_tiles)
            break;
        int64_t r = tile * kB + threadIdx.x;
        bool active = r < a.n_rays;
        if (kCompact) {
            const bool small = tile < n_small;
            r = (small ? tile * kSmallTile : n_small * kSmallTile + (tile - n_small) * kB) + threadIdx.x;
            active = r < a.n_rays && (!small || threadIdx.x < kSmallTile);
        }
        RayEnd e;
        SegOut so;
        so.base = reinterpret_cast<char *>(a.out.seg);
        so.row_bytes = ld * 8;
        so.voff = (uint32_t)r * 8u;
        v3 pt0{0, 0, 0}, dir0{0, 0, 1};
        int wi = a.wvl_idx_all;
        bool wi_ok = true;
        if (active) {
            // ---- ray start ---------------------------------------------------
            const int64_t rg = a.ray_base + r;
            if (GEN == GEN_PUPIL) {
                double px, py;
                if (a.axis_kind == AXIS_PRODUCT) {
                    px = a.px[a.row_begin + rg / a.axis_num];
                    py = a.py[rg % a.axis_num];
                } else {
                    px = a.px[rg];
                    py = a.py[rg];
                }
                ray_start(a.fld, flags, px, py, pt0, dir0);
                if (!kCompact && a.out.pupil) {
                    a.out.pupil[r] = px;
                    a.out.pupil[ld + r] = py;
                }
            } else {
                pt0 = v3{a.pt0[rg], a.pt0[a.in_ld + rg], a.pt0[2 * a.in_ld + rg]};
                dir0 = v3{a.dir0[rg], a.dir0[a.in_ld + rg], a.dir0[2 * a.in_ld + rg]};
            }
            if (PER_RAY_WVL) {
                wi = a.wvl_idx[rg];
                wi_ok = (wi >= 0 && wi < a.n_wvls);     // a bad index never reaches the table
                if (!wi_ok)
                    wi = 0;
            }
        }
        if constexpr (kFast)
            trace_ray_fast<OUT_MODE, PER_RAY_WVL, FEAT>(c, so, pt0, dir0, wi, active, e);
        else if constexpr (ROX_REDUCED_STRAIGHT && OUT_MODE != ROX_OUT_FULL)
            trace_ray_reduced<OUT_MODE, PER_RAY_WVL, FEAT>(c, pt0, dir0, wi, active, e);
        else
            trace_ray<OUT_MODE, PER_RAY_WVL, FEAT>(c, so, pt0, dir0, wi, active, e);
        if (active) {
            if (PER_RAY_WVL && !wi_ok) {
                // reported as a miss at the object surface; no packet
                e.status = ROX_MISSED_SURFACE;
                e.fail_surf = 0;
                e.opl = __builtin_nan("");
            }

            // ---- per-ray outputs ---------------------------------------------
            if (e.status == ROX_OK) {
                if (OUT_MODE == ROX_OUT_LAST) {             // trace.py:214-217
                    so.pdn(0, e.inc, e.ad, e.nrm);
                    so.dst(0, 0.0);
                } else if (OUT_MODE == ROX_OUT_OPD || OUT_MODE == ROX_OUT_FAN) {
                    const double op = e.phs + e.opl;
                    so.put(0, OUT_MODE == ROX_OUT_FAN ? 2 : 0,
                           a.opts.wf.kind == ROX_WF_FINITE            // (wave-uniform)
                           ? wave_abr_finite_pup<kFast>(a.opts.wf, e.ray1_p, dir0, e.rayk_p, e.rayk_d, op)
                           : wave_abr_inf_ref<kFast>(a.opts.wf, e.ray1_p, dir0, e.rayk_p, e.rayk_d,
                                                     e.inc, e.ad, op));
                    if (OUT_MODE == ROX_OUT_FAN) {          // analyses.py:258-262
                        const double dist = kFast ? a.opts.foc * rcp_f(e.ad.z) : a.opts.foc / e.ad.z;
                        so.put(0, 0, (e.inc.x + dist * e.ad.x) - a.opts.image_pt[0]);
                        so.put(0, 1, (e.inc.y + dist * e.ad.y) - a.opts.image_pt[1]);
                    }
                } else if (OUT_MODE == ROX_OUT_HITS) {      // axisarrayfigure.py:229-238
                    const double dist = kFast ? a.opts.foc * rcp_f(e.ad.z) : a.opts.foc / e.ad.z;
                    so.put(0, 0, (e.inc.x + dist * e.ad.x) - a.opts.image_pt[0]);
                    so.put(0, 1, (e.inc.y + dist * e.ad.y) - a.opts.image_pt[1]);
                }
            }
            if (!kCompact) {
                if (a.out.op)       // op_delta = phs + opl on success; opl on failure (:236)
                    a.out.op[r] = (e.status == ROX_OK) ? e.phs + e.opl : e.opl;
                if (a.out.fail_surf)
                    a.out.fail_surf[r] = (int16_t)e.fail_surf;
            }
            if (a.out.status)
                a.out.status[r] = (uint8_t)e.status;
        }

        if (kCompact) {
            // ---- stable compaction of the hits: ballot ranks within the wave, wave counts
            // through LDS, the tile's survivors packed into an LDS stash; finish_tile() then
            // finds the tile's place by decoupled look-back and copies the stash there.
            const bool ok = active && e.status == ROX_OK;
            const uint64_t mask = __ballot(ok);
            const int lrank = __popcll(mask & ((1ull << lane) - 1ull));
            if (lane == 0)
                s_wcnt[wave] = __popcll(mask);
            __syncthreads();
            int woff = 0, total = 0;
#pragma unroll
            for (int w = 0; w < kB / 64; ++w) {
                const int cnt = s_wcnt[w];
                if (w < wave)
                    woff += cnt;
                total += cnt;
            }
            woff = __builtin_amdgcn_readfirstlane(woff);        // per-wave / per-workgroup values
            total = __builtin_amdgcn_readfirstlane(total);
            d2 *stash = stash_w + (size_t)(it & 1) * kB;
            if (ok) {
                const double dist = kFast ? a.opts.foc * rcp_f(e.ad.z) : a.opts.foc / e.ad.z;
                d2 xy;
                xy.x = (e.inc.x + dist * e.ad.x) - a.opts.image_pt[0];
                xy.y = (e.inc.y + dist * e.ad.y) - a.opts.image_pt[1];
                stash[woff + lrank] = xy;
            }
            if (threadIdx.x == 0)       // the count is published at once; tile 0's is its prefix
                __hip_atomic_store(&a.tile_state[tile],
                                   ts_pack(a.epoch, tile == 0 ? TS_PREFIX : TS_AGG, (uint32_t)total),
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __syncthreads();            // the stash is complete
            // Deferred finish: a tile looks back and copies only after the workgroup has traced
            // its *next* tile -- by then every predecessor has long published its count, so no
            // wave sits in a look-back while the SIMDs have nothing else to run.  A workgroup's
            // first tile finishes at once (early output for a destination behind PCIe).
            if (!kDefer || it == 0) {
                finish_tile<kB>(a, tile, total, stash, n_tiles, &s_excl);
            } else {
                if (pend_tile >= 0)
                    finish_tile<kB>(a, pend_tile, pend_total, stash_w + (size_t)((it - 1) & 1) * kB,
                                    n_tiles, &s_excl);
                pend_tile = tile;
                pend_total = total;
                pend_it = it;
            }
        }
    }
    if (kCompact && kDefer && pend_tile >= 0)
        finish_tile<kB>(a, pend_tile, pend_total, stash_w + (size_t)(pend_it & 1) * kB, n_tiles, &s_excl);
    if (kCompact && threadIdx.x == 0) {
        // the last workgroup out re-arms the ticket for the next launch of this
        // stream context (every workgroup leaves exactly once, after its last draw)
        if (atomicAdd(&a.ticket[1], 1u) == gridDim.x - 1) {
            __hip_atomic_store(&a.ticket[0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&a.ticket[1], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

template <int OUT_MODE, int GEN, bool PER_RAY_WVL, int FEAT, bool SMALL = false>
__global__ void __launch_bounds__(block_of(OUT_MODE, FEAT, SMALL), min_waves_of(OUT_MODE, FEAT, SMALL))
ROX_KERNEL_ALIGNED trace_kernel(const TraceArgs a)
{
    trace_tiles<OUT_MODE, GEN, PER_RAY_WVL, FEAT, SMALL>(a);
}

// One launch for several pupil grids of one system -- the (field x wavelength) loops of
// SequentialModel.trace_grid / trace_wavefront (rayoptics/seq/sequential.py:1058-1114) and of
// the figures that call them per field: blockIdx.y picks the item, blockIdx.x strides over
// that item's tiles.  The items sit in device memory and are read through the constant
// address space, i.e. with the same scalar loads that read a kernel argument.
// A batch of up to kInlineItems items travels IN the kernel argument (`inl`, `items` null): no
// upload in front of the kernel -- a copy-engine transfer and its cross-queue signal, ~6 us
// before an 18-30 us kernel of a figure's or BASELINE configs[3]'s few small grids.  (gfx950
// takes kernel arguments of 31 KiB and more; tools/kernarg_probe.hip: a launch with 16 items
// inline 4.6 us against 7.0 us for upload + launch, back to back.)
typedef const __attribute__((address_space(4))) TraceArgs *ConstTraceArgs;
constexpr int kInlineItems = 16;
struct BatchArgs {
    const TraceArgs *items;             // device array [gridDim.y], or nullptr: the items are inl[]
    TraceArgs inl[kInlineItems];
};
template <int OUT_MODE, int FEAT, bool SMALL = false>
__global__ void __launch_bounds__(block_of(OUT_MODE, FEAT, SMALL), min_waves_of(OUT_MODE, FEAT, SMALL))
ROX_KERNEL_ALIGNED trace_kernel_batch(const BatchArgs b)
{
    // (the argument block read in place, with a wave-uniform index: scalar loads either way)
    typedef const __attribute__((address_space(4))) char *cbytes;
    const ConstTraceArgs inl =
        (ConstTraceArgs)((cbytes)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(BatchArgs, inl));
    const ConstTraceArgs items = b.items ? (ConstTraceArgs)b.items : inl;
    trace_tiles<OUT_MODE, GEN_PUPIL, false, FEAT, SMALL>(items[blockIdx.y]);
}

// ------------------------------------------------------------------ launching
struct LaunchCfg {
    int gen;            // GEN_*
    bool per_ray_wvl;
    bool small;         // workgroups of ROX_BLOCK_SMALL threads (pupil launches; see block_of())
    bool fast;          // the tolerance-mode instance (ROX_FAST_FP64 on a reduced-output mode)
    bool gtab;          // the table does not fit the LDS: the general instance over global memory
    int n_inline;       // batches: > 0 = `items` is a HOST array of this many items for the kernel argument
    int out_mode;       // ROX_OUT_*
    dim3 grid;
    size_t lds;
    hipStream_t stream;
};

// Tables beyond ~110 interfaces need more than the 64 KiB of dynamic LDS a kernel may
// use by default (gfx950 has 160 KiB per CU): the limit of the instance is raised once.
constexpr size_t kDefaultDynLds = 64 * 1024;
template <class K>
inline void launch_with_lds(K kernel, const dim3 &grid, const dim3 &block, size_t lds,
                            hipStream_t st, const TraceArgs &a)
{
    if (lds > kDefaultDynLds)
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kernel),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(kernel, grid, block, lds, st, a);
}

// the small-workgroup kernels exist for the modes whose regular workgroup is larger
// (per-ray-wavelength lists keep to the regular one)
template <int OUT_MODE, int GEN, bool PRW, int FEAT>
inline void launch_one(const LaunchCfg &k, const TraceArgs &a)
{
    if constexpr (!PRW && has_small(OUT_MODE, FEAT)) {
        if (k.small) {
            constexpr int bs = block_of(OUT_MODE, FEAT, true);
            auto kern = trace_kernel<OUT_MODE, GEN, PRW, FEAT, true>;
            launch_with_lds(kern, k.grid, dim3(bs), k.lds, k.stream, a);
            return;
        }
    }
    constexpr int bs = block_of(OUT_MODE, FEAT);
    auto kern = trace_kernel<OUT_MODE, GEN, PRW, FEAT, false>;
    launch_with_lds(kern, k.grid, dim3(bs), k.lds, k.stream, a);
}

// (FEAT: the instance's feature set | F_FAST for a tolerance-mode instance | F_GTAB for the
// instance of tables beyond the LDS; whether a mode's kernel reads the table through scalar
// loads is gtab_of()'s per-mode decision)
template <int GEN, bool PRW, int FEAT>
inline void launch_mode(const LaunchCfg &k, const TraceArgs &a)
{
    constexpr bool kF = (FEAT & F_FAST) != 0;
    constexpr int FR = FEAT | gtab_of(FEAT, kF, ROX_OUT_HITS), FF = FEAT | gtab_of(FEAT, kF, ROX_OUT_FULL);
    switch (k.out_mode) {
    case ROX_OUT_FULL: launch_one<ROX_OUT_FULL, GEN, PRW, FF>(k, a); break;
    case ROX_OUT_LAST: launch_one<ROX_OUT_LAST, GEN, PRW, FR>(k, a); break;
    case ROX_OUT_OPD: launch_one<ROX_OUT_OPD, GEN, PRW, FR>(k, a); break;
    case ROX_OUT_HITS_COMPACT: launch_one<ROX_OUT_HITS_COMPACT, GEN, PRW, FR>(k, a); break;
    case ROX_OUT_FAN: launch_one<ROX_OUT_FAN, GEN, PRW, FR>(k, a); break;
    default: launch_one<ROX_OUT_HITS, GEN, PRW, FR>(k, a); break;
    }
}

// all (output mode, ray source) variants of one feature instance
template <int FEAT>
inline void launch_instance(const LaunchCfg &k, const TraceArgs &a)
{
    if (k.gen == GEN_PUPIL)
        launch_mode<GEN_PUPIL, false, FEAT>(k, a);
    else if (k.per_ray_wvl)
        launch_mode<GEN_RAYS, true, FEAT>(k, a);
    else
        launch_mode<GEN_RAYS, false, FEAT>(k, a);
}

// the batched form (trace_kernel_batch): pupil grids, one wavelength per item
// the host-side argument block of a batched launch: one zero-initialised 16 KB block per host
// thread (the launch copies its argument before it returns; inl beyond n_inline is never read by
// the kernel and keeps what an earlier launch left there)
inline BatchArgs &batch_args_block()
{
    static thread_local BatchArgs b{};
    return b;
}

template <class K>
inline void launch_batch_with_lds(K kernel, const dim3 &grid, const dim3 &block, size_t lds,
                                  hipStream_t st, const TraceArgs *items, int n_inline)
{
    if (lds > kDefaultDynLds)
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kernel),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    BatchArgs &b = batch_args_block();
    b.items = n_inline > 0 ? nullptr : items;
    if (n_inline > 0)
        memcpy(b.inl, items, sizeof(TraceArgs) * (size_t)n_inline);
    hipLaunchKernelGGL(kernel, grid, block, lds, st, b);
}

template <int OUT_MODE, int FEAT>
inline void launch_one_batch(const LaunchCfg &k, const TraceArgs *items)
{
    if constexpr (has_small(OUT_MODE, FEAT)) {
        if (k.small) {
            constexpr int bs = block_of(OUT_MODE, FEAT, true);
            auto kern = trace_kernel_batch<OUT_MODE, FEAT, true>;
            launch_batch_with_lds(kern, k.grid, dim3(bs), k.lds, k.stream, items, k.n_inline);
            return;
        }
    }
    constexpr int bs = block_of(OUT_MODE, FEAT);
    auto kern = trace_kernel_batch<OUT_MODE, FEAT, false>;
    launch_batch_with_lds(kern, k.grid, dim3(bs), k.lds, k.stream, items, k.n_inline);
}

template <int FEAT>
inline void launch_instance_batch(const LaunchCfg &k, const TraceArgs *items)
{
    constexpr bool kF = (FEAT & F_FAST) != 0;
    constexpr int FR = FEAT | gtab_of(FEAT, kF, ROX_OUT_HITS), FF = FEAT | gtab_of(FEAT, kF, ROX_OUT_FULL);
    switch (k.out_mode) {
    case ROX_OUT_FULL: launch_one_batch<ROX_OUT_FULL, FF>(k, items); break;
    case ROX_OUT_LAST: launch_one_batch<ROX_OUT_LAST, FR>(k, items); break;
    case ROX_OUT_OPD: launch_one_batch<ROX_OUT_OPD, FR>(k, items); break;
    case ROX_OUT_HITS_COMPACT: launch_one_batch<ROX_OUT_HITS_COMPACT, FR>(k, items); break;
    case ROX_OUT_FAN: launch_one_batch<ROX_OUT_FAN, FR>(k, items); break;
    default: launch_one_batch<ROX_OUT_HITS, FR>(k, items); break;
    }
}

// the feature instances that are compiled (one translation unit each,
// csrc/inst_*.hip); the host launches the first one that covers the need
// (F_EVEN | F_APLIST: what a Zemax import with an EVENASPH surface needs -- every .zmx interface
// carries a clear-aperture list.  BASELINE configs[2]'s file ran on the general instance until
// round 5: 392 wave-VALU per wave-surface against 300 lean, profiles/r05_valu_account.json.)
constexpr int kInstances[] = {0, F_EVEN, F_RADIAL, F_POLY, F_APLIST, F_EVEN | F_APLIST, F_ALL};
void launch_lean(const LaunchCfg &, const TraceArgs &);
void launch_even(const LaunchCfg &, const TraceArgs &);
void launch_radial(const LaunchCfg &, const TraceArgs &);
void launch_poly(const LaunchCfg &, const TraceArgs &);
void launch_aplist(const LaunchCfg &, const TraceArgs &);
void launch_evenap(const LaunchCfg &, const TraceArgs &);
void launch_general(const LaunchCfg &, const TraceArgs &);
void launch_lean_batch(const LaunchCfg &, const TraceArgs *);
void launch_even_batch(const LaunchCfg &, const TraceArgs *);
void launch_radial_batch(const LaunchCfg &, const TraceArgs *);
void launch_poly_batch(const LaunchCfg &, const TraceArgs *);
void launch_aplist_batch(const LaunchCfg &, const TraceArgs *);
void launch_evenap_batch(const LaunchCfg &, const TraceArgs *);
void launch_general_batch(const LaunchCfg &, const TraceArgs *);
// the general instance over a table left in global memory (csrc/gtab_general.hip)
void launch_general_gtab(const LaunchCfg &, const TraceArgs &);
void launch_general_gtab_batch(const LaunchCfg &, const TraceArgs *);
// ... and their tolerance-mode twins (csrc/fast_*.hip: kInstances[i] | F_FAST)
void launch_lean_fast(const LaunchCfg &, const TraceArgs &);
void launch_even_fast(const LaunchCfg &, const TraceArgs &);
void launch_radial_fast(const LaunchCfg &, const TraceArgs &);
void launch_poly_fast(const LaunchCfg &, const TraceArgs &);
void launch_aplist_fast(const LaunchCfg &, const TraceArgs &);
void launch_evenap_fast(const LaunchCfg &, const TraceArgs &);
void launch_general_fast(const LaunchCfg &, const TraceArgs &);
void launch_lean_fast_batch(const LaunchCfg &, const TraceArgs *);
void launch_even_fast_batch(const LaunchCfg &, const TraceArgs *);
void launch_radial_fast_batch(const LaunchCfg &, const TraceArgs *);
void launch_poly_fast_batch(const LaunchCfg &, const TraceArgs *);
void launch_aplist_fast_batch(const LaunchCfg &, const TraceArgs *);
void launch_evenap_fast_batch(const LaunchCfg &, const TraceArgs *);
void launch_general_fast_batch(const LaunchCfg &, const TraceArgs *);

// the pack pass of two-pass packed hits (csrc/pack.hip): a plain ROX_OUT_HITS launch has left
// (x, y)[2][ld] and status[n_rays]; survivors go to dst in ray order, exactly where the fused
// HITS_COMPACT instance would have put them (same tile-state / ticket / base / total protocol)
struct PackArgs {
    const uint8_t *status;
    const double *xy;          // [2][ld]
    int64_t ld, n_rays;
    uint64_t *tile_state;
    uint32_t *ticket;          // [0] next tile, [1] workgroups done
    const int64_t *hits_base_in;
    int64_t *hits_total_out;
    uint32_t epoch;
    double *dst;               // rox_out.seg of the HITS_COMPACT call: (x, y) pairs
    int64_t ld_dst;            // its capacity in pairs (rox_out.ld)
};
constexpr int kPackBlock = 1024, kPackSub = 4, kPackTile = kPackBlock * kPackSub;
void launch_pack(const PackArgs &, unsigned blocks, hipStream_t);

// chief-ray aiming (csrc/rox_search.hpp)
struct AimArgs {
    const double *rows, *n_table, *ph_consts, *wvls;
    const int32_t *slots;
    int32_t n_ifcs, n_wvls, n;
    const rox_aim *probs;      // device
    double eps;
    double *aim_xy;            // device [n][2]
    int32_t *result;           // device [n]
    double *last_xy;           // device [n][2] or nullptr: the last trial ray's (x1, y1)
    int32_t *last_status;      // device [n] or nullptr: ... and its trace status
    int32_t wave_per_problem;  // 1: one wave (block) per problem; 0: one lane per problem
};

// wide-angle pupil search (csrc/rox_search.hpp)
struct EnpArgs {
    const double *rows, *n_table, *ph_consts, *wvls;
    const int32_t *slots;
    int32_t n_ifcs, n_wvls, n;
    const rox_enp *probs;      // device
    double eps;
    double *z_out;             // device [n][2]
    int32_t *result;           // device [n]
};

// vignetting search (csrc/rox_search.hpp)
struct VigArgs {
    const double *rows, *n_table, *ph_consts, *wvls;
    const int32_t *slots;
    int32_t n_ifcs, n_wvls, n;
    const rox_vig *probs;      // device
    double eps;
    double *vig;               // device [n]
    int32_t *clip;             // device [n]
    // rox_iterate_pupil_rays: iterate_pupil_ray on its own (probs unused, vig = start_r)
    const rox_pupil_iter *iters;
    int32_t wave_per_problem;  // 1: one wave (block) per problem; 0: one lane per problem
};

// the feature instances the search kernels are compiled for (csrc/search_*.hip, rox_search.hpp);
// the host launches the first one that covers the system's features
constexpr int kSearchInstances[] = {0, F_EVEN, F_RADIAL, F_APLIST, F_EVEN | F_APLIST, F_ALL};
#define ROX_SEARCH_DECL(name)                                          \
    void launch_aim_##name(const AimArgs &, size_t lds, hipStream_t); \
    void launch_enp_##name(const EnpArgs &, size_t lds, hipStream_t); \
    void launch_vig_##name(const VigArgs &, size_t lds, hipStream_t);
ROX_SEARCH_DECL(lean)
ROX_SEARCH_DECL(even)
ROX_SEARCH_DECL(radial)
ROX_SEARCH_DECL(aplist)
ROX_SEARCH_DECL(evenap)
ROX_SEARCH_DECL(general)
ROX_SEARCH_DECL(general_gtab)
#undef ROX_SEARCH_DECL

}  // namespace rox
