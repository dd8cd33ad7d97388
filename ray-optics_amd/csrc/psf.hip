// rox_calc_psf: analyses.calc_psf (rayoptics/raytr/analyses.py:848-875) on the
// device.
//
// The reference embeds the ndim x ndim OPD grid (waves; NaN = no data) in the
// middle of a maxdim x maxdim zero array W, forms exp(i 2 pi W) with the
// entries that equal 1 (W == 0: padding and missing data) zeroed, and takes
//     AP = |fftshift(fft2(fftshift(phase)))|^2 / max.
// Only the central n x n block of `phase` is non-zero, so the transform is the
// pruned DFT   out = F P F^T   with P the n x n block and F the M x n slice of
// the (shifted) DFT matrix: two complex GEMMs of shapes (M x n)(n x n) and
// (M x n)(n x M).  That is GEMM-shaped fp64 work, so it runs on the matrix
// cores (v_mfma_f64_16x16x4_f64), for any M -- no power-of-two restriction --
// and costs 8 M n (n + M) flop instead of a full M x M FFT's passes over HBM.
//
//   psf_prepare   P^T (phase of the block, transposed) and F (twiddles: once per shape)
//   cgemm_nt<0>   T = F P            C[i][j] = sum_k A[i][k] B[j][k], complex
//   cgemm_nt<1>   AP = |T F^T|^2 and its maximum (epilogue)
//   psf_scale     AP / max
// Up to maxdim 512 (the sizes figures use) the GEMMs take 32 x 32 workgroup tiles; the padded
// planes are zeroed and F is formed once per shape, not per call.
//
// Matrices are kept as separate real / imaginary planes, row-major with the
// reduction index contiguous and padded with zeros to a multiple of 16 (rows
// to a multiple of 64), so the GEMM loads need no guards.
#include <hip/hip_runtime.h>

#include <cfloat>
#include <cstdint>
#include <cstdio>
#include <mutex>
#include <new>
#include <vector>

#include "../../include/roxtrace.h"

namespace rox {
int host_fail(int code, const char *msg);     // roxtrace.hip: sets rox_last_error()
}

namespace {

typedef double d4 __attribute__((ext_vector_type(4)));

constexpr int kTile = 64;           // workgroup tile (4 waves, 32 x 32 each)
constexpr int kKBlock = 16;         // reduction step of the main loop

__host__ __device__ inline int64_t round_up(int64_t v, int64_t m) { return (v + m - 1) / m * m; }

// P^T and F.  n = ndim, M = maxdim, kp = padded n.
//   block origin o = M/2 - (n/2 - 1)                       (analyses.py:861-863)
//   fftshift = roll by h = M/2 on input and output          (numpy.fft.fftshift)
//   F[u][a]  = exp(-2 pi i ((u - h) mod M) ((o + a + h) mod M) / M)
// exp(i 2 pi W) of one OPD entry with the entries that equal 1 zeroed (analyses.py:864-870)
__device__ __forceinline__ void pupil_phase(double w, double &c, double &s)
{
    if (w != w)                                     // np.nan_to_num
        w = 0.0;
    else if (w == __builtin_inf())
        w = DBL_MAX;
    else if (w == -__builtin_inf())
        w = -DBL_MAX;
    // 1j*2*np.pi*W: the imaginary part is the single product 2 pi W
    const double x = 6.283185307179586 * w;
    sincos(x, &s, &c);
    if (c == 1.0 && s == 0.0)                       // phase[i][j] == 1 -> 0 (:867-870)
        c = s = 0.0;
}

// P^T (per call) and, when `twiddles`, F (per shape: it depends on (n, M) only)
__global__ void psf_prepare(const double *opd, int n, int M, int kp, double *ptr, double *pti,
                            double *fr, double *fi, unsigned long long *maxbits, int phases, int twiddles)
{
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx == 0 && phases)
        *maxbits = 0;               // this call's maximum starts from zero (the second GEMM folds into it)
    if (phases && idx < (int64_t)n * n) {
        const int a = (int)(idx / n), b = (int)(idx % n);
        double s, c;
        pupil_phase(opd[idx], c, s);
        ptr[(int64_t)b * kp + a] = c;
        pti[(int64_t)b * kp + a] = s;
    }
    if (twiddles && idx < (int64_t)M * n) {
        const int u = (int)(idx / n), a = (int)(idx % n);
        const int h = M / 2, o = M / 2 - (n / 2 - 1);
        const int64_t uu = (u - h + M) % M, ii = (o + a + h) % M;
        const int64_t e = (uu * ii) % M;
        double s, c;
        sincospi(2.0 * (double)e / (double)M, &s, &c);
        fr[(int64_t)u * kp + a] = c;
        fi[(int64_t)u * kp + a] = -s;
    }
}

template <int EPI, int TM>      // TM x TM MFMA tiles per wave: 2 (32 x 32, large problems) or 1 (figure sizes)
__global__ __launch_bounds__(256) void cgemm_nt(const double *__restrict__ ar_, const double *__restrict__ ai_,
                                                const double *__restrict__ br_, const double *__restrict__ bi_,
                                                int kp, int I, int J, double *c0, double *c1, int ldc,
                                                unsigned long long *maxbits)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int r = lane & 15, g = lane >> 4;
    constexpr int WT = 16 * TM;             // rows / columns of the wave's tile; the block's is 2 WT
    const int i0 = blockIdx.y * (2 * WT) + (wave >> 1) * WT;
    const int j0 = blockIdx.x * (2 * WT) + (wave & 1) * WT;
    d4 cr[TM][TM], ci[TM][TM];
    for (int a = 0; a < TM; ++a)
        for (int b = 0; b < TM; ++b)
            cr[a][b] = ci[a][b] = d4{0., 0., 0., 0.};
    const double *pa_r = ar_ + (size_t)(i0 + r) * kp + 4 * g;
    const double *pa_i = ai_ + (size_t)(i0 + r) * kp + 4 * g;
    const double *pb_r = br_ + (size_t)(j0 + r) * kp + 4 * g;
    const double *pb_i = bi_ + (size_t)(j0 + r) * kp + 4 * g;
    const size_t t16 = (size_t)16 * kp;
    // software pipeline: the operands of k block i + 1 are in flight while the 64 MFMAs of
    // block i issue (a lone wave per SIMD has nobody else to hide the load latency behind)
    d4 xr[TM], xi[TM], yr[TM], yi[TM];
    for (int t = 0; t < TM; ++t) {
        xr[t] = *(const d4 *)(pa_r + t * t16);
        xi[t] = *(const d4 *)(pa_i + t * t16);
        yr[t] = *(const d4 *)(pb_r + t * t16);
        yi[t] = *(const d4 *)(pb_i + t * t16);
    }
    for (int k0 = 0; k0 < kp; k0 += kKBlock) {
        d4 nxr[TM], nxi[TM], nyr[TM], nyi[TM], xn[TM];
        const int kn = (k0 + kKBlock < kp) ? k0 + kKBlock : k0;     // last block: a harmless reload
        for (int t = 0; t < TM; ++t) {
            nxr[t] = *(const d4 *)(pa_r + t * t16 + kn);
            nxi[t] = *(const d4 *)(pa_i + t * t16 + kn);
            nyr[t] = *(const d4 *)(pb_r + t * t16 + kn);
            nyi[t] = *(const d4 *)(pb_i + t * t16 + kn);
            xn[t] = -xi[t];
        }
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int b = 0; b < TM; ++b) {
                    cr[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(xr[a][e], yr[b][e], cr[a][b], 0, 0, 0);
                    cr[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(xn[a][e], yi[b][e], cr[a][b], 0, 0, 0);
                    ci[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(xr[a][e], yi[b][e], ci[a][b], 0, 0, 0);
                    ci[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(xi[a][e], yr[b][e], ci[a][b], 0, 0, 0);
                }
        for (int t = 0; t < TM; ++t) {
            xr[t] = nxr[t]; xi[t] = nxi[t]; yr[t] = nyr[t]; yi[t] = nyi[t];
        }
    }
    double vmax = 0.0;
    for (int a = 0; a < TM; ++a)
        for (int b = 0; b < TM; ++b)
            for (int q = 0; q < 4; ++q) {
                const int row = i0 + 16 * a + g + 4 * q, col = j0 + 16 * b + r;
                if (row >= I || col >= J)
                    continue;
                const size_t at = (size_t)row * ldc + col;
                if (EPI == 0) {
                    c0[at] = cr[a][b][q];
                    c1[at] = ci[a][b][q];
                } else {
                    const double m = hypot(cr[a][b][q], ci[a][b][q]);     // abs(z)
                    const double v = m * m;                               // ... ** 2
                    c0[at] = v;
                    vmax = fmax(vmax, v);                                 // nanmax
                }
            }
    if (EPI == 1) {
        for (int off = 32; off; off >>= 1)
            vmax = fmax(vmax, __shfl_xor(vmax, off));
        if (lane == 0)
            atomicMax(maxbits, (unsigned long long)__double_as_longlong(vmax));
    }
}

__global__ void psf_scale(double *ap, int64_t count, const unsigned long long *maxbits)
{
    const double m = __longlong_as_double((long long)*maxbits);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count;
         i += (int64_t)gridDim.x * blockDim.x)
        ap[i] = ap[i] / m;                                                // AP / AP_max
}

// grow-only workspace per (device, stream)
struct Workspace {
    int device = 0;
    hipStream_t stream = nullptr;
    char *buf = nullptr;
    size_t cap = 0;
    int64_t zeroed_for[3] = {0, 0, 0};      // (n, M, bytes) the padded planes were last zeroed for
    std::mutex mu;          // one call at a time builds / enqueues on this workspace
};
std::mutex g_mu;
std::vector<Workspace *> g_ws;

Workspace *workspace_for(int device, hipStream_t st)
{
    std::lock_guard<std::mutex> lock(g_mu);
    for (Workspace *w : g_ws)
        if (w->device == device && w->stream == st)
            return w;
    Workspace *w = new (std::nothrow) Workspace;
    if (w) {
        w->device = device;
        w->stream = st;
        g_ws.push_back(w);
    }
    return w;
}

int hip_fail(const char *what, hipError_t e)
{
    char msg[256];
    snprintf(msg, sizeof msg, "rox_calc_psf: %s: %s", what, hipGetErrorString(e));
    return rox::host_fail(ROX_E_HIP, msg);
}

#define PSF_TRY(expr)                                   \
    do {                                                \
        hipError_t e_ = (expr);                         \
        if (e_ != hipSuccess)                           \
            return hip_fail(#expr, e_);                 \
    } while (0)

}  // namespace

extern "C" int rox_calc_psf(const double *opd, int32_t ndim, int32_t maxdim, double *psf,
                            uint32_t flags, void *stream)
{
    if (!opd || !psf)
        return rox::host_fail(ROX_E_ARG, "rox_calc_psf: null argument");
    if (flags & ~(uint32_t)ROX_HOST_POINTERS)
        return rox::host_fail(ROX_E_ARG, "rox_calc_psf: unknown flag");
    // the reference's slice assignment (analyses.py:861-863) only has a matching shape
    // for an even ndim whose block fits inside the maxdim array
    if (ndim < 2 || (ndim & 1))
        return rox::host_fail(ROX_E_ARG, "rox_calc_psf: ndim must be even and >= 2");
    const int n = ndim, M = maxdim;
    const int o = M / 2 - (n / 2 - 1);
    if (M < 2 || o < 0 || o + n > M)
        return rox::host_fail(ROX_E_ARG, "rox_calc_psf: the ndim block does not fit in maxdim");
    if (M > 32768)
        return rox::host_fail(ROX_E_ARG, "rox_calc_psf: maxdim > 32768");
    hipStream_t st = (hipStream_t)stream;
    int device = 0;
    PSF_TRY(hipGetDevice(&device));
    Workspace *ws = workspace_for(device, st);
    if (!ws)
        return rox::host_fail(ROX_E_NOMEM, "rox_calc_psf: out of host memory");
    // calls on one stream share the workspace: enqueue them whole, one after the other
    // (stream order then keeps their kernels apart); host-pointer calls hold it until done
    std::lock_guard<std::mutex> turn(ws->mu);

    const bool host = (flags & ROX_HOST_POINTERS) != 0;
    const int64_t kp = round_up(n, kKBlock), mp = round_up(M, kTile), np_ = round_up(n, kTile);
    const size_t pl_f = sizeof(double) * (size_t)mp * kp, pl_p = sizeof(double) * (size_t)np_ * kp;
    const size_t b_opd = host ? sizeof(double) * (size_t)n * n : 0;
    const size_t b_psf = host ? sizeof(double) * (size_t)M * M : 0;
    const size_t zeroed = 4 * pl_f + 2 * pl_p + 64;         // F, T, P^T planes + the maximum
    const size_t total = zeroed + b_opd + b_psf + 64;
    if (ws->cap < total) {
        if (ws->buf)
            PSF_TRY(hipFree(ws->buf));
        ws->buf = nullptr;
        ws->cap = 0;
        PSF_TRY(hipMalloc((void **)&ws->buf, total));
        ws->cap = total;
        ws->zeroed_for[0] = ws->zeroed_for[1] = ws->zeroed_for[2] = 0;
    }
    char *p = ws->buf;
    double *fr = (double *)p;               p += pl_f;
    double *fi = (double *)p;               p += pl_f;
    double *tr = (double *)p;               p += pl_f;
    double *ti = (double *)p;               p += pl_f;
    double *ptr = (double *)p;              p += pl_p;
    double *pti = (double *)p;              p += pl_p;
    unsigned long long *maxbits = (unsigned long long *)p;  p += 64;
    double *d_opd = (double *)p;            p += (b_opd + 63) & ~size_t(63);
    double *d_psf = (double *)p;
    // The planes are zero-padded to the GEMM tiles.  Every element inside the (n, M) shape is
    // rewritten by each call and the padding is never written, so the planes need zeroing only
    // when the workspace is new or the shape changes; the twiddles F depend on the shape alone
    // and are formed then too (at figure sizes a call is little but stream operations and
    // sincospi).  Forming the phases inside the first product instead -- three launches --
    // measured slower from (64, 256) up: every block row repeats the n^2 sincos.
    const bool new_shape = ws->zeroed_for[0] != n || ws->zeroed_for[1] != M ||
                           ws->zeroed_for[2] != (int64_t)zeroed;
    if (new_shape) {
        PSF_TRY(hipMemsetAsync(ws->buf, 0, zeroed, st));
        ws->zeroed_for[0] = n; ws->zeroed_for[1] = M; ws->zeroed_for[2] = (int64_t)zeroed;
    }
    const double *src = opd;
    double *dst = psf;
    if (host) {
        PSF_TRY(hipMemcpyAsync(d_opd, opd, b_opd, hipMemcpyHostToDevice, st));
        src = d_opd;
        dst = d_psf;
    }
    const int64_t work = (int64_t)M * n;        // >= n * n
    const bool small = M <= 512;
    hipLaunchKernelGGL(psf_prepare, dim3((unsigned)(((new_shape ? work : (int64_t)n * n) + 255) / 256)), dim3(256), 0,
                       st, src, n, M, (int)kp, ptr, pti, fr, fi, maxbits, 1, new_shape ? 1 : 0);
    // T[u][b] = sum_a F[u][a] P[a][b];  AP[u][v] = |sum_b T[u][b] F[v][b]|^2.
    // 64 x 64 workgroup tiles give a (64, 256) problem 4 and 16 workgroups on 256 CUs: up to
    // maxdim 512 the 32 x 32 instance is used (one MFMA tile per wave, four times the workgroups)
    if (small) {
        hipLaunchKernelGGL((cgemm_nt<0, 1>), dim3((unsigned)(np_ / 32), (unsigned)(mp / 32)), dim3(256), 0, st,
                           fr, fi, ptr, pti, (int)kp, M, n, tr, ti, (int)kp, maxbits);
        hipLaunchKernelGGL((cgemm_nt<1, 1>), dim3((unsigned)(mp / 32), (unsigned)(mp / 32)), dim3(256), 0, st,
                           tr, ti, fr, fi, (int)kp, M, M, dst, (double *)nullptr, M, maxbits);
    } else {
        hipLaunchKernelGGL((cgemm_nt<0, 2>), dim3((unsigned)(np_ / kTile), (unsigned)(mp / kTile)), dim3(256), 0, st,
                           fr, fi, ptr, pti, (int)kp, M, n, tr, ti, (int)kp, maxbits);
        hipLaunchKernelGGL((cgemm_nt<1, 2>), dim3((unsigned)(mp / kTile), (unsigned)(mp / kTile)), dim3(256), 0, st,
                           tr, ti, fr, fi, (int)kp, M, M, dst, (double *)nullptr, M, maxbits);
    }
    hipLaunchKernelGGL(psf_scale, dim3(1024), dim3(256), 0, st, dst, (int64_t)M * M, maxbits);
    PSF_TRY(hipGetLastError());
    if (host) {
        PSF_TRY(hipMemcpyAsync(psf, d_psf, b_psf, hipMemcpyDeviceToHost, st));
        PSF_TRY(hipStreamSynchronize(st));
    }
    return 0;
}
