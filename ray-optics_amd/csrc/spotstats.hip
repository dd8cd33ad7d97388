// spotstats.hip -- rox_spot_stats: what a spot diagram's consumers reduce the image-plane hits
// to, computed where the hits are (HBM) instead of on 16 B per ray pulled over PCIe:
//   * RayGeoPSF.ray_data_bounds (rayoptics/mpl/analysisfigure.py:237-248): min / max of x and y;
//   * RayGeoPSF.plot's `ax.hist2d(x, y, bins=[x_edges, y_edges])` (:250-290) = numpy.histogram2d
//     with explicit edges: bin i holds edges[i] <= v < edges[i + 1], the last bin its right edge
//     too, values outside the edges are dropped;
//   * the centroid and RMS spot radius merit functions are built from (sum x, sum y, sum x^2, sum y^2).
// Input: the ROX_OUT_HITS rows of a launch ((x, y)[2][ld] + status: no pack pass needed) or
// packed pairs (ROX_OUT_HITS_COMPACT).  Memory-bound: 17 B read per ray.
//
// Histogram: a wave first merges the lanes that fall into the same bin (a spot diagram puts
// most rays into a handful of bins: un-merged, the atomics of a launch queue on a few
// addresses), then adds to one of kCopies private copies of the histogram (the copy of its
// workgroup's slot), which the finishing pass sums into the caller's array -- device memory or
// device-mapped pinned host memory.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstring>
#include <mutex>

#include "rox_device.hpp"

namespace rox {
int host_fail(int code, const char *msg);

namespace {

constexpr int kStatBlock = 256;
constexpr int kCopies = 32;

struct StatArgs {
    const double *x, *y;        // rows: x[r], y[r];  pairs: x = base, stride 2 (y = x + 1)
    int64_t stride;             // 1 (rows) or 2 (pairs)
    const uint8_t *status;      // rows only; nullptr = every entry counts
    const int64_t *n_dev;       // pairs: the count on the device (rox_out.n_hits), or nullptr
    int64_t n;                  // entries (rays of the launch; upper bound when n_dev is set)
    const double *xe, *ye;      // histogram edges (device), nx + 1 / ny + 1 values; or nullptr
    int32_t nx, ny;
    uint32_t *copies;           // [kCopies][nx * ny]
    double *partial;            // [blocks][10]: n, sx, sy, sxx, syy, minx, maxx, miny, maxy, (pad)
};

// bin of v among edges e[0..n] as numpy.histogram with explicit edges decides it:
// searchsorted(e, v, 'right') - 1, the right-most edge belonging to the last bin; -1 = outside
__device__ __forceinline__ int bin_of(const double *e, int n, double v)
{
    if (!(v >= e[0]) || !(v <= e[n]))
        return -1;                              // (NaN included)
    int lo = 0, hi = n;                         // invariant: e[lo] <= v, (hi == n or v < e[hi])
    // uniform edges (linspace): the proportional guess is right or one off -- two probes
    const double w = e[n] - e[0];
    const double t = w > 0.0 ? (v - e[0]) / w * n : 0.0;
    int g = t < (double)(n - 1) ? (int)t : n - 1;
    if (e[g] <= v) {
        lo = g;
        if (g + 1 <= n && (g + 1 == n || v < e[g + 1]))
            return g;
    } else {
        hi = g;
    }
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (e[mid] <= v)
            lo = mid;
        else
            hi = mid;
    }
    return lo;
}

__global__ void __launch_bounds__(kStatBlock) spot_stats_kernel(const StatArgs a)
{
    const int64_t n = a.n_dev ? (*a.n_dev < 0 ? -*a.n_dev : *a.n_dev) : a.n;
    const int64_t n_eff = n < a.n ? n : a.n;
    double cnt = 0, sx = 0, sy = 0, sxx = 0, syy = 0;
    double mnx = __builtin_inf(), mxx = -__builtin_inf(), mny = __builtin_inf(), mxy = -__builtin_inf();
    uint32_t *copy = a.copies ? a.copies + (size_t)(blockIdx.x % kCopies) * ((size_t)a.nx * a.ny) : nullptr;
    const int lane = threadIdx.x & 63;
    for (int64_t r0 = (int64_t)blockIdx.x * kStatBlock; r0 < n_eff; r0 += (int64_t)gridDim.x * kStatBlock) {
        const int64_t r = r0 + threadIdx.x;
        bool ok = r < n_eff && (!a.status || a.status[r] == ROX_OK);
        double x = 0, y = 0;
        if (ok) {
            x = a.x[r * a.stride];
            y = a.y[r * a.stride];
            cnt += 1.0;
            sx += x; sy += y;
            sxx = fma(x, x, sxx); syy = fma(y, y, syy);
            mnx = fmin(mnx, x); mxx = fmax(mxx, x);
            mny = fmin(mny, y); mxy = fmax(mxy, y);
        }
        if (copy) {
            int b = -1;
            if (ok) {
                const int bx = bin_of(a.xe, a.nx, x), by = bin_of(a.ye, a.ny, y);
                if (bx >= 0 && by >= 0)
                    b = bx * a.ny + by;
            }
            // merge equal bins within the wave: the lowest lane of each group adds the group's size
            uint64_t todo = __ballot(b >= 0);
            while (todo) {
                const int lead = __builtin_ctzll(todo);
                const int bl = __shfl(b, lead);
                const uint64_t same = __ballot(b == bl) & todo;
                if (lane == lead)
                    atomicAdd(&copy[bl], (uint32_t)__popcll(same));
                todo &= ~same;
            }
        }
    }
    // workgroup reduction of the moments: wave shuffles, then LDS
    auto wsum = [](double v) { for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o); return v; };
    auto wmin = [](double v) { for (int o = 32; o > 0; o >>= 1) v = fmin(v, __shfl_xor(v, o)); return v; };
    auto wmax = [](double v) { for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o)); return v; };
    cnt = wsum(cnt); sx = wsum(sx); sy = wsum(sy); sxx = wsum(sxx); syy = wsum(syy);
    mnx = wmin(mnx); mxx = wmax(mxx); mny = wmin(mny); mxy = wmax(mxy);
    __shared__ double s[kStatBlock / 64][9];
    const int wave = threadIdx.x >> 6;
    if (lane == 0) {
        s[wave][0] = cnt; s[wave][1] = sx; s[wave][2] = sy; s[wave][3] = sxx; s[wave][4] = syy;
        s[wave][5] = mnx; s[wave][6] = mxx; s[wave][7] = mny; s[wave][8] = mxy;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double *p = a.partial + (size_t)blockIdx.x * 10;
        for (int k = 0; k < 9; ++k) {
            double v = s[0][k];
            for (int w = 1; w < kStatBlock / 64; ++w)
                v = (k < 5) ? v + s[w][k] : ((k == 5 || k == 7) ? fmin(v, s[w][k]) : fmax(v, s[w][k]));
            p[k] = v;
        }
    }
}

// partial[blocks][10] -> summary[9] (one wave); copies[kCopies][bins] -> hist[bins]
__global__ void __launch_bounds__(kStatBlock) spot_finish_kernel(const double *partial, int blocks,
                                                                 double *summary, const uint32_t *copies,
                                                                 int64_t bins, uint32_t *hist)
{
    if (blockIdx.x == 0 && threadIdx.x < 64) {
        double v[9] = {0, 0, 0, 0, 0, __builtin_inf(), -__builtin_inf(), __builtin_inf(), -__builtin_inf()};
        for (int b = threadIdx.x; b < blocks; b += 64)
            for (int k = 0; k < 9; ++k) {
                const double p = partial[(size_t)b * 10 + k];
                v[k] = (k < 5) ? v[k] + p : ((k == 5 || k == 7) ? fmin(v[k], p) : fmax(v[k], p));
            }
        for (int k = 0; k < 9; ++k) {
            for (int o = 32; o > 0; o >>= 1) {
                const double q = __shfl_xor(v[k], o);
                v[k] = (k < 5) ? v[k] + q : ((k == 5 || k == 7) ? fmin(v[k], q) : fmax(v[k], q));
            }
            if (threadIdx.x == 0)
                summary[k] = v[k];
        }
    }
    if (hist)
        for (int64_t i = (int64_t)blockIdx.x * kStatBlock + threadIdx.x; i < bins;
             i += (int64_t)gridDim.x * kStatBlock) {
            uint32_t t = 0;
            for (int c = 0; c < kCopies; ++c)
                t += copies[(size_t)c * bins + i];
            hist[i] = t;
        }
}

// per-process scratch (grow-only), one call at a time
struct StatScratch {
    std::mutex mu;
    char *d = nullptr;          // device: edges, copies, partials
    size_t d_cap = 0;
    char *h = nullptr;          // device-mapped pinned: summary, histogram for host destinations
    size_t h_cap = 0;
} g_scratch;

}  // namespace
}  // namespace rox

using namespace rox;

extern "C" int rox_spot_stats(const double *seg, int64_t ld, const uint8_t *status, const int64_t *n_hits,
                              int64_t n, int32_t layout, const double *x_edges, int32_t n_x_edges,
                              const double *y_edges, int32_t n_y_edges, rox_spot_summary *summary,
                              uint32_t *hist, void *stream)
{
    if (!seg || !summary || n < 0 || (layout != ROX_SPOT_ROWS && layout != ROX_SPOT_PAIRS))
        return host_fail(ROX_E_ARG, "rox_spot_stats: bad argument");
    if (layout == ROX_SPOT_ROWS && ld < n)
        return host_fail(ROX_E_ARG, "rox_spot_stats: ld < n");
    const bool want_hist = hist != nullptr;
    if (want_hist && (!x_edges || !y_edges || n_x_edges < 2 || n_y_edges < 2 ||
                      (int64_t)(n_x_edges - 1) * (n_y_edges - 1) > (int64_t(1) << 26)))
        return host_fail(ROX_E_ARG, "rox_spot_stats: a histogram needs >= 2 edges per axis (and <= 2^26 bins)");
    if (want_hist)
        for (int a = 0; a < 2; ++a) {           // numpy: bins must increase monotonically
            const double *e = a ? y_edges : x_edges;
            const int m = a ? n_y_edges : n_x_edges;
            for (int i = 1; i < m; ++i)
                if (!(e[i] >= e[i - 1]))
                    return host_fail(ROX_E_ARG, "rox_spot_stats: edges must increase monotonically");
        }
    hipStream_t st = (hipStream_t)stream;
    const int nx = want_hist ? n_x_edges - 1 : 0, ny = want_hist ? n_y_edges - 1 : 0;
    const int64_t bins = (int64_t)nx * ny;
    int blocks = (int)((n + kStatBlock - 1) / kStatBlock);
    blocks = blocks < 1 ? 1 : (blocks > 2048 ? 2048 : blocks);
    const size_t b_edges = ((size_t)(n_x_edges + n_y_edges) * 8 + 255) & ~size_t(255);
    const size_t b_copies = ((size_t)kCopies * bins * 4 + 255) & ~size_t(255);
    const size_t b_part = (size_t)blocks * 10 * 8;
    const size_t d_need = (want_hist ? b_edges + b_copies : 0) + b_part;
    const size_t h_hist = (((size_t)bins * 4) + 255) & ~size_t(255);
    const size_t h_need = 256 + h_hist + (want_hist ? b_edges : 0);
    std::lock_guard<std::mutex> lock(g_scratch.mu);
#define SPOT_TRY(expr)                                                                   \
    do {                                                                                 \
        hipError_t e_ = (expr);                                                          \
        if (e_ != hipSuccess)                                                            \
            return host_fail(ROX_E_HIP, hipGetErrorString(e_));                          \
    } while (0)
    if (g_scratch.d_cap < d_need) {
        if (g_scratch.d)
            SPOT_TRY(hipFree(g_scratch.d));
        g_scratch.d = nullptr;
        g_scratch.d_cap = 0;
        SPOT_TRY(hipMalloc((void **)&g_scratch.d, d_need + d_need / 4));
        g_scratch.d_cap = d_need + d_need / 4;
    }
    if (g_scratch.h_cap < h_need) {
        if (g_scratch.h)
            SPOT_TRY(hipHostFree(g_scratch.h));
        g_scratch.h = nullptr;
        g_scratch.h_cap = 0;
        SPOT_TRY(hipHostMalloc((void **)&g_scratch.h, h_need + h_need / 4, hipHostMallocMapped));
        g_scratch.h_cap = h_need + h_need / 4;
    }
    char *d = g_scratch.d;
    StatArgs a{};
    a.n = n;
    a.n_dev = n_hits;
    if (layout == ROX_SPOT_ROWS) {
        a.x = seg; a.y = seg + ld; a.stride = 1; a.status = status;
    } else {
        a.x = seg; a.y = seg + 1; a.stride = 2; a.status = nullptr;
    }
    if (want_hist) {
        // the edges: into the pinned block, one stream-ordered DMA from there (a copy from the
        // caller's pageable arrays would be staged by the runtime, synchronously)
        double *xe = (double *)d;
        double *ye = xe + n_x_edges;
        double *he = (double *)(g_scratch.h + 256 + h_hist);
        memcpy(he, x_edges, sizeof(double) * n_x_edges);
        memcpy(he + n_x_edges, y_edges, sizeof(double) * n_y_edges);
        SPOT_TRY(hipMemcpyAsync(xe, he, sizeof(double) * (n_x_edges + n_y_edges), hipMemcpyHostToDevice, st));
        a.xe = xe; a.ye = ye; a.nx = nx; a.ny = ny;
        a.copies = (uint32_t *)(d + b_edges);
        SPOT_TRY(hipMemsetAsync(a.copies, 0, (size_t)kCopies * bins * 4, st));
        d += b_edges + b_copies;
    }
    a.partial = (double *)d;
    double *summ = (double *)g_scratch.h;               // device-mapped: the finishing pass writes it
    uint32_t *hist_h = (uint32_t *)(g_scratch.h + 256);
    hipLaunchKernelGGL(spot_stats_kernel, dim3(blocks), dim3(kStatBlock), 0, st, a);
    const int fblocks = want_hist ? (int)((bins + kStatBlock - 1) / kStatBlock > 1024 ? 1024
                                                                                     : (bins + kStatBlock - 1) / kStatBlock)
                                  : 1;
    hipLaunchKernelGGL(spot_finish_kernel, dim3(fblocks < 1 ? 1 : fblocks), dim3(kStatBlock), 0, st, a.partial,
                       blocks, summ, a.copies, bins, want_hist ? hist_h : nullptr);
    SPOT_TRY(hipGetLastError());
    SPOT_TRY(hipStreamSynchronize(st));
    summary->n = (int64_t)summ[0];
    summary->sum[0] = summ[1]; summary->sum[1] = summ[2];
    summary->sum_sq[0] = summ[3]; summary->sum_sq[1] = summ[4];
    summary->min[0] = summ[5]; summary->max[0] = summ[6];
    summary->min[1] = summ[7]; summary->max[1] = summ[8];
    if (want_hist)
        memcpy(hist, hist_h, (size_t)bins * 4);
#undef SPOT_TRY
    return 0;
}
