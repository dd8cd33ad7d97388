// pcie_store.hip -- how fast can a kernel write a compacted spot diagram
// (13 MB of 16-byte (x, y) pairs) straight into pinned host memory, compared
// with the copy engine?  Variables: coherent (hipHostMallocDefault, what
// torch's pin_memory gives) vs non-coherent host memory, aligned vs unaligned
// wave stores, non-temporal vs plain stores, 78 % lane occupancy (compaction).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef double d2 __attribute__((ext_vector_type(2)));

// every lane writes one pair; wave w writes pairs [w*64 + shift, ...)
template <bool NT>
__global__ void __launch_bounds__(512) dense(d2 *out, long n, long shift)
{
    for (long i = (long)blockIdx.x * 512 + threadIdx.x; i < n; i += (long)gridDim.x * 512) {
        d2 v; v.x = (double)i; v.y = 1.0;
        if (NT) __builtin_nontemporal_store(v, out + i + shift); else out[i + shift] = v;
    }
}

// compaction-like: ~78 % of the lanes write, packed by ballot rank, tile bases
// precomputed (no look-back): the store pattern of HITS_COMPACT
__global__ void __launch_bounds__(512) packed(d2 *out, long n, const int *tile_base)
{
    __shared__ int wc[8];
    for (long t = blockIdx.x; t * 512 < n; t += gridDim.x) {
        const long i = t * 512 + threadIdx.x;
        const bool ok = ((i * 2654435761u) >> 7) % 100 < 78;
        const unsigned long long m = __ballot(ok);
        const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
        if (lane == 0) wc[w] = __popcll(m);
        __syncthreads();
        int off = 0;
        for (int k = 0; k < w; ++k) off += wc[k];
        __syncthreads();
        if (ok) {
            d2 v; v.x = (double)i; v.y = 1.0;
            __builtin_nontemporal_store(v, out + tile_base[t] + off + __popcll(m & ((1ull << lane) - 1)));
        }
    }
}

template <class F>
double time_us(F f, int reps)
{
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    f(); f();
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(a));
    for (int i = 0; i < reps; ++i) f();
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms; CHECK(hipEventElapsedTime(&ms, a, b));
    return ms * 1e3 / reps;
}

int main()
{
    const long n = 821936;                  // pairs of the bench spot diagram
    const size_t bytes = (size_t)n * 16;
    const long n_in = 1048576;
    d2 *dev;
    CHECK(hipMalloc(&dev, bytes + 4096));
    // tile bases for `packed`
    int *h_base = (int *)malloc(sizeof(int) * 2048), acc = 0;
    for (long t = 0; t < 2048; ++t) {
        h_base[t] = acc;
        for (long i = t * 512; i < t * 512 + 512; ++i)
            acc += ((i * 2654435761u) >> 7) % 100 < 78;
    }
    int *d_base; CHECK(hipMalloc(&d_base, sizeof(int) * 2048));
    CHECK(hipMemcpy(d_base, h_base, sizeof(int) * 2048, hipMemcpyHostToDevice));
    const size_t pbytes = (size_t)acc * 16;
    for (int mode = 0; mode < 3; ++mode) {
        unsigned flags = mode == 0 ? hipHostMallocDefault : mode == 1 ? hipHostMallocNonCoherent
                                                                    : (hipHostMallocNonCoherent | hipHostMallocWriteCombined);
        const char *nm = mode == 0 ? "coherent" : mode == 1 ? "noncoherent" : "noncoherent_wc";
        d2 *host;
        CHECK(hipHostMalloc((void **)&host, bytes + 65536, flags));
        for (long shift : {0L, 3L}) {
            double t = time_us([&] { hipLaunchKernelGGL(dense<true>, dim3(2048), dim3(512), 0, 0, host, n, shift); }, 10);
            printf("{\"mem\": \"%s\", \"kernel\": \"dense_nt\", \"shift\": %ld, \"us\": %.1f, \"GBps\": %.1f}\n", nm, shift, t, bytes / t / 1e3);
            t = time_us([&] { hipLaunchKernelGGL(dense<false>, dim3(2048), dim3(512), 0, 0, host, n, shift); }, 10);
            printf("{\"mem\": \"%s\", \"kernel\": \"dense_plain\", \"shift\": %ld, \"us\": %.1f, \"GBps\": %.1f}\n", nm, shift, t, bytes / t / 1e3);
        }
        for (int grid : {2048, 512, 256}) {
            double t = time_us([&] { hipLaunchKernelGGL(packed, dim3(grid), dim3(512), 0, 0, host, n_in, d_base); }, 10);
            printf("{\"mem\": \"%s\", \"kernel\": \"packed_nt\", \"grid\": %d, \"us\": %.1f, \"GBps\": %.1f}\n", nm, grid, t, pbytes / t / 1e3);
        }
        double t = time_us([&] { CHECK(hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, 0)); }, 10);
        printf("{\"mem\": \"%s\", \"kernel\": \"hipMemcpyAsync_D2H\", \"us\": %.1f, \"GBps\": %.1f}\n", nm, t, bytes / t / 1e3);
        t = time_us([&] { for (int k = 0; k < 4; ++k) CHECK(hipMemcpyAsync((char *)host + k * (bytes / 4), (char *)dev + k * (bytes / 4), bytes / 4, hipMemcpyDeviceToHost, 0)); }, 10);
        printf("{\"mem\": \"%s\", \"kernel\": \"hipMemcpyAsync_D2H_x4\", \"us\": %.1f, \"GBps\": %.1f}\n", nm, t, bytes / t / 1e3);
        CHECK(hipHostFree(host));
    }
    // same kernels into HBM for scale
    double t = time_us([&] { hipLaunchKernelGGL(packed, dim3(2048), dim3(512), 0, 0, dev, n_in, d_base); }, 10);
    printf("{\"mem\": \"hbm\", \"kernel\": \"packed_nt\", \"us\": %.1f, \"GBps\": %.1f}\n", t, pbytes / t / 1e3);
    return 0;
}
