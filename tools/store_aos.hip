// store_aos.hip -- packet layout [segment][ray][10]: a wave writes one 5 KiB
// contiguous chunk per segment (64 rays x 80 B, as 5 x 16 B per lane), so the
// chip has 13 concurrent write streams instead of 130.  Workgroups are spread
// over the segments (phase) like the trace kernel's waves are.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double d2 __attribute__((ext_vector_type(2)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <bool NT>
__global__ void __launch_bounds__(256) aos(double *out, long plane, long n, int segs, int phase)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (long blk = blockIdx.x; blk * 256 < n; blk += gridDim.x) {
        const long r0 = blk * 256 + wave * 64;              // first ray of this wave
        const int k0 = (int)((blk * phase) % segs);
        for (int i = 0; i < segs; ++i) {
            int k = k0 + i; if (k >= segs) k -= segs;
            d2 *chunk = (d2 *)(out + (long)k * plane + r0 * 10);   // 5 KiB = 320 x 16 B
            d2 v{(double)k, (double)lane};
#pragma unroll
            for (int u = 0; u < 5; ++u) {
                if (NT) __builtin_nontemporal_store(v, chunk + u * 64 + lane);
                else chunk[u * 64 + lane] = v;
            }
        }
    }
}

template <class F>
double time_us(F f, int reps)
{
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    f(); f();
    CHECK(hipEventRecord(a));
    for (int i = 0; i < reps; ++i) f();
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms; CHECK(hipEventElapsedTime(&ms, a, b));
    return ms * 1e3 / reps;
}

int main()
{
    const long n = 1024L * 1024;
    const int segs = 13;
    double *buf;
    const long pads[] = {0, 256, 2560};
    CHECK(hipMalloc(&buf, (size_t)segs * (n * 10 + 4096) * 8));
    const size_t bytes = (size_t)segs * n * 80;
    for (long pad : pads) for (int phase : {0, 1, 5}) {
        const long plane = n * 10 + pad;
        double t = time_us([&] { hipLaunchKernelGGL(aos<true>, dim3(4096), dim3(256), 0, 0, buf, plane, n, segs, phase); }, 10);
        double t2 = time_us([&] { hipLaunchKernelGGL(aos<false>, dim3(4096), dim3(256), 0, 0, buf, plane, n, segs, phase); }, 10);
        printf("{\"layout\": \"seg_ray_10\", \"plane_pad\": %ld, \"phase\": %d, \"nt_us\": %.1f, \"nt_GBps\": %.0f, \"plain_us\": %.1f, \"plain_GBps\": %.0f}\n",
               pad, phase, t, bytes / t / 1e3, t2, bytes / t2 / 1e3);
    }
    return 0;
}
