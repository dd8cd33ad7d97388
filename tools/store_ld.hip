// store_ld.hip -- does the row pitch (ld) of the SoA packet buffer matter?
// Rows exactly 2^k bytes apart put the same ray index of all 130 rows on the
// same HBM channel/bank; a padded pitch spreads them.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <bool NT>
__global__ void __launch_bounds__(256) soa8(double *out, long ld, long n, int rows)
{
    for (long r = (long)blockIdx.x * 256 + threadIdx.x; r < n; r += (long)gridDim.x * 256) {
        double v = (double)r;
        for (int k = 0; k < rows; ++k) {
            if (NT) __builtin_nontemporal_store(v, out + (long)k * ld + r);
            else out[(long)k * ld + r] = v;
            v += 1.0;
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
    const int rows = 130;
    const long pads[] = {0, 16, 32, 64, 128, 256, 512, 1024, 2048, 4096, 8192, 16384, 65536, 1 << 17,
                         (1 << 17) + 512, 3 * 4096 + 64, 5 * 1024 + 32};
    double *buf;
    CHECK(hipMalloc(&buf, (size_t)rows * (n + (1 << 18)) * 8));
    const size_t bytes = (size_t)rows * n * 8;
    for (long pad : pads) {
        const long ld = n + pad;
        for (int g : {2048, 4096}) {
            double t = time_us([&] { hipLaunchKernelGGL(soa8<true>, dim3(g), dim3(256), 0, 0, buf, ld, n, rows); }, 20);
            double t2 = time_us([&] { hipLaunchKernelGGL(soa8<false>, dim3(g), dim3(256), 0, 0, buf, ld, n, rows); }, 20);
            printf("{\"pad\": %ld, \"grid\": %d, \"nt_us\": %.1f, \"nt_GBps\": %.0f, \"plain_us\": %.1f, \"plain_GBps\": %.0f}\n",
                   pad, g, t, bytes / t / 1e3, t2, bytes / t2 / 1e3);
        }
    }
    return 0;
}
