// store_occ.hip -- store-only SoA stream at the occupancy the trace kernel has.
// Dynamic LDS caps workgroups per CU: 40 KiB -> 4 WG/CU = 4 waves/SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void __launch_bounds__(256) soa8(double *out, long ld, long n, int rows, int burst)
{
    extern __shared__ double lds[];
    if (threadIdx.x == 0) lds[0] = 0;
    for (long r = (long)blockIdx.x * 256 + threadIdx.x; r < n; r += (long)gridDim.x * 256) {
        double v = (double)r;
        for (int k = 0; k < rows; ++k) {
            __builtin_nontemporal_store(v, out + (long)k * ld + r);
            v += 1.0;
            if (burst && (k % 10) == 9) {           // ~ one surface worth of fp64 work between bursts
                for (int j = 0; j < burst; ++j) v = __builtin_fma(v, 1.0000001, 1e-9);
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
    const long n = 1024L * 1024, ld = n + 256;
    const int rows = 130;
    double *buf;
    CHECK(hipMalloc(&buf, (size_t)rows * ld * 8));
    const size_t bytes = (size_t)rows * n * 8;
    CHECK(hipFuncSetAttribute((const void *)soa8, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    for (int lds_kb : {0, 20, 32, 40, 53, 80}) {
        for (int burst : {0, 100, 200, 300}) {
            double t = time_us([&] { hipLaunchKernelGGL(soa8, dim3(4096), dim3(256), lds_kb * 1024, 0, buf, ld, n, rows, burst); }, 10);
            printf("{\"lds_kb\": %d, \"wg_per_cu\": %d, \"fma_per_surface\": %d, \"us\": %.1f, \"GBps\": %.0f}\n",
                   lds_kb, lds_kb ? 160 / lds_kb : 8, burst, t, bytes / t / 1e3);
        }
    }
    return 0;
}
