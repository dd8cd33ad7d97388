// store_writer.hip -- would dedicated writer waves (a few per CU, each streaming
// a wide slice of rays) write the desynchronised SoA packet stream faster than
// every compute wave storing its own 64 rays?  One wave per workgroup; LDS
// allocation caps the waves per CU; each wave owns 64*M consecutive rays and
// writes, row after row, M x (64 lanes x VEC doubles).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double d2 __attribute__((ext_vector_type(2)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int VEC>
__global__ void __launch_bounds__(64) writer(double *out, long ld, long n, int rows, int phase, int M)
{
    extern __shared__ double lds[];
    if (threadIdx.x == 0) lds[0] = 0;
    const long span = 64L * M;                      // rays per wave
    for (long blk = blockIdx.x; blk * span < n; blk += gridDim.x) {
        const long r0 = blk * span;
        const int k0 = (int)((blk * phase) % rows);
        for (int i = 0; i < rows; ++i) {
            int k = k0 + i; if (k >= rows) k -= rows;
            double *row = out + (long)k * ld + r0;
            if (VEC == 1) {
                for (int m = 0; m < M; ++m)
                    __builtin_nontemporal_store((double)k, row + m * 64 + threadIdx.x);
            } else {
                for (int m = 0; m < M / 2; ++m)
                    __builtin_nontemporal_store(d2{(double)k, 1.0}, (d2 *)row + m * 64 + threadIdx.x);
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
    CHECK(hipFuncSetAttribute((const void *)writer<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CHECK(hipFuncSetAttribute((const void *)writer<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    for (int wpc : {1, 2, 4, 8, 16}) {
        const int lds_b = wpc >= 16 ? 8 * 1024 : (160 / wpc) * 1024;
        for (int M : {4, 8, 16}) {
            const int grid = (int)(n / (64L * M));
            for (int phase : {0, 7}) {
                double t1 = time_us([&] { hipLaunchKernelGGL(writer<1>, dim3(grid), dim3(64), lds_b, 0, buf, ld, n, rows, phase, M); }, 10);
                double t2 = time_us([&] { hipLaunchKernelGGL(writer<2>, dim3(grid), dim3(64), lds_b, 0, buf, ld, n, rows, phase, M); }, 10);
                printf("{\"waves_per_cu\": %d, \"rays_per_wave\": %d, \"phase\": %d, \"x8_us\": %.1f, \"x8_GBps\": %.0f, \"x16_us\": %.1f, \"x16_GBps\": %.0f}\n",
                       wpc, 64 * M, phase, t1, bytes / t1 / 1e3, t2, bytes / t2 / 1e3);
            }
        }
    }
    return 0;
}
