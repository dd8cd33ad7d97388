// store_policy.hip -- cache policy of the packet stores on the desynchronised SoA
// stream: plain, nt, sc1 (agent-scope relaxed atomic store = write-through),
// sc0 sc1 nt via inline asm.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int POLICY>
__device__ __forceinline__ void put(double *p, double v)
{
    if (POLICY == 0) *p = v;
    else if (POLICY == 1) __builtin_nontemporal_store(v, p);
    else if (POLICY == 2) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else if (POLICY == 3) asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1 nt" :: "v"(p), "v"(v) : "memory");
    else if (POLICY == 4) asm volatile("global_store_dwordx2 %0, %1, off sc1 nt" :: "v"(p), "v"(v) : "memory");
    else asm volatile("global_store_dwordx2 %0, %1, off sc0" :: "v"(p), "v"(v) : "memory");
}

template <int POLICY>
__global__ void __launch_bounds__(256) soa(double *out, long ld, long n, int rows, int phase)
{
    for (long blk = blockIdx.x; blk * 256 < n; blk += gridDim.x) {
        const long r = blk * 256 + threadIdx.x;
        const int k0 = (int)((blk * phase) % rows);
        double v = (double)r;
        for (int i = 0; i < rows; ++i) {
            int k = k0 + i; if (k >= rows) k -= rows;
            put<POLICY>(out + (long)k * ld + r, v);
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

#define RUN(P, name) do { for (int phase : {0, 7}) { \
    double t = time_us([&] { hipLaunchKernelGGL(soa<P>, dim3(4096), dim3(256), 0, 0, buf, ld, n, rows, phase); }, 10); \
    printf("{\"policy\": \"%s\", \"phase\": %d, \"us\": %.1f, \"GBps\": %.0f}\n", name, phase, t, bytes / t / 1e3); } } while (0)

int main()
{
    const long n = 1024L * 1024, ld = n + 256;
    const int rows = 130;
    double *buf;
    CHECK(hipMalloc(&buf, (size_t)rows * ld * 8));
    const size_t bytes = (size_t)rows * n * 8;
    RUN(0, "plain"); RUN(1, "nt"); RUN(2, "sc1_atomic"); RUN(3, "sc0_sc1_nt"); RUN(4, "sc1_nt"); RUN(5, "sc0");
    return 0;
}
