// store_patterns.hip -- which *linear* write pattern reaches hipMemset's rate?
// Every variant writes the same 1.09 GB; only the lane->address map changes.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double d2 __attribute__((ext_vector_type(2)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// grid-stride, VEC doubles per lane per store, UNROLL stores per iteration,
// consecutive stores of one lane are `blockDim*VEC` doubles apart (block-contiguous)
template <int VEC, int UNROLL, bool NT>
__global__ void fill(double *out, long n)
{
    const long per_iter = (long)blockDim.x * VEC * UNROLL;
    for (long base = (long)blockIdx.x * per_iter; base < n; base += (long)gridDim.x * per_iter) {
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            long i = base + ((long)u * blockDim.x + threadIdx.x) * VEC;
            if (i < n) {
                if (VEC == 2) {
                    if (NT) __builtin_nontemporal_store(d2{1.0, 2.0}, (d2 *)(out + i));
                    else *(d2 *)(out + i) = d2{1.0, 2.0};
                } else {
                    if (NT) __builtin_nontemporal_store(1.0, out + i);
                    else out[i] = 1.0;
                }
            }
        }
    }
}

// each block owns ONE contiguous span of n/gridDim doubles and walks it front to back
template <int VEC>
__global__ void span(double *out, long n)
{
    const long per = n / gridDim.x;
    double *b = out + (long)blockIdx.x * per;
    for (long i = (long)threadIdx.x * VEC; i < per; i += (long)blockDim.x * VEC) {
        if (VEC == 2) *(d2 *)(b + i) = d2{1.0, 2.0};
        else b[i] = 1.0;
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

#define RUN(name, kern, grid, block)                                                        \
    do {                                                                                    \
        double t = time_us([&] { hipLaunchKernelGGL(kern, dim3(grid), dim3(block), 0, 0, buf, n); }, 20); \
        printf("{\"pattern\": \"%s\", \"grid\": %d, \"block\": %d, \"us\": %.1f, \"GBps\": %.0f}\n", name, grid, block, t, bytes / t / 1e3); \
    } while (0)

int main()
{
    const long n = 130L * 1024 * 1024;
    const size_t bytes = (size_t)n * 8;
    double *buf;
    CHECK(hipMalloc(&buf, bytes));
    for (int g : {256, 512, 1024, 2048, 4096, 16384, 65536}) {
        RUN("x1_u1", (fill<1, 1, false>), g, 256);
        RUN("x2_u1", (fill<2, 1, false>), g, 256);
        RUN("x1_u4", (fill<1, 4, false>), g, 256);
        RUN("x2_u4", (fill<2, 4, false>), g, 256);
        RUN("x2_u8", (fill<2, 8, false>), g, 256);
        RUN("x2_u4_nt", (fill<2, 4, true>), g, 256);
    }
    for (int g : {256, 1024, 4096}) {
        RUN("x2_u4_b1024", (fill<2, 4, false>), g, 1024);
        RUN("x2_u4_b64", (fill<2, 4, false>), g * 4, 64);
        RUN("span_x2", (span<2>), g, 256);
        RUN("span_x1", (span<1>), g, 256);
    }
    double t = time_us([&] { CHECK(hipMemsetAsync(buf, 0, bytes, 0)); }, 10);
    printf("{\"pattern\": \"hipMemset\", \"us\": %.1f, \"GBps\": %.0f}\n", t, bytes / t / 1e3);
    t = time_us([&] { CHECK(hipMemsetD32Async((hipDeviceptr_t)buf, 0x3f800000, bytes / 4, 0)); }, 10);
    printf("{\"pattern\": \"hipMemsetD32\", \"us\": %.1f, \"GBps\": %.0f}\n", t, bytes / t / 1e3);
    return 0;
}
