// write_ceiling.hip -- what does a pure write stream reach on this GPU today, by shape?
// 1.09 GB per launch (the size of a FULL packet set of 2^20 rays x 13 segments), non-temporal
// and plain stores, hipMalloc memory:
//   one16     one 16-byte store per thread, blocks in address order (no grid stride)
//   one8      one 8-byte store per thread
//   stride16  grid-stride loop, 16 bytes per lane per iteration (4096 x 256 threads)
//   chunk     every workgroup fills its own contiguous 64 KiB, 16 bytes per lane per iteration
//   memset    hipMemsetAsync of the same bytes
// next to the packet pattern itself (rows<>: SoA rows, 1024-thread workgroups, barrier per segment)
//   hipcc --offload-arch=gfx950 -O3 tools/write_ceiling.hip -o build/write_ceiling
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef double d2 __attribute__((ext_vector_type(2)));

template <bool NT> __device__ __forceinline__ void st(d2 *p, d2 v) { if (NT) __builtin_nontemporal_store(v, p); else *p = v; }
template <bool NT> __device__ __forceinline__ void st(double *p, double v) { if (NT) __builtin_nontemporal_store(v, p); else *p = v; }

template <bool NT> __global__ void __launch_bounds__(256) one16(d2 *out)
{
    st<NT>(out + (long)blockIdx.x * 256 + threadIdx.x, d2{1.0, 2.0});
}
template <bool NT> __global__ void __launch_bounds__(256) one8(double *out)
{
    st<NT>(out + (long)blockIdx.x * 256 + threadIdx.x, 1.0);
}
template <bool NT> __global__ void __launch_bounds__(256) stride16(d2 *out, long n2)
{
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n2; i += (long)gridDim.x * 256)
        st<NT>(out + i, d2{1.0, 2.0});
}
template <bool NT> __global__ void __launch_bounds__(256) chunk(d2 *out)
{
    d2 *p = out + (long)blockIdx.x * 4096 + threadIdx.x;       // 64 KiB per workgroup
#pragma unroll
    for (int k = 0; k < 16; ++k)
        st<NT>(p + k * 256, d2{1.0, 2.0});
}
template <bool NT>
__global__ void __launch_bounds__(1024) rows(double *out, long ld, int segs, int phase)
{
    const long blk = blockIdx.x;
    const long r = blk * 1024 + threadIdx.x;
    const int s0 = (int)((blk * phase) % segs);
    double v = (double)r;
    for (int i = 0; i < segs; ++i) {
        __builtin_amdgcn_s_barrier();
        int sg = s0 + i; if (sg >= segs) sg -= segs;
        double *base = out + (long)sg * 10 * ld + r;
#pragma unroll
        for (int c = 0; c < 10; ++c) {
            st<NT>(base + (long)c * ld, v);
            v += 1.0;
        }
    }
}

template <class F>
double time_us(F f, int reps)
{
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    for (int i = 0; i < 300; ++i) f();
    CHECK(hipEventRecord(a));
    for (int i = 0; i < reps; ++i) f();
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms; CHECK(hipEventElapsedTime(&ms, a, b));
    return ms * 1e3 / reps;
}
#define REPORT(name, nt, call) do { double t = time_us([&] { call; }, 300); \
    printf("{\"shape\": \"%s\", \"stores\": \"%s\", \"us\": %.1f, \"GBps\": %.0f}\n", name, nt, t, bytes / t / 1e3); } while (0)

int main()
{
    const long n = 1024L * 1024, ld = n + 256;
    const int segs = 13;
    const size_t alloc = (size_t)segs * 10 * ld * 8, bytes = (size_t)segs * 10 * n * 8;
    const long n2 = bytes / 16, n1 = bytes / 8;
    double *buf;
    CHECK(hipMalloc(&buf, alloc));
    REPORT("one16", "nt", hipLaunchKernelGGL(one16<true>, dim3((unsigned)(n2 / 256)), dim3(256), 0, 0, (d2 *)buf));
    REPORT("one16", "plain", hipLaunchKernelGGL(one16<false>, dim3((unsigned)(n2 / 256)), dim3(256), 0, 0, (d2 *)buf));
    REPORT("one8", "nt", hipLaunchKernelGGL(one8<true>, dim3((unsigned)(n1 / 256)), dim3(256), 0, 0, buf));
    REPORT("one8", "plain", hipLaunchKernelGGL(one8<false>, dim3((unsigned)(n1 / 256)), dim3(256), 0, 0, buf));
    REPORT("stride16 4096x256", "nt", hipLaunchKernelGGL(stride16<true>, dim3(4096), dim3(256), 0, 0, (d2 *)buf, n2));
    REPORT("stride16 4096x256", "plain", hipLaunchKernelGGL(stride16<false>, dim3(4096), dim3(256), 0, 0, (d2 *)buf, n2));
    REPORT("stride16 1024x256", "nt", hipLaunchKernelGGL(stride16<true>, dim3(1024), dim3(256), 0, 0, (d2 *)buf, n2));
    REPORT("chunk 64KiB", "nt", hipLaunchKernelGGL(chunk<true>, dim3((unsigned)(n2 / 4096)), dim3(256), 0, 0, (d2 *)buf));
    REPORT("chunk 64KiB", "plain", hipLaunchKernelGGL(chunk<false>, dim3((unsigned)(n2 / 4096)), dim3(256), 0, 0, (d2 *)buf));
    REPORT("memset", "-", CHECK(hipMemsetAsync(buf, 0, bytes, 0)));
    for (int phase : {0, 1, 5}) {
        char nm[64]; snprintf(nm, sizeof nm, "packet rows, phase %d", phase);
        REPORT(nm, "nt", hipLaunchKernelGGL(rows<true>, dim3(1024), dim3(1024), 0, 0, buf, ld, segs, phase));
    }
    return 0;
}
