// store_alloc.hip -- does the memory type of the packet buffer change what the FULL packet
// store pattern reaches?  The shipped pattern (SoA rows, 1024-thread workgroups, a barrier per
// segment, non-temporal 8-byte stores; tools/store_pairs.hip x2) into
//   default      hipMalloc (what torch.empty gives: cached, MTYPE RW)
//   uncached     hipExtMallocWithFlags(hipDeviceMallocUncached)
//   finegrained  hipExtMallocWithFlags(hipDeviceMallocFinegrained)
// plus a plain linear fill of the same number of bytes as the write ceiling of the day.
//   hipcc --offload-arch=gfx950 -O3 tools/store_alloc.hip -o /tmp/store_alloc && /tmp/store_alloc
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef double d2 __attribute__((ext_vector_type(2)));

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
            if (NT) __builtin_nontemporal_store(v, base + (long)c * ld);
            else base[(long)c * ld] = v;
            v += 1.0;
        }
    }
}

__global__ void __launch_bounds__(256) linear(d2 *out, long n2)
{
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n2; i += (long)gridDim.x * 256)
        __builtin_nontemporal_store(d2{1.0, 2.0}, out + i);
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

int main()
{
    const long n = 1024L * 1024, ld = n + 256;
    const int segs = 13;
    const size_t alloc = (size_t)segs * 10 * ld * 8, bytes = (size_t)segs * 10 * n * 8;
    const char *names[] = {"default", "uncached", "finegrained"};
    for (int kind = 0; kind < 3; ++kind) {
        double *buf = nullptr;
        hipError_t e = kind == 0 ? hipMalloc(&buf, alloc)
                     : hipExtMallocWithFlags((void **)&buf, alloc, kind == 1 ? hipDeviceMallocUncached : hipDeviceMallocFinegrained);
        if (e != hipSuccess) { printf("{\"memory\": \"%s\", \"error\": \"%s\"}\n", names[kind], hipGetErrorString(e)); continue; }
        for (int phase : {0, 1, 5}) {
            double t = time_us([&] { hipLaunchKernelGGL(rows<true>, dim3(1024), dim3(1024), 0, 0, buf, ld, segs, phase); }, 300);
            printf("{\"memory\": \"%s\", \"stores\": \"nt\", \"phase\": %d, \"us\": %.1f, \"GBps\": %.0f}\n", names[kind], phase, t, bytes / t / 1e3);
            t = time_us([&] { hipLaunchKernelGGL(rows<false>, dim3(1024), dim3(1024), 0, 0, buf, ld, segs, phase); }, 300);
            printf("{\"memory\": \"%s\", \"stores\": \"plain\", \"phase\": %d, \"us\": %.1f, \"GBps\": %.0f}\n", names[kind], phase, t, bytes / t / 1e3);
        }
        double t = time_us([&] { hipLaunchKernelGGL(linear, dim3(4096), dim3(256), 0, 0, (d2 *)buf, (long)(bytes / 16)); }, 300);
        printf("{\"memory\": \"%s\", \"stores\": \"linear fill, 16 B per lane\", \"us\": %.1f, \"GBps\": %.0f}\n", names[kind], t, bytes / t / 1e3);
        CHECK(hipFree(buf));
    }
    return 0;
}
