// store_pairs.hip -- would 16-byte stores help the FULL packet writer?
// SoA rows [segment][component][ray]; 1024-thread workgroups, a barrier per segment, workgroups
// started at different segments (phase) as in the trace kernel.  x2: every lane stores its own
// 8-byte value to each of the 10 component rows (the shipped pattern).  x4: neighbouring lanes swap
// one value per component pair, even lanes store (ray, ray+1) of component 2j, odd lanes of
// component 2j+1 -- 5 store instructions of 16 bytes per lane instead of 10 of 8.
//   hipcc --offload-arch=gfx950 -O3 tools/store_pairs.hip -o /tmp/store_pairs && /tmp/store_pairs
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef double d2 __attribute__((ext_vector_type(2)));

__global__ void __launch_bounds__(1024) x2(double *out, long ld, long n, int segs, int phase, int sync)
{
    const long blk = blockIdx.x;
    const long r = blk * 1024 + threadIdx.x;
    const int s0 = (int)((blk * phase) % segs);
    double v = (double)r;
    for (int i = 0; i < segs; ++i) {
        if (sync)
            __builtin_amdgcn_s_barrier();
        int sg = s0 + i; if (sg >= segs) sg -= segs;
        double *base = out + (long)sg * 10 * ld + r;
#pragma unroll
        for (int c = 0; c < 10; ++c) {
            __builtin_nontemporal_store(v, base + (long)c * ld);
            v += 1.0;
        }
    }
}

__global__ void __launch_bounds__(1024) x4(double *out, long ld, long n, int segs, int phase, int sync)
{
    const long blk = blockIdx.x;
    const long r = blk * 1024 + threadIdx.x;
    const int odd = threadIdx.x & 1;
    const int s0 = (int)((blk * phase) % segs);
    double v = (double)r;
    for (int i = 0; i < segs; ++i) {
        if (sync)
            __builtin_amdgcn_s_barrier();
        int sg = s0 + i; if (sg >= segs) sg -= segs;
        // lane pair (2k, 2k+1): even lane writes rays 2k, 2k+1 of component 2j, odd lane of 2j+1
        double *base = out + (long)sg * 10 * ld + (r & ~1L) + (long)odd * ld;
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const double a = v, b = v + 1.0;            // this lane's components 2j and 2j+1
            const double send = odd ? a : b;            // even sends 2j+1, odd sends 2j
            const int lo = __builtin_amdgcn_mov_dpp(__double2loint(send), 0xB1, 0xF, 0xF, true);  // quad_perm [1,0,3,2]
            const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(send), 0xB1, 0xF, 0xF, true);
            const double got = __hiloint2double(hi, lo);
            const d2 w = odd ? d2{got, b} : d2{a, got};
            __builtin_nontemporal_store(w, (d2 *)(base + (long)(2 * j) * ld));
            v += 2.0;
        }
    }
}

// tiled: the 130 rows of a workgroup's 1024 rays are contiguous (1.04 MB per workgroup):
// element (row k, ray r) at out[((r / 1024) * rows + k) * 1024 + r % 1024]
__global__ void __launch_bounds__(1024) tiled(double *out, long ld, long n, int segs, int phase, int sync)
{
    const long blk = blockIdx.x;
    const int s0 = (int)((blk * phase) % segs);
    double v = (double)threadIdx.x;
    for (int i = 0; i < segs; ++i) {
        if (sync)
            __builtin_amdgcn_s_barrier();
        int sg = s0 + i; if (sg >= segs) sg -= segs;
        double *base = out + ((long)blk * segs * 10 + (long)sg * 10) * 1024 + threadIdx.x;
#pragma unroll
        for (int c = 0; c < 10; ++c) {
            __builtin_nontemporal_store(v, base + (long)c * 1024);
            v += 1.0;
        }
    }
}

template <class F>
double time_us(F f, int reps)
{
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    for (int i = 0; i < 200; ++i) f();                  // clocks
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
    double *buf;
    CHECK(hipMalloc(&buf, (size_t)segs * 10 * ld * 8));
    const size_t bytes = (size_t)segs * 10 * n * 8;
    for (int sync : {0, 1})
        for (int phase : {0, 1, 5}) {
            double t = time_us([&] { hipLaunchKernelGGL(x2, dim3(1024), dim3(1024), 0, 0, buf, ld, n, segs, phase, sync); }, 300);
            printf("{\"stores\": \"10 x 8 B\", \"barrier\": %d, \"phase\": %d, \"us\": %.1f, \"GBps\": %.0f}\n", sync, phase, t, bytes / t / 1e3);
            t = time_us([&] { hipLaunchKernelGGL(x4, dim3(1024), dim3(1024), 0, 0, buf, ld, n, segs, phase, sync); }, 300);
            printf("{\"stores\": \"5 x 16 B\", \"barrier\": %d, \"phase\": %d, \"us\": %.1f, \"GBps\": %.0f}\n", sync, phase, t, bytes / t / 1e3);
            t = time_us([&] { hipLaunchKernelGGL(tiled, dim3(1024), dim3(1024), 0, 0, buf, ld, n, segs, phase, sync); }, 300);
            printf("{\"stores\": \"tiled per workgroup, 10 x 8 B\", \"barrier\": %d, \"phase\": %d, \"us\": %.1f, \"GBps\": %.0f}\n", sync, phase, t, bytes / t / 1e3);
        }
    return 0;
}
