// store_rot.hip -- is the packet write stream limited by how many separate address windows the
// chip writes at once?  tools/write_ceiling.hip: one dense moving window reaches 6.7-6.9 TB/s, a
// workgroup-strided one (every workgroup filling its own 64 KiB) 5.0-5.5, the packet rows
// (10 dense windows in lock-step, up to 130 when workgroups drift apart) 5.1-5.6.
// Here: layouts in which a segment of a 1024-ray tile is one contiguous 80 KiB block
// ([segment][tile][10][1024]), so that a segment is ONE window chip-wide --
//   segtile      every workgroup writes its ten 8 KiB pieces in the same order (at any instant
//                the chip writes at an 80 KiB stride: the strided case above)
//   segtile_rot  workgroup b starts at piece b % 10: at any instant all ten pieces of the
//                window are being written somewhere -- a dense window
//   rows         the shipped SoA rows, for reference
// each with workgroups in lock-step (phase 0) and started at different segments (1, 5).
//   hipcc --offload-arch=gfx950 -O3 tools/store_rot.hip -o build/store_rot
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int MODE>     // 0 rows (SoA), 1 segtile, 2 segtile rotated
__global__ void __launch_bounds__(1024) k(double *out, long ld, long ntiles, int segs, int phase)
{
    const long blk = blockIdx.x;
    const int s0 = (int)((blk * phase) % segs);
    double v = (double)threadIdx.x;
    for (int i = 0; i < segs; ++i) {
        __builtin_amdgcn_s_barrier();
        int sg = s0 + i; if (sg >= segs) sg -= segs;
        if (MODE == 0) {
            double *base = out + (long)sg * 10 * ld + blk * 1024 + threadIdx.x;
#pragma unroll
            for (int c = 0; c < 10; ++c) { __builtin_nontemporal_store(v, base + (long)c * ld); v += 1.0; }
        } else {
            double *base = out + ((long)sg * ntiles + blk) * 10240 + threadIdx.x;
            const int r0 = MODE == 2 ? (int)(blk % 10) : 0;
#pragma unroll
            for (int c = 0; c < 10; ++c) {
                int pc = c + r0; if (pc >= 10) pc -= 10;
                __builtin_nontemporal_store(v, base + (long)pc * 1024);
                v += 1.0;
            }
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

int main()
{
    const long n = 1024L * 1024, ld = n + 256, ntiles = n / 1024;
    const int segs = 13;
    const size_t alloc = (size_t)segs * 10 * ld * 8, bytes = (size_t)segs * 10 * n * 8;
    double *buf;
    CHECK(hipMalloc(&buf, alloc));
    const char *names[] = {"rows (SoA, shipped)", "segtile", "segtile, rotated piece order"};
    for (int rep = 0; rep < 2; ++rep)
        for (int phase : {0, 1, 5}) {
            double t0 = time_us([&] { hipLaunchKernelGGL(k<0>, dim3(1024), dim3(1024), 0, 0, buf, ld, ntiles, segs, phase); }, 300);
            double t1 = time_us([&] { hipLaunchKernelGGL(k<1>, dim3(1024), dim3(1024), 0, 0, buf, ld, ntiles, segs, phase); }, 300);
            double t2 = time_us([&] { hipLaunchKernelGGL(k<2>, dim3(1024), dim3(1024), 0, 0, buf, ld, ntiles, segs, phase); }, 300);
            const double t[3] = {t0, t1, t2};
            for (int m = 0; m < 3; ++m)
                printf("{\"layout\": \"%s\", \"phase\": %d, \"us\": %.1f, \"GBps\": %.0f}\n", names[m], phase, t[m], bytes / t[m] / 1e3);
        }
    return 0;
}
