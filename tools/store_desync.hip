// store_desync.hip -- the trace kernel's waves are spread over all 13 surfaces,
// so the chip writes ~130 packet rows concurrently.  How do the SoA layout and a
// per-workgroup-contiguous ("tiled") layout take that, compared with the
// lock-step row sweep of the plain store micro-benchmarks?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// SoA: element (k, r) at out[k*ld + r]; workgroup b starts at row (b*phase) % rows
__global__ void __launch_bounds__(256) soa(double *out, long ld, long n, int rows, int phase)
{
    for (long blk = blockIdx.x; blk * 256 < n; blk += gridDim.x) {
        const long r = blk * 256 + threadIdx.x;
        const int k0 = (int)((blk * phase) % rows);
        double v = (double)r;
        for (int i = 0; i < rows; ++i) {
            int k = k0 + i; if (k >= rows) k -= rows;
            __builtin_nontemporal_store(v, out + (long)k * ld + r);
            v += 1.0;
        }
    }
}

// tiled: element (k, r) at out[(r/T)*rows*T + k*T + r%T], T = 256 (one workgroup)
__global__ void __launch_bounds__(256) tiled(double *out, long ld, long n, int rows, int phase)
{
    for (long blk = blockIdx.x; blk * 256 < n; blk += gridDim.x) {
        double *base = out + blk * (long)rows * 256 + threadIdx.x;
        const int k0 = (int)((blk * phase) % rows);
        double v = (double)threadIdx.x;
        for (int i = 0; i < rows; ++i) {
            int k = k0 + i; if (k >= rows) k -= rows;
            __builtin_nontemporal_store(v, base + (long)k * 256);
            v += 1.0;
        }
    }
}

// segment-major tiles: element (seg, c, r) at out[((seg*tiles + r/T)*10 + c)*T + r%T]:
// 13 streams (one per segment), each workgroup writing 10*T*8 contiguous bytes per segment
template <int T>
__global__ void __launch_bounds__(T) segtile(double *out, long tiles, long n, int rows, int phase)
{
    const int segs = rows / 10;
    for (long blk = blockIdx.x; blk * T < n; blk += gridDim.x) {
        const int s0 = (int)((blk * phase) % segs);
        double v = (double)threadIdx.x;
        for (int i = 0; i < segs; ++i) {
            int sg = s0 + i; if (sg >= segs) sg -= segs;
            double *base = out + ((long)sg * tiles + blk) * 10 * T + threadIdx.x;
            for (int c = 0; c < 10; ++c) {
                __builtin_nontemporal_store(v, base + (long)c * T);
                v += 1.0;
            }
        }
    }
}

// SoA with 512-thread workgroups (the trace kernel's shape)
__global__ void __launch_bounds__(512) soa512(double *out, long ld, long n, int rows, int phase)
{
    for (long blk = blockIdx.x; blk * 512 < n; blk += gridDim.x) {
        const long r = blk * 512 + threadIdx.x;
        const int k0 = (int)((blk * phase) % rows);
        double v = (double)r;
        for (int i = 0; i < rows; ++i) {
            int k = k0 + i; if (k >= rows) k -= rows;
            __builtin_nontemporal_store(v, out + (long)k * ld + r);
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
    const long n = 1024L * 1024, ld = n + 256;
    const int rows = 130;
    double *buf;
    CHECK(hipMalloc(&buf, (size_t)rows * ld * 8));
    const size_t bytes = (size_t)rows * n * 8;
    for (int phase : {0, 1, 7, 37}) {
        double t = time_us([&] { hipLaunchKernelGGL(soa, dim3(4096), dim3(256), 0, 0, buf, ld, n, rows, phase); }, 10);
        printf("{\"layout\": \"soa\", \"phase\": %d, \"us\": %.1f, \"GBps\": %.0f}\n", phase, t, bytes / t / 1e3);
        t = time_us([&] { hipLaunchKernelGGL(tiled, dim3(4096), dim3(256), 0, 0, buf, ld, n, rows, phase); }, 10);
        printf("{\"layout\": \"tiled256\", \"phase\": %d, \"us\": %.1f, \"GBps\": %.0f}\n", phase, t, bytes / t / 1e3);
        t = time_us([&] { hipLaunchKernelGGL(soa512, dim3(2048), dim3(512), 0, 0, buf, ld, n, rows, phase); }, 10);
        printf("{\"layout\": \"soa512\", \"phase\": %d, \"us\": %.1f, \"GBps\": %.0f}\n", phase, t, bytes / t / 1e3);
        t = time_us([&] { hipLaunchKernelGGL(segtile<256>, dim3(4096), dim3(256), 0, 0, buf, n / 256, n, rows, phase); }, 10);
        printf("{\"layout\": \"segtile256\", \"phase\": %d, \"us\": %.1f, \"GBps\": %.0f}\n", phase, t, bytes / t / 1e3);
        t = time_us([&] { hipLaunchKernelGGL(segtile<512>, dim3(2048), dim3(512), 0, 0, buf, n / 512, n, rows, phase); }, 10);
        printf("{\"layout\": \"segtile512\", \"phase\": %d, \"us\": %.1f, \"GBps\": %.0f}\n", phase, t, bytes / t / 1e3);
        t = time_us([&] { hipLaunchKernelGGL(segtile<1024>, dim3(1024), dim3(1024), 0, 0, buf, n / 1024, n, rows, phase); }, 10);
        printf("{\"layout\": \"segtile1024\", \"phase\": %d, \"us\": %.1f, \"GBps\": %.0f}\n", phase, t, bytes / t / 1e3);
    }
    return 0;
}
