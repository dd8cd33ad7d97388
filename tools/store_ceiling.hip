// store_ceiling.hip -- what HBM write bandwidth does the trace kernel's store
// pattern admit on this GPU, with no arithmetic in the way?
//   pattern A  "soa8":  every lane writes ROWS doubles at stride ld (the FULL
//              packet layout seg[row][ray]: 512 B contiguous per wave-store)
//   pattern B  "soa16": two rays per lane, 16-B stores (1 KiB per wave-store)
//   pattern C  "linear": plain streaming fill, 16 B per lane, fully contiguous
// hipcc --offload-arch=gfx950 -O3 tools/store_ceiling.hip -o build/store_ceiling
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef double d2 __attribute__((ext_vector_type(2)));
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

template <bool NT>
__global__ void __launch_bounds__(256) soa16(d2 *out, long ld2, long n2, int rows)
{
    for (long r = (long)blockIdx.x * 256 + threadIdx.x; r < n2; r += (long)gridDim.x * 256) {
        d2 v{(double)r, (double)r};
        for (int k = 0; k < rows; ++k) {
            if (NT) __builtin_nontemporal_store(v, out + (long)k * ld2 + r);
            else out[(long)k * ld2 + r] = v;
            v.x += 1.0;
        }
    }
}

__global__ void __launch_bounds__(256) linear(d2 *out, long n2)
{
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n2; i += (long)gridDim.x * 256)
        __builtin_nontemporal_store(d2{1.0, 2.0}, out + i);
}

// pattern D "tiled": ray axis cut into tiles of T rays; a tile's ROWS x T block
// is contiguous (AoSoA).  One workgroup (256 lanes) walks T/256 sub-tiles.
template <bool NT>
__global__ void __launch_bounds__(256) tiled(double *out, long n, int rows, int T)
{
    const long ntiles = n / T;
    for (long t = blockIdx.x; t < ntiles; t += gridDim.x) {
        double *base = out + t * (long)rows * T;
        for (int s = threadIdx.x; s < T; s += 256) {
            double v = (double)s;
            for (int k = 0; k < rows; ++k) {
                if (NT) __builtin_nontemporal_store(v, base + (long)k * T + s);
                else base[(long)k * T + s] = v;
                v += 1.0;
            }
        }
    }
}

// pattern E: linear fill, 4 x 16 B per lane per iteration, block-contiguous 16 KiB
__global__ void __launch_bounds__(256) linear4(d2 *out, long n2)
{
    const long chunk = 4L * 256;
    for (long c = blockIdx.x; c * chunk < n2; c += gridDim.x) {
        d2 *b = out + c * chunk + threadIdx.x;
#pragma unroll
        for (int u = 0; u < 4; ++u)
            b[u * 256] = d2{1.0, 2.0};
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
    const int rows = 130;                   // 13 segments x 10 doubles
    const size_t bytes = (size_t)rows * n * 8;
    double *buf;
    CHECK(hipMalloc(&buf, bytes));
    const int grids[] = {1024, 2048, 4096};
    for (int g : grids) {
        double t;
        t = time_us([&] { hipLaunchKernelGGL(soa8<true>, dim3(g), dim3(256), 0, 0, buf, n, n, rows); }, 20);
        printf("{\"pattern\": \"soa8_nt\", \"grid\": %d, \"us\": %.1f, \"GBps\": %.0f}\n", g, t, bytes / t / 1e3);
        t = time_us([&] { hipLaunchKernelGGL(soa8<false>, dim3(g), dim3(256), 0, 0, buf, n, n, rows); }, 20);
        printf("{\"pattern\": \"soa8\", \"grid\": %d, \"us\": %.1f, \"GBps\": %.0f}\n", g, t, bytes / t / 1e3);
        t = time_us([&] { hipLaunchKernelGGL(soa16<true>, dim3(g), dim3(256), 0, 0, (d2 *)buf, n / 2, n / 2, rows); }, 20);
        printf("{\"pattern\": \"soa16_nt\", \"grid\": %d, \"us\": %.1f, \"GBps\": %.0f}\n", g, t, bytes / t / 1e3);
        t = time_us([&] { hipLaunchKernelGGL(soa16<false>, dim3(g), dim3(256), 0, 0, (d2 *)buf, n / 2, n / 2, rows); }, 20);
        printf("{\"pattern\": \"soa16\", \"grid\": %d, \"us\": %.1f, \"GBps\": %.0f}\n", g, t, bytes / t / 1e3);
        t = time_us([&] { hipLaunchKernelGGL(linear, dim3(g), dim3(256), 0, 0, (d2 *)buf, (long)(bytes / 16)); }, 20);
        printf("{\"pattern\": \"linear16_nt\", \"grid\": %d, \"us\": %.1f, \"GBps\": %.0f}\n", g, t, bytes / t / 1e3);
    }
    const int Ts[] = {256, 1024, 4096};
    for (int T : Ts) for (int g : {2048, 4096}) {
        double t = time_us([&] { hipLaunchKernelGGL(tiled<false>, dim3(g), dim3(256), 0, 0, buf, n, rows, T); }, 20);
        printf("{\"pattern\": \"tiled\", \"T\": %d, \"grid\": %d, \"us\": %.1f, \"GBps\": %.0f}\n", T, g, t, bytes / t / 1e3);
        t = time_us([&] { hipLaunchKernelGGL(tiled<true>, dim3(g), dim3(256), 0, 0, buf, n, rows, T); }, 20);
        printf("{\"pattern\": \"tiled_nt\", \"T\": %d, \"grid\": %d, \"us\": %.1f, \"GBps\": %.0f}\n", T, g, t, bytes / t / 1e3);
    }
    for (int g : {512, 1024, 2048, 8192}) {
        double t = time_us([&] { hipLaunchKernelGGL(linear4, dim3(g), dim3(256), 0, 0, (d2 *)buf, (long)(bytes / 16)); }, 20);
        printf("{\"pattern\": \"linear4x16\", \"grid\": %d, \"us\": %.1f, \"GBps\": %.0f}\n", g, t, bytes / t / 1e3);
    }
    double t = time_us([&] { CHECK(hipMemsetAsync(buf, 0, bytes, 0)); }, 10);
    printf("{\"pattern\": \"hipMemset\", \"us\": %.1f, \"GBps\": %.0f}\n", t, bytes / t / 1e3);
    return 0;
}
