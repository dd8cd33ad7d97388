// probe: how large may a by-value kernel argument be, and what does a launch cost with it?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <chrono>
#include <vector>
struct Item { double v[124]; };             // 992 bytes
template <int K> struct Big { const Item *ptr; Item inl[K]; };
typedef const __attribute__((address_space(4))) Item *CItem;
template <int K>
__global__ void kern(const Big<K> b, double *out)
{
    CItem base = (CItem)((const __attribute__((address_space(4))) char *)__builtin_amdgcn_kernarg_segment_ptr()
                         + offsetof(Big<K>, inl));
    CItem p = b.ptr ? (CItem)b.ptr + blockIdx.y : base + blockIdx.y;
    if (threadIdx.x == 0 && blockIdx.x == 0)
        out[blockIdx.y] = p->v[3] + p->v[123];
}
template <int K> int run(double *d_out, Item *d_items)
{
    Big<K> b{};
    for (int i = 0; i < K; ++i) { b.inl[i].v[3] = i; b.inl[i].v[123] = 1000; }
    hipLaunchKernelGGL(kern<K>, dim3(4, K), dim3(64), 0, 0, b, d_out);
    hipError_t e = hipDeviceSynchronize();
    std::vector<double> h(K);
    hipMemcpy(h.data(), d_out, K * 8, hipMemcpyDeviceToHost);
    bool ok = e == hipSuccess;
    for (int i = 0; i < K; ++i) ok = ok && h[i] == 1000 + i;
    // time: inline launch vs upload + pointer launch
    std::vector<Item> hi(K);
    Item *pin; hipHostMalloc(&pin, sizeof(Item) * K);
    for (int rep = 0; rep < 2; ++rep) {
        auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < 2000; ++i) hipLaunchKernelGGL(kern<K>, dim3(4, K), dim3(64), 0, 0, b, d_out);
        hipDeviceSynchronize();
        auto t1 = std::chrono::steady_clock::now();
        Big<K> c{}; c.ptr = d_items;
        for (int i = 0; i < 2000; ++i) {
            hipMemcpyAsync(d_items, pin, sizeof(Item) * K, hipMemcpyHostToDevice, 0);
            hipLaunchKernelGGL(kern<K>, dim3(4, K), dim3(64), 0, 0, c, d_out);
        }
        hipDeviceSynchronize();
        auto t2 = std::chrono::steady_clock::now();
        if (rep)
            printf("K=%d kernarg=%zu B ok=%d err=%s inline %.2f us/launch, upload+pointer %.2f us/launch\n", K, sizeof(Big<K>),
                   (int)ok, hipGetErrorString(e), std::chrono::duration<double, std::micro>(t1 - t0).count() / 2000,
                   std::chrono::duration<double, std::micro>(t2 - t1).count() / 2000);
    }
    hipHostFree(pin);
    return ok;
}
int main()
{
    double *d_out; hipMalloc(&d_out, 8 * 64);
    Item *d_items; hipMalloc(&d_items, sizeof(Item) * 64);
    run<4>(d_out, d_items); run<8>(d_out, d_items); run<16>(d_out, d_items); run<32>(d_out, d_items);
    return 0;
}
