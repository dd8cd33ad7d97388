// pack.hip -- the pack pass of two-pass packed hits.
//
// ROX_OUT_HITS_COMPACT is what SequentialModel.trace_grid(spot, ..., form='list',
// append_if_none=False) returns (rayoptics/seq/sequential.py:1058-1085 around
// rayoptics/mpl/axisarrayfigure.py:229-238): the (x, y) of the rays that got through, packed in
// ray order.  The fused instance (rox_device.hpp trace_tiles<HITS_COMPACT>) compacts inside the
// trace: one persistent 1024-thread workgroup per CU, three workgroup barriers and a look-back
// per 1024 rays -- cheap on a shallow system, but on a deep one (44 interfaces: the lithography
// lens of BASELINE configs[4]) or one with Newton iterations the waves of a tile finish far
// apart and the tile-synchronous workgroup idles: 93 ms against 70 ms for the plain HITS launch
// of the same rays.  For such tables the library runs the unsynchronised HITS instance into a
// scratch [2][R] + status and then this kernel: a pure streaming pass (17 B read per ray, 16 B
// written per survivor) that puts every pair exactly where the fused instance would have --
// same ticket / tile-state / running-base protocol, so chunked launches, ROX_HITS_APPEND and
// the capacity clamp behave identically.
#include "rox_device.hpp"

namespace rox {

__global__ void __launch_bounds__(kPackBlock)
pack_kernel(const PackArgs a)
{
    __shared__ int64_t s_tile;
    __shared__ int32_t s_wcnt[kPackSub * (kPackBlock / 64)];    // survivors per (sub-row, wave)
    __shared__ int32_t s_woff[kPackSub * (kPackBlock / 64)];    // exclusive offsets of the same
    __shared__ int32_t s_total;
    __shared__ uint32_t s_excl;
    static_assert(kPackSub * (kPackBlock / 64) == 64, "one wave scans the (sub-row, wave) counts");
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t n_tiles = (a.n_rays + kPackTile - 1) / kPackTile;
    d2 *dst = reinterpret_cast<d2 *>(a.dst);
    for (;;) {
        if (threadIdx.x == 0)
            s_tile = (int64_t)atomicAdd(&a.ticket[0], 1u);
        __syncthreads();
        const int64_t tile = (int64_t)__builtin_amdgcn_readfirstlane((int)s_tile);
        if (tile >= n_tiles)
            break;
        const int64_t r0 = tile * kPackTile + threadIdx.x;
        bool ok[kPackSub];
        d2 xy[kPackSub];
        int lrank[kPackSub];
#pragma unroll
        for (int k = 0; k < kPackSub; ++k) {        // sub-row k: rays r0 + k * kPackBlock (coalesced)
            const int64_t r = r0 + (int64_t)k * kPackBlock;
            ok[k] = r < a.n_rays && a.status[r] == ROX_OK;
            xy[k] = d2{0.0, 0.0};
            if (ok[k]) {
                xy[k].x = __builtin_nontemporal_load(a.xy + r);
                xy[k].y = __builtin_nontemporal_load(a.xy + a.ld + r);
            }
            const uint64_t mask = __ballot(ok[k]);
            lrank[k] = __popcll(mask & ((1ull << lane) - 1ull));
            if (lane == 0)
                s_wcnt[k * (kPackBlock / 64) + wave] = __popcll(mask);
        }
        __syncthreads();
        if (wave == 0) {
            // exclusive scan of the 64 (sub-row, wave) counts, then the tile's place
            const int cnt = s_wcnt[lane];
            int inc = cnt;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int up = __shfl_up(inc, o);
                if (lane >= o)
                    inc += up;
            }
            s_woff[lane] = inc - cnt;
            const int total = __shfl(inc, 63);
            if (lane == 0) {
                s_total = total;
                __hip_atomic_store(&a.tile_state[tile],
                                   ts_pack(a.epoch, tile == 0 ? TS_PREFIX : TS_AGG, (uint32_t)total),
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            const uint32_t excl = look_back(a.tile_state, a.epoch, tile, total, lane);
            if (lane == 0)
                s_excl = excl;
        }
        __syncthreads();
        const int64_t base = a.hits_base_in ? *a.hits_base_in : 0;
        const int64_t at = (base < 0 ? -base : base) + (int64_t)s_excl;
        const int total = s_total;
        const int64_t room = base < 0 ? 0 : a.ld_dst - at;      // pairs that still fit (may be <= 0)
#pragma unroll
        for (int k = 0; k < kPackSub; ++k) {
            const int j = s_woff[k * (kPackBlock / 64) + wave] + lrank[k];
            if (ok[k] && j < room)
                __builtin_nontemporal_store(xy[k], dst + at + j);
        }
        if (tile == n_tiles - 1 && threadIdx.x == 0) {
            const int64_t all = at + total;
            *a.hits_total_out = (base < 0 || all > a.ld_dst) ? -all : all;
        }
    }
    if (threadIdx.x == 0) {
        // the last workgroup out re-arms the ticket for the next launch of this stream context
        if (atomicAdd(&a.ticket[1], 1u) == gridDim.x - 1) {
            __hip_atomic_store(&a.ticket[0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&a.ticket[1], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

void launch_pack(const PackArgs &a, unsigned blocks, hipStream_t st)
{
    hipLaunchKernelGGL(pack_kernel, dim3(blocks), dim3(kPackBlock), 0, st, a);
}

}  // namespace rox
