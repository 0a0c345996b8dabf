// k0_membench.hip — measurement aid, not part of the hot path: plain streaming kernels that
// establish the HBM ceiling of the box the roofline fractions are quoted against
// (MI355X_MICROARCH.md: 8.0 TB/s spec, ~6.3 TB/s measured for a float4 copy).
#include "dyd_common.h"

namespace dyd {

constexpr int K0_BLOCK = 256;

// mode 0: dst[i] = src[i] (16 B per lane);  mode 1: read-only (sum folded into dst[0] only if
// non-zero to keep the loads alive);  mode 2: write-only.
template <int MODE>
__global__ __launch_bounds__(K0_BLOCK) void k0_stream(const uint4 *__restrict__ src, uint4 *__restrict__ dst,
                                                      int64_t n16) {
    const int64_t stride = (int64_t)gridDim.x * K0_BLOCK;
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (int64_t i = (int64_t)blockIdx.x * K0_BLOCK + threadIdx.x; i < n16; i += stride) {
        if (MODE == 0) {
            dst[i] = src[i];
        } else if (MODE == 3) {  // copy with non-temporal loads and stores
            const uint4 v = make_uint4(__builtin_nontemporal_load(&src[i].x), __builtin_nontemporal_load(&src[i].y),
                                       __builtin_nontemporal_load(&src[i].z), __builtin_nontemporal_load(&src[i].w));
            __builtin_nontemporal_store(v.x, &dst[i].x); __builtin_nontemporal_store(v.y, &dst[i].y);
            __builtin_nontemporal_store(v.z, &dst[i].z); __builtin_nontemporal_store(v.w, &dst[i].w);
        } else if (MODE == 4) {  // nt read only
            const uint4 v = make_uint4(__builtin_nontemporal_load(&src[i].x), __builtin_nontemporal_load(&src[i].y),
                                       __builtin_nontemporal_load(&src[i].z), __builtin_nontemporal_load(&src[i].w));
            acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w;
        } else if (MODE == 5) {  // nt write only
            __builtin_nontemporal_store((unsigned)i, &dst[i].x); __builtin_nontemporal_store(1u, &dst[i].y);
            __builtin_nontemporal_store(2u, &dst[i].z); __builtin_nontemporal_store(3u, &dst[i].w);
        } else if (MODE == 1) {
            const uint4 v = src[i];
            acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w;
        } else {
            dst[i] = make_uint4((unsigned)i, 1u, 2u, 3u);
        }
    }
    if ((MODE == 1 || MODE == 4) && (acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x9e3779b9u) dst[0] = acc;
}

// The fused kernel's stream: 72 % of its bytes are read, 28 % written (16 P + 4 B + 4 N in, 48 B + N out).  modes 10-12 move
// exactly that mix and nothing else: per iteration a lane issues FIVE 16-byte loads from the read stream before it touches any of
// them (with 8 workgroups of 256 lanes per CU that is 160 KiB in flight per CU) and TWO 16-byte stores to the write stream —
// 5 : 2 = 71.4 % : 28.6 %.  mode 10 plain loads and stores, 11 plain loads + non-temporal stores (what k12_wave_kernel does),
// 12 non-temporal both.  `n16` counts the READ stream's 16-byte units; the write stream gets 2/5 as many.
template <int MODE>
__global__ __launch_bounds__(K0_BLOCK) void k0_mix(const uint4 *__restrict__ src, uint4 *__restrict__ dst, int64_t n16) {
    const int64_t stride = (int64_t)gridDim.x * K0_BLOCK;
    const int64_t groups = n16 / 5;
    for (int64_t g = (int64_t)blockIdx.x * K0_BLOCK + threadIdx.x; g < groups; g += stride) {
        uint4 v[5];
#pragma unroll
        for (int k = 0; k < 5; ++k) {          // lane-contiguous inside every one of the five sub-streams: 256 lanes x 16 B per instruction
            const uint4 *q = src + (int64_t)k * groups + g;
            if (MODE == 12) v[k] = make_uint4(__builtin_nontemporal_load(&q->x), __builtin_nontemporal_load(&q->y),
                                              __builtin_nontemporal_load(&q->z), __builtin_nontemporal_load(&q->w));
            else v[k] = *q;
        }
        const uint4 a = make_uint4(v[0].x ^ v[1].x ^ v[4].x, v[0].y ^ v[1].y ^ v[4].y, v[0].z ^ v[1].z, v[0].w ^ v[1].w);
        const uint4 b = make_uint4(v[2].x ^ v[3].x ^ v[4].z, v[2].y ^ v[3].y ^ v[4].w, v[2].z ^ v[3].z, v[2].w ^ v[3].w);
        uint4 *o0 = dst + g, *o1 = dst + groups + g;
        if (MODE == 10) {
            *o0 = a;
            *o1 = b;
        } else {
            __builtin_nontemporal_store(a.x, &o0->x); __builtin_nontemporal_store(a.y, &o0->y);
            __builtin_nontemporal_store(a.z, &o0->z); __builtin_nontemporal_store(a.w, &o0->w);
            __builtin_nontemporal_store(b.x, &o1->x); __builtin_nontemporal_store(b.y, &o1->y);
            __builtin_nontemporal_store(b.z, &o1->z); __builtin_nontemporal_store(b.w, &o1->w);
        }
    }
}

// Random-access ceilings for the hash-table and permutation kernels (K4/K5/K6): every lane touches one 8-byte word at a
// pseudo-random place of a table of 2^k words (a bijective xorshift-multiply mix of the lane's index: every word is hit
// exactly once, like a permutation).  mode 6: scatter (store), 7: gather (load), 8: atomicMin on the word (K4's insert).
__device__ __forceinline__ uint64_t mix_bits(uint64_t x, int k) {
    const uint64_t mask = (k >= 64) ? ~0ull : ((1ull << k) - 1);
    const int s = (k + 1) / 2;
    x ^= x >> s;
    x = (x * 0x9e3779b97f4a7c15ull) & mask;
    x ^= x >> s;
    x = (x * 0xd6e8feb86659fd93ull) & mask;
    x ^= x >> s;
    return x;
}

template <int MODE>
__global__ __launch_bounds__(K0_BLOCK) void k0_random(unsigned long long *__restrict__ table, int k, unsigned long long *__restrict__ sink) {
    const int64_t n = (int64_t)1 << k;
    const int64_t stride = (int64_t)gridDim.x * K0_BLOCK;
    unsigned long long acc = 0;
    for (int64_t i = (int64_t)blockIdx.x * K0_BLOCK + threadIdx.x; i < n; i += stride) {
        const uint64_t j = mix_bits((uint64_t)i, k);
        if (MODE == 6) table[j] = (unsigned long long)i;
        else if (MODE == 7) acc ^= table[j];
        else if (MODE == 8) atomicMin(&table[j], (unsigned long long)i);
        else reinterpret_cast<unsigned int *>(table)[j] = (unsigned int)i;   // mode 9: 4-byte scatter over 2^k 4-byte words
    }
    if (MODE == 7 && acc == 0x9e3779b97f4a7c15ull) sink[0] = acc;
}

}  // namespace dyd

using namespace dyd;

extern "C" int dyd_membench_dev(int mode, const void *src, void *dst, int64_t bytes, int blocks, void *stream) {
    DYD_API_ENTER();
    DYD_REQUIRE(mode >= 0 && mode <= 12 && bytes >= 0 && dst, "bad membench arguments");
    if (mode >= 10) {   // the 5 : 2 read / write mix: `bytes` of src are read, 2/5 of that written to dst
        const int64_t n16 = bytes / 16;
        if (n16 < 5) return DYD_OK;
        DYD_REQUIRE(src != nullptr, "bad membench arguments");
        if (blocks <= 0) blocks = ctx().num_cu * 8;
        hipStream_t st = pick_stream(stream);
        const uint4 *s = static_cast<const uint4 *>(src);
        uint4 *d = static_cast<uint4 *>(dst);
        if (mode == 10) hipLaunchKernelGGL(k0_mix<10>, dim3(blocks), dim3(K0_BLOCK), 0, st, s, d, n16);
        else if (mode == 11) hipLaunchKernelGGL(k0_mix<11>, dim3(blocks), dim3(K0_BLOCK), 0, st, s, d, n16);
        else hipLaunchKernelGGL(k0_mix<12>, dim3(blocks), dim3(K0_BLOCK), 0, st, s, d, n16);
        DYD_HIP(hipGetLastError());
        return DYD_OK;
    }
    if (mode >= 6) {   // random access over the largest power-of-two number of 8-byte words in `bytes` of dst
        int k = 0;
        const int64_t word = (mode == 9) ? 4 : 8;
        while ((word << (k + 1)) <= bytes) ++k;
        DYD_REQUIRE(bytes >= 8, "bad membench arguments");
        if (blocks <= 0) blocks = ctx().num_cu * 8;
        hipStream_t st = pick_stream(stream);
        unsigned long long *t = static_cast<unsigned long long *>(dst);
        unsigned long long *sink = const_cast<unsigned long long *>(static_cast<const unsigned long long *>(src ? src : dst));
        if (mode == 6) hipLaunchKernelGGL(k0_random<6>, dim3(blocks), dim3(K0_BLOCK), 0, st, t, k, sink);
        else if (mode == 7) hipLaunchKernelGGL(k0_random<7>, dim3(blocks), dim3(K0_BLOCK), 0, st, t, k, sink);
        else if (mode == 8) hipLaunchKernelGGL(k0_random<8>, dim3(blocks), dim3(K0_BLOCK), 0, st, t, k, sink);
        else hipLaunchKernelGGL(k0_random<9>, dim3(blocks), dim3(K0_BLOCK), 0, st, t, k, sink);
        DYD_HIP(hipGetLastError());
        return DYD_OK;
    }
    const int64_t n16 = bytes / 16;
    if (n16 == 0) return DYD_OK;
    if (blocks <= 0) blocks = ctx().num_cu * 8;
    hipStream_t st = pick_stream(stream);
    const uint4 *s = static_cast<const uint4 *>(src);
    uint4 *d = static_cast<uint4 *>(dst);
    if (mode == 0) hipLaunchKernelGGL(k0_stream<0>, dim3(blocks), dim3(K0_BLOCK), 0, st, s, d, n16);
    else if (mode == 1) hipLaunchKernelGGL(k0_stream<1>, dim3(blocks), dim3(K0_BLOCK), 0, st, s, d, n16);
    else if (mode == 2) hipLaunchKernelGGL(k0_stream<2>, dim3(blocks), dim3(K0_BLOCK), 0, st, s, d, n16);
    else if (mode == 3) hipLaunchKernelGGL(k0_stream<3>, dim3(blocks), dim3(K0_BLOCK), 0, st, s, d, n16);
    else if (mode == 4) hipLaunchKernelGGL(k0_stream<4>, dim3(blocks), dim3(K0_BLOCK), 0, st, s, d, n16);
    else hipLaunchKernelGGL(k0_stream<5>, dim3(blocks), dim3(K0_BLOCK), 0, st, s, d, n16);
    DYD_HIP(hipGetLastError());
    return DYD_OK;
}
