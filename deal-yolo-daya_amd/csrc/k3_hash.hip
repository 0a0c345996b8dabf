// k3_hash.hip — K3: 128-bit hash of every cell of a string column.
//
// The equality test hidden inside DataFrame.drop_duplicates (reference core/processor.py:140)
// and Series.isin (:198) — pandas' khash over Python str objects — is replaced by equality of
// 128-bit hashes of the host's canonical byte form of each cell.  Hash = MurmurHash3 x64_128,
// seed 0 (public-domain algorithm by A. Appleby; restated, not copied).  Collision probability
// at 1e8 rows ~1.5e-23.
//
// Layout in HBM: bytes = concatenated cells (sum L bytes), off = N+1 int64 byte offsets,
// out = N x (h1,h2) u64.  Algorithmic bytes per launch: sum L + 8*(N+1) + 16*N.
// Bound: HBM (integer mixing is ~1 op/B).  One lane per cell; cells are short (~30 B) and
// adjacent lanes read adjacent cells, so a wave's loads fall in a ~2 KiB window that the
// vector L1 serves after the first touch.
#include "dyd_common.h"

namespace dyd {

constexpr int K3_BLOCK = 256;

__device__ __forceinline__ uint64_t rotl64(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }
__device__ __forceinline__ uint64_t fmix64(uint64_t k) {
    k ^= k >> 33;
    k *= 0xff51afd7ed558ccdULL;
    k ^= k >> 33;
    k *= 0xc4ceb9fe1a85ec53ULL;
    k ^= k >> 33;
    return k;
}
__device__ __forceinline__ uint64_t load_u64_unaligned(const uint8_t *p) {
    uint64_t v;
    __builtin_memcpy(&v, p, 8);
    return v;
}

__global__ __launch_bounds__(K3_BLOCK) void k3_hash_kernel(const uint8_t *__restrict__ bytes,
                                                           const int64_t *__restrict__ off, int64_t n,
                                                           uint64_t *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * K3_BLOCK + threadIdx.x;
    if (i >= n) return;
    const int64_t s = off[i];
    const int64_t len = off[i + 1] - s;
    const int64_t total_bytes = off[n];
    const uint8_t *p = bytes + s;
    const uint64_t c1 = 0x87c37b91114253d5ULL, c2 = 0x4cf5ad432745937fULL;
    uint64_t h1 = 0, h2 = 0;
    const int64_t nblocks = len >> 4;
    auto mix = [&](uint64_t k1, uint64_t k2) {
        k1 *= c1; k1 = rotl64(k1, 31); k1 *= c2; h1 ^= k1;
        h1 = rotl64(h1, 27); h1 += h2; h1 = h1 * 5 + 0x52dce729;
        k2 *= c2; k2 = rotl64(k2, 33); k2 *= c1; h2 ^= k2;
        h2 = rotl64(h2, 31); h2 += h1; h2 = h2 * 5 + 0x38495ab5;
    };
    int64_t b = 0;
    // long cells: the state chain is sequential but the loads are not — four 16-byte blocks (one 64-byte line) are requested
    // before the first is mixed, so a lane keeps four loads in flight instead of one
    for (; b + 4 <= nblocks; b += 4) {
        const uint8_t *q = p + 16 * b;
        const uint64_t a0 = load_u64_unaligned(q), a1 = load_u64_unaligned(q + 8), b0 = load_u64_unaligned(q + 16),
                       b1 = load_u64_unaligned(q + 24), c0 = load_u64_unaligned(q + 32), c1_ = load_u64_unaligned(q + 40),
                       d0 = load_u64_unaligned(q + 48), d1 = load_u64_unaligned(q + 56);
        mix(a0, a1); mix(b0, b1); mix(c0, c1_); mix(d0, d1);
    }
    for (; b < nblocks; ++b) mix(load_u64_unaligned(p + 16 * b), load_u64_unaligned(p + 16 * b + 8));
    const uint8_t *tail = p + 16 * nblocks;
    const int rem = (int)(len & 15);
    uint64_t k1 = 0, k2 = 0;
    if (s + 16 * nblocks + 16 <= total_bytes) {
        // the 1..15 tail bytes with two 8-byte loads and a mask: the bytes read beyond the cell belong to the next cells (the
        // buffer is known to reach that far), not one byte load per tail byte
        if (rem > 0) {
            k1 = load_u64_unaligned(tail);
            if (rem < 8) k1 &= (1ull << (8 * rem)) - 1;
        }
        if (rem > 8) {
            k2 = load_u64_unaligned(tail + 8);
            k2 &= (1ull << (8 * (rem - 8))) - 1;   // rem - 8 is 1..7
        }
    } else {   // the last cells of the buffer: byte by byte
        for (int t = 0; t < rem; ++t) {
            const uint64_t v = tail[t];
            if (t < 8) k1 |= v << (8 * t);
            else k2 |= v << (8 * (t - 8));
        }
    }
    if (rem > 8) { k2 *= c2; k2 = rotl64(k2, 33); k2 *= c1; h2 ^= k2; }
    if (rem > 0) { k1 *= c1; k1 = rotl64(k1, 31); k1 *= c2; h1 ^= k1; }
    h1 ^= (uint64_t)len;
    h2 ^= (uint64_t)len;
    h1 += h2; h2 += h1;
    h1 = fmix64(h1); h2 = fmix64(h2);
    h1 += h2; h2 += h1;
    *reinterpret_cast<ulonglong2 *>(out + 2 * i) = make_ulonglong2(h1, h2);
}

int launch_k3(const uint8_t *bytes, const int64_t *off, int64_t n, uint64_t *out, hipStream_t st) {
    if (n == 0) return DYD_OK;
    const int64_t blocks = ceil_div(n, K3_BLOCK);
    if (blocks > 0x7fffffffLL) {
        set_error("n=%lld exceeds one launch", (long long)n);
        return DYD_ERR_RANGE;
    }
    hipLaunchKernelGGL(k3_hash_kernel, dim3((unsigned)blocks), dim3(K3_BLOCK), 0, st, bytes, off, n, out);
    DYD_HIP(hipGetLastError());
    return DYD_OK;
}

}  // namespace dyd

using namespace dyd;

extern "C" {

int dyd_hash128_dev(const uint8_t *bytes, const int64_t *off, int64_t n, uint64_t *out_hi_lo, void *stream) {
    DYD_API_ENTER();
    DYD_REQUIRE(n >= 0, "n < 0");
    if (n == 0) return DYD_OK;
    DYD_REQUIRE(off && out_hi_lo, "null pointer");
    DYD_REQUIRE((reinterpret_cast<uintptr_t>(out_hi_lo) & 15) == 0, "out must be 16-byte aligned");
    return launch_k3(bytes, off, n, out_hi_lo, pick_stream(stream));
}

int dyd_hash128(const uint8_t *bytes, const int64_t *off, int64_t n, uint64_t *out_hi_lo) {
    DYD_API_ENTER();
    DYD_REQUIRE(n >= 0, "n < 0");
    if (n == 0) return DYD_OK;
    DYD_REQUIRE(off && out_hi_lo, "null pointer");
    DYD_REQUIRE(off[0] == 0, "off[0] != 0");
    for (int64_t i = 0; i < n; ++i) DYD_REQUIRE(off[i + 1] >= off[i], "off not monotone");
    const int64_t total = off[n];
    DYD_REQUIRE(total == 0 || bytes, "bytes is null");
    DevBuf d_bytes, d_off, d_out;
    int rc;
    if ((rc = d_bytes.alloc((size_t)total + 16)) || (rc = d_off.alloc(8 * (size_t)(n + 1))) ||
        (rc = d_out.alloc(16 * (size_t)n)))
        return rc;
    hipStream_t st = ctx().stream;
    if (total) DYD_HIP(hipMemcpyAsync(d_bytes.p, bytes, (size_t)total, hipMemcpyHostToDevice, st));
    DYD_HIP(hipMemcpyAsync(d_off.p, off, 8 * (size_t)(n + 1), hipMemcpyHostToDevice, st));
    KernelTimer t(st);
    rc = launch_k3(d_bytes.as<uint8_t>(), d_off.as<int64_t>(), n, d_out.as<uint64_t>(), st);
    if (rc) return rc;
    t.finish();
    DYD_HIP(hipMemcpyAsync(out_hi_lo, d_out.p, 16 * (size_t)n, hipMemcpyDeviceToHost, st));
    DYD_HIP(hipStreamSynchronize(st));
    return DYD_OK;
}

}  // extern "C"
