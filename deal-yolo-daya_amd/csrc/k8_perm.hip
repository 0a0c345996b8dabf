// k8_perm.hip — K8: numpy's legacy RandomState(seed).permutation(n) on the device, in parallel, bit for bit.
//
// Replaces the shuffle inside DataFrame.sample(frac=1, random_state=seed) (reference core/processor.py:800): MT19937 seeded
// by init_genrand, then a reversed Fisher-Yates whose partner for step i = n-1 .. 1 is a 32-bit output masked to the next
// 2^k-1 >= i and REJECTED while it exceeds i.  Written as a loop it is sequential three times over — the generator's
// recurrence, the rejection (which draw belongs to which step depends on every earlier rejection) and the swap chain (165 M
// dependent cache misses on the host: 1.4 s for configs[2]'s two categories).  None of the three needs to be:
//
//   stream    MT19937 regenerates its 624-word state in place, but every new word is an XOR of at most three OLD-state
//             terms plus old-state twists (s'[k] = F(k)^s[k+397] | F(k)^F(k-227)^s[k+170] | F(k)^F(k-227)^F(k-454)^s[k-57],
//             F(k) = twist(s[k], s[k+1])), so one workgroup produces a whole block per barrier: k8_mt_stream;
//   resolve   c(t) = number of accepted draws before draw t obeys c(t+1) = c(t) + [ (d[t] & mask(i)) <= i ], i = n-1-c(t).
//             Iterating c <- scan(flags(c)) from an analytic first guess converges to THE sequential solution (the correct
//             prefix grows every round; in practice the error falls like (t/mask)^r / r!): a handful of device-wide scans;
//   shuffle   the FINAL position of every value follows from the partners H[] alone.  Slot j is touched by the steps that
//             target it, L_j = {i : H[i] = j} (all >= j), and by its own step j, which carries its content on to H[j]; after
//             step j the slot is final.  So value v either is fetched by the first step that targets slot v before step v
//             runs (max(L_v) > v: final position max(L_v)), or rides its own step to slot H[v], where the next smaller
//             member of L_{H[v]} fetches it for good — or, if there is none, that slot's own step carries it one hop
//             further, and so on (a hop survives with probability ~1/2: chains are a few hops long).  One stable radix sort
//             of (H[i], i) lays every L_j out in order; "next smaller member" is then the neighbour in the sorted array.
//
// K6 wants exactly this INVERSE (shuffled position of the record with in-category rank v), so its former permutation
// inversion — a 165 M-word random scatter, 64 % of K6 — disappears with the host loop.
#include <cmath>
#include <cstring>
#include <iterator>
#include <string>
#include <thread>
#include <vector>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/functional.hpp>
#include <rocprim/iterator/counting_iterator.hpp>
#include <rocprim/iterator/transform_iterator.hpp>

#include "dyd_common.h"

#define K8_JUMP_QUALIFIER __device__ const
#include "k8_jump_table.h"

namespace dyd {

constexpr int K8_MT_N = 624;
constexpr int K8_MT_THREADS = 640;   // 10 waves, lanes 624..639 idle
constexpr size_t K8_HEAD_BYTES = (((size_t)34 * K8_MT_N * 4) + 4095) & ~(size_t)4095;   // seeded state + the 33 head blocks, untempered

__device__ __forceinline__ uint32_t k8_twist(uint32_t u, uint32_t v) {
    const uint32_t y = (u & 0x80000000u) | (v & 0x7fffffffu);
    return (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
}
__device__ __forceinline__ uint32_t k8_temper(uint32_t y) {
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
}

__device__ __forceinline__ uint32_t k8_untemper(uint32_t y) {
    y ^= y >> 18;
    y ^= (y << 15) & 0xefc60000u;
    uint32_t t = y;
    for (int r = 0; r < 4; ++r) t = y ^ ((t << 7) & 0x9d2c5680u);
    y = t;
    t = y;
    for (int r = 0; r < 2; ++r) t = y ^ (t >> 11);
    return t;
}

// One block step of the generator for the workgroup's thread k: the new state word from the OLD state (see the header).
__device__ __forceinline__ uint32_t k8_next_word(const uint32_t *s, int k) {
    if (k < 227) return k8_twist(s[k], s[k + 1]) ^ s[k + 397];
    if (k < 454) return k8_twist(s[k], s[k + 1]) ^ k8_twist(s[k - 227], s[k - 226]) ^ s[k + 170];
    if (k < 623) return k8_twist(s[k], s[k + 1]) ^ k8_twist(s[k - 227], s[k - 226]) ^ k8_twist(s[k - 454], s[k - 453]) ^ s[k - 57];
    const uint32_t n0 = k8_twist(s[0], s[1]) ^ s[397];   // k == 623 twists with the NEW word 0
    return k8_twist(s[623], n0) ^ k8_twist(s[396], s[397]) ^ k8_twist(s[169], s[170]) ^ s[566];
}

// raw word x[m] of the sequence: the seeded state for m < 624, else the untempered draw m - 624 (already generated)
__device__ __forceinline__ uint32_t k8_raw(const uint32_t *__restrict__ seeded, const uint32_t *__restrict__ out, int64_t m) {
    return m < K8_MT_N ? seeded[m] : k8_untemper(out[m - K8_MT_N]);
}

// The stream in parallel.  MT19937's raw words obey a linear recurrence over GF(2), so the state J words ahead is a fixed,
// seed-independent XOR-combination of the words that follow the current state: x[J + k] = XOR_{i in g_J} x[i + k] with
// g_J = x^J mod the characteristic polynomial (k8_jump_table.h, made by tools/make_mt_jump_table.py and checked there against
// numpy).  Workgroup w of pass p therefore builds the state of chunk c = p*W + w directly — from the first 19937 + 624 words of
// the sequence with g_{cJ} in pass 0, from the first words of chunk c - W with g_{WJ} afterwards — and generates its own
// K8_JUMP_BLOCKS blocks, a block of 624 words per barrier.  `head` runs first, as one workgroup, to put those first words there.
__global__ __launch_bounds__(K8_MT_THREADS) void k8_mt_stream(uint32_t seed, int pass, int head, int64_t total_blocks,
                                                              uint32_t *__restrict__ seeded, uint32_t *__restrict__ out) {
    __shared__ uint32_t st[2][K8_MT_N];
    const int k = threadIdx.x;
    const int64_t chunk = head ? 0 : (int64_t)pass * K8_JUMP_W + blockIdx.x;
    const int64_t b0 = chunk * K8_JUMP_BLOCKS;                 // first block of the chunk
    int64_t nb = head ? 33 : K8_JUMP_BLOCKS;                   // blocks to generate
    if (b0 + nb > total_blocks) nb = total_blocks - b0;
    if (nb <= 0) return;
    if (chunk == 0) {
        if (k == 0) {   // init_genrand
            uint32_t s = seed;
            st[0][0] = s;
            for (int i = 1; i < K8_MT_N; ++i) {
                s = 1812433253u * (s ^ (s >> 30)) + (uint32_t)i;
                st[0][i] = s;
            }
        }
        __syncthreads();
        if (head && k < K8_MT_N) seeded[k] = st[0][k];
    } else {
        const uint32_t *g = k8_jump_table[pass == 0 ? chunk - 1 : K8_JUMP_W - 1];
        const int64_t base = (pass == 0 ? 0 : (chunk - K8_JUMP_W) * (int64_t)K8_JUMP_BLOCKS * K8_MT_N) + k;
        uint32_t acc = 0;
        if (k < K8_MT_N)
            for (int wd = 0; wd < K8_MT_N; ++wd) {
                uint32_t bits = g[wd];   // the same word for every lane
                if (pass == 0) {   // the first 19937 + 624 raw words, kept untempered behind the stream by `head`
                    while (bits) {
                        const int bit = __builtin_ctz(bits);
                        bits &= bits - 1;
                        acc ^= seeded[base + wd * 32 + bit];
                    }
                } else {
                    while (bits) {
                        const int bit = __builtin_ctz(bits);
                        bits &= bits - 1;
                        acc ^= k8_raw(seeded, out, base + wd * 32 + bit);
                    }
                }
            }
        if (k < K8_MT_N) st[0][k] = acc;
        __syncthreads();
    }
    int cur = 0;
    uint32_t *dst = out + b0 * K8_MT_N;
    for (int64_t b = 0; b < nb; ++b) {
        if (k < K8_MT_N) {
            const uint32_t v = k8_next_word(st[cur], k);
            st[cur ^ 1][k] = v;
            dst[b * K8_MT_N + k] = k8_temper(v);
            if (head) seeded[(b + 1) * K8_MT_N + k] = v;   // raw copy of the head's 33 blocks: what every jump of pass 0 reads
        }
        __syncthreads();
        cur ^= 1;
    }
}

__device__ __forceinline__ uint32_t k8_mask(uint32_t i) {   // smallest 2^k - 1 >= i
    return i ? (0xffffffffu >> __builtin_clz(i)) : 0u;
}

// is draw t accepted, given that c draws before it were?  (steps still to do: i = n-1-c down to 1)
struct K8Flag {
    const uint32_t *d;
    const uint32_t *c_prev;
    uint32_t n;
    __device__ __forceinline__ uint32_t operator()(uint32_t t) const {
        const uint32_t c = c_prev[t];
        if (c >= n - 1u) return 0u;
        const uint32_t i = n - 1u - c;
        return ((d[t] & k8_mask(i)) <= i) ? 1u : 0u;
    }
};

// Output side of a resolve round: stores the new count and, where it differs from the previous round's, records the position
// (atomicMin) — the "first difference" search rides on the scan's own store instead of costing a pass of its own.
struct K8DiffOut {
    uint32_t *dst;
    const uint32_t *old;
    uint32_t *first;
    uint32_t base;
    struct Ref {
        uint32_t *p;
        const uint32_t *o;
        uint32_t *first;
        uint32_t idx;
        __device__ __forceinline__ Ref &operator=(uint32_t v) {
            // almost every count differs in the first rounds: only a position below the minimum seen so far goes to the atomic
            if (*o != v && idx < __hip_atomic_load(first, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(first, idx);
            *p = v;
            return *this;
        }
    };
    using iterator_category = std::random_access_iterator_tag;
    using value_type = uint32_t;
    using difference_type = std::ptrdiff_t;
    using pointer = uint32_t *;
    using reference = Ref;
    __host__ __device__ Ref operator[](difference_type i) const { return Ref{dst + i, old + i, first, base + (uint32_t)i}; }
    __host__ __device__ Ref operator*() const { return (*this)[0]; }
    __host__ __device__ K8DiffOut operator+(difference_type i) const { return K8DiffOut{dst + i, old + i, first, base + (uint32_t)i}; }
    __host__ __device__ K8DiffOut &operator+=(difference_type i) { dst += i; old += i; base += (uint32_t)i; return *this; }
};

// ---- the banded resolve ----------------------------------------------------------------------------------------------------
// The analytic guess c0(t) is off by the fluctuation of the rejection count, a few standard deviations of sqrt(t - c0(t)) at most.
// For every count inside the band c0(t) +- K(t) most draws are decided the same way: accepted even at the band's smallest step
// index, or rejected even at its largest.  Only the draws whose value falls between the two (2K/mask of them: 1-3 % for
// n >= 2^22) or whose band straddles an octave boundary depend on the exact count.  So: one scan counts the certain acceptances
// (base), the uncertain draws are compacted into a list, the Picard rounds run on THAT list (count = base + accepted uncertain
// draws before: the same recurrence, two orders of magnitude shorter), their outcomes are scattered back as a byte per draw,
// and one more scan yields the exact counts — which are checked against the band they were derived under (a count outside it
// anywhere voids the result and the full-length rounds run instead).
constexpr float K8_BAND_SIGMAS = 5.0f;
__device__ __forceinline__ uint32_t k8_band(uint32_t t, uint32_t c0) {
    const float r = (float)(t > c0 ? t - c0 : 0u);   // expected rejections so far
    return (uint32_t)(K8_BAND_SIGMAS * sqrtf(r + 16.0f) + 8.0f);
}
// 0 = rejected, 1 = accepted for every count in the band, 2 = depends on the count
__device__ __forceinline__ uint32_t k8_classify(uint32_t dt, uint32_t t, uint32_t c0, uint32_t n) {
    const uint32_t K = k8_band(t, c0);
    const uint32_t lo = c0 > K ? c0 - K : 0u, hi = c0 + K;
    if (lo >= n - 1u) return 0u;           // every step is done
    if (hi >= n - 1u) return 2u;
    const uint32_t i_hi = n - 1u - lo, i_lo = n - 1u - hi;
    const uint32_t m = k8_mask(i_hi);
    if (m != k8_mask(i_lo)) return 2u;     // an octave boundary inside the band
    const uint32_t v = dt & m;
    return v <= i_lo ? 1u : (v > i_hi ? 0u : 2u);
}
// one scan instead of a scan and a compaction: the certain acceptances in the low word, the uncertain draws in the high word; the
// output side stores the base count of every draw and, for an uncertain draw, its position at its rank in the list
struct K8Packed {
    const uint32_t *d, *c0;
    uint32_t n;
    __device__ __forceinline__ unsigned long long operator()(uint32_t t) const {
        const uint32_t k = k8_classify(d[t], t, c0[t], n);
        return k == 1u ? 1ull : (k == 2u ? (1ull << 32) : 0ull);
    }
};
struct K8SplitOut {
    uint32_t *base, *pos_u;
    const uint32_t *d, *c0;
    uint32_t *total;        // [0] = number of uncertain draws (written by the last draw's store)
    uint32_t n, cap, last_t, t0;
    struct Ref {
        uint32_t *base, *pos_u;
        const uint32_t *d, *c0;
        uint32_t *total;
        uint32_t n, cap, last_t, t;
        __device__ __forceinline__ Ref &operator=(unsigned long long v) {
            base[t] = (uint32_t)v;
            const uint32_t rank = (uint32_t)(v >> 32);
            const bool unc = k8_classify(d[t], t, c0[t], n) == 2u;
            if (unc && rank < cap) pos_u[rank] = t;
            if (t == last_t) *total = rank + (unc ? 1u : 0u);
            return *this;
        }
    };
    using iterator_category = std::random_access_iterator_tag;
    using value_type = unsigned long long;
    using difference_type = std::ptrdiff_t;
    using pointer = unsigned long long *;
    using reference = Ref;
    __host__ __device__ Ref operator[](difference_type i) const { return Ref{base, pos_u, d, c0, total, n, cap, last_t, t0 + (uint32_t)i}; }
    __host__ __device__ Ref operator*() const { return (*this)[0]; }
    __host__ __device__ K8SplitOut operator+(difference_type i) const { K8SplitOut r = *this; r.t0 += (uint32_t)i; return r; }
    __host__ __device__ K8SplitOut &operator+=(difference_type i) { t0 += (uint32_t)i; return *this; }
};
struct K8FlagU {           // the recurrence on the list of uncertain draws: count = base + accepted uncertain draws before
    const uint32_t *d, *pos, *bu, *cnt_prev;
    uint32_t n;
    __device__ __forceinline__ uint32_t operator()(uint32_t u) const {
        const uint32_t c = bu[u] + cnt_prev[u];
        if (c >= n - 1u) return 0u;
        const uint32_t i = n - 1u - c;
        return ((d[pos[u]] & k8_mask(i)) <= i) ? 1u : 0u;
    }
};
struct K8FlagFinal {       // scan input of the last pass: certain outcomes from the band, the others from the byte array
    const uint32_t *d, *c0;
    const uint8_t *mark;
    uint32_t n;
    __device__ __forceinline__ uint32_t operator()(uint32_t t) const {
        const uint32_t k = k8_classify(d[t], t, c0[t], n);
        return k == 2u ? (uint32_t)mark[t] : k;
    }
};
// output of the last pass: the exact count, checked against the band it was derived under — and, since the count says which step the
// draw belongs to, the partner of that step straight away (what k8_partners does for the full-length rounds)
struct K8BandOut {
    uint32_t *dst;
    const uint32_t *c0, *d;
    uint32_t *key, *violated;
    uint32_t n, base;
    struct Ref {
        uint32_t *p, *key, *violated;
        uint32_t c0v, dv, n, t;
        __device__ __forceinline__ Ref &operator=(uint32_t v) {
            const uint32_t K = k8_band(t, c0v);
            if ((v > c0v ? v - c0v : c0v - v) > K) *violated = 1u;
            *p = v;
            if (t == 0u) key[0] = 0u;   // step 0 does not exist: a no-op
            if (v < n - 1u) {
                const uint32_t i = n - 1u - v, h = dv & k8_mask(i);
                if (h <= i) key[i] = h;
            }
            return *this;
        }
    };
    using iterator_category = std::random_access_iterator_tag;
    using value_type = uint32_t;
    using difference_type = std::ptrdiff_t;
    using pointer = uint32_t *;
    using reference = Ref;
    __host__ __device__ Ref operator[](difference_type i) const { return Ref{dst + i, key, violated, c0[i], d[i], n, base + (uint32_t)i}; }
    __host__ __device__ Ref operator*() const { return (*this)[0]; }
    __host__ __device__ K8BandOut operator+(difference_type i) const { return K8BandOut{dst + i, c0 + i, d + i, key, violated, n, base + (uint32_t)i}; }
    __host__ __device__ K8BandOut &operator+=(difference_type i) { dst += i; c0 += i; d += i; base += (uint32_t)i; return *this; }
};
// ---- the recurrence on the list of uncertain draws, in ONE launch ------------------------------------------------------------
// Round 2 iterated c <- scan(flags(c)) over the whole list (~1.2 M draws for 82 M steps) with rocPRIM, one 8-byte read-back per
// round to learn how far the exact prefix had grown: 45-47 rounds of ~90 us each per category, a fifth of K8.  The recurrence
// is sequential only from tile to tile: a single workgroup walks the list once, tile after tile of 4096 draws, carrying the EXACT
// count into each tile and iterating inside the tile until no flag changes — the counts inside a tile are off by at most the
// tile's own length at the start, far inside the band the draws were selected with, so three to five local rounds settle it
// (each one a ballot, a 16-entry LDS scan and two barriers).  No host round trip, no second workgroup to wait for.
constexpr int K8L_THREADS = 1024, K8L_EPT = 4, K8L_TILE = K8L_THREADS * K8L_EPT;

// dvu[u] = the draw's word, bu[u] = certain acceptances before it
__global__ __launch_bounds__(256) void k8_list_init(const uint32_t *__restrict__ pos, const uint32_t *__restrict__ d,
                                                    const uint32_t *__restrict__ base, uint32_t n_u, uint32_t *__restrict__ dvu,
                                                    uint32_t *__restrict__ bu) {
    const uint32_t u = blockIdx.x * 256u + threadIdx.x;
    if (u >= n_u) return;
    const uint32_t t = pos[u];
    dvu[u] = d[t];
    bu[u] = base[t];
}

__device__ __forceinline__ uint32_t k8_accept(uint32_t dv, uint32_t c, uint32_t n) {
    if (c >= n - 1u) return 0u;
    const uint32_t i = n - 1u - c;
    return ((dv & k8_mask(i)) <= i) ? 1u : 0u;
}

__global__ __launch_bounds__(K8L_THREADS) void k8_list_resolve(const uint32_t *__restrict__ dvu, const uint32_t *__restrict__ bu,
                                                               uint32_t n_u, uint32_t n, uint8_t *__restrict__ fu,
                                                               uint32_t *__restrict__ rounds_out) {
    __shared__ uint32_t wsum[K8L_THREADS / 64];
    __shared__ int changed_any;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint32_t c_in = 0;                      // accepted uncertain draws before the tile: exact
    uint32_t rounds = 0;
    for (uint32_t tile = 0; tile < n_u; tile += K8L_TILE) {
        const uint32_t u0 = tile + (uint32_t)tid * K8L_EPT;
        uint32_t dv[K8L_EPT], b[K8L_EPT], f[K8L_EPT];
#pragma unroll
        for (int k = 0; k < K8L_EPT; ++k) {
            const bool in = u0 + k < n_u;
            dv[k] = in ? dvu[u0 + k] : 0u;
            b[k] = in ? bu[u0 + k] : 0xffffffffu;             // beyond the list: never accepted (count >= n - 1)
            f[k] = in ? k8_accept(dv[k], b[k] + c_in, n) : 0u; // first guess: nothing inside the tile counted yet
        }
        uint32_t total = 0;
        for (;;) {
            uint32_t s = 0;
#pragma unroll
            for (int k = 0; k < K8L_EPT; ++k) s += f[k];
            uint32_t incl = s;                                  // inclusive scan of the threads' sums inside the wave
#pragma unroll
            for (int dlt = 1; dlt < 64; dlt <<= 1) {
                const uint32_t o = (uint32_t)__shfl_up((int)incl, dlt);
                if (lane >= dlt) incl += o;
            }
            if (tid == 0) changed_any = 0;
            if (lane == 63) wsum[wave] = incl;
            __syncthreads();
            uint32_t before = 0, all = 0;
#pragma unroll
            for (int w = 0; w < K8L_THREADS / 64; ++w) {
                const uint32_t v = wsum[w];
                before += w < wave ? v : 0u;
                all += v;
            }
            uint32_t c = c_in + before + incl - s;              // accepted uncertain draws before this thread's first element
            bool changed = false;
#pragma unroll
            for (int k = 0; k < K8L_EPT; ++k) {
                const uint32_t nf = (b[k] != 0xffffffffu) ? k8_accept(dv[k], b[k] + c, n) : 0u;
                changed |= nf != f[k];
                c += f[k];                                      // the counts of this round are those of the OLD flags (Picard)
                f[k] = nf;
            }
            if (changed) changed_any = 1;
            ++rounds;
            __syncthreads();
            const int again = changed_any;
            total = all;
            __syncthreads();                                    // changed_any is reset by thread 0 at the top of the next round
            if (!again) break;                                  // the flags reproduced themselves: `all` is their sum
        }
#pragma unroll
        for (int k = 0; k < K8L_EPT; ++k)
            if (u0 + k < n_u) fu[u0 + k] = (uint8_t)f[k];
        c_in += total;
    }
    if (tid == 0 && rounds_out) *rounds_out = rounds;
}

// the outcome of every uncertain draw, as a byte at the draw's place
__global__ __launch_bounds__(256) void k8_list_mark(const uint32_t *__restrict__ pos, const uint8_t *__restrict__ fu, uint32_t n_u,
                                                    uint8_t *__restrict__ mark) {
    const uint32_t u = blockIdx.x * 256u + threadIdx.x;
    if (u >= n_u) return;
    mark[pos[u]] = fu[u];
}

// first guess of c(t): inside an octave of mask+1 = M the step index decays like i+1 ~ (i_s+1) exp(-(t-t_s)/M)
struct K8Octaves {
    int count;
    double t_s[34], i_s1[34], m[34];   // octave k starts at draw t_s with i+1 = i_s1
};
__global__ __launch_bounds__(256) void k8_guess(K8Octaves oc, uint32_t n, int64_t n_draws, uint32_t *__restrict__ c0) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= n_draws) return;
    int k = 0;
    while (k + 1 < oc.count && (double)t >= oc.t_s[k + 1]) ++k;
    double i1 = oc.i_s1[k] * exp(-((double)t - oc.t_s[k]) / oc.m[k]);
    if (i1 < 1.0) i1 = 1.0;
    double c = (double)n - i1;
    if (c < 0.0) c = 0.0;
    if (c > (double)(n - 1u)) c = (double)(n - 1u);
    c0[t] = (uint32_t)c;
}

// first index in [lo, n) where the two count arrays differ (0xffffffff: none) — everything before it is final
__global__ __launch_bounds__(256) void k8_first_diff(const uint32_t *__restrict__ a, const uint32_t *__restrict__ b, int64_t lo, int64_t n,
                                                     uint32_t *__restrict__ res) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    uint32_t first = 0xffffffffu;
    for (int64_t t = lo + (int64_t)blockIdx.x * 256 + threadIdx.x; t < n; t += stride)
        if (a[t] != b[t]) { first = (uint32_t)t; break; }
    for (int d = 32; d >= 1; d >>= 1) {
        const uint32_t o = (uint32_t)__shfl_down((int)first, d);
        first = o < first ? o : first;
    }
    if ((threadIdx.x & 63) == 0 && first != 0xffffffffu) atomicMin(res, first);
}
// res[1] = the (exact) count at that index: the next round's scan starts there
__global__ void k8_pick(const uint32_t *__restrict__ c_new, uint32_t *__restrict__ res) {
    if (threadIdx.x == 0) res[1] = (res[0] != 0xffffffffu) ? c_new[res[0]] : 0u;
}

// partners: key[i] = H[i] for the step every accepted draw belongs to; key[0] = 0 (step 0 does not exist: a no-op)
__global__ __launch_bounds__(256) void k8_partners(const uint32_t *__restrict__ d, const uint32_t *__restrict__ c, uint32_t n,
                                                   int64_t lo, int64_t hi, uint32_t *__restrict__ key) {
    const int64_t t = lo + (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t == 0) key[0] = 0u;
    if (t >= hi) return;
    const uint32_t ct = c[t];
    if (ct >= n - 1u) return;
    const uint32_t i = n - 1u - ct;
    const uint32_t v = d[t] & k8_mask(i);
    if (v <= i) key[i] = v;
}

// From the sorted (slot, step) pairs to what the chase needs, indexed by STEP: prev[i] = the next smaller step that targets the
// same slot as step i (0xffffffff: none).  And the values that need no chase at all: the last pair of a slot's run is the largest
// step i that targets slot x — if i > x, value x is fetched by that step before its own step runs and stays at position i
// (inv[x] = i; inv must come in filled with 0xffffffff).
__global__ __launch_bounds__(256) void k8_links(const uint32_t *__restrict__ hs, const uint32_t *__restrict__ is, uint32_t n,
                                                uint32_t *__restrict__ prev, uint32_t *__restrict__ inv) {
    const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (k >= n) return;
    const uint32_t h = hs[k], i = is[k];
    prev[i] = (k > 0 && hs[k - 1] == h) ? is[k - 1] : 0xffffffffu;
    if ((k + 1 == n || hs[k + 1] != h) && i > h) inv[h] = i;
}

// inv[v] = final position of value v (see the header) for the values k8_links left open; optionally inv64 / perm64[inv[v]] = v.
// Value v rides its own step T = v to slot key[T], where the next smaller step targeting that slot (prev[T]) fetches it for good —
// or, if there is none, that slot's own step carries it one hop further.  One thread per VALUE: the first hop reads prev[v] and
// key[v] in order, only the later hops (half as many each time) are random accesses.
__global__ __launch_bounds__(256) void k8_inverse(const uint32_t *__restrict__ key, const uint32_t *__restrict__ prev, uint32_t n,
                                                  uint32_t *__restrict__ inv32, int64_t *__restrict__ inv64,
                                                  int64_t *__restrict__ perm64) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= n) return;
    const uint32_t v = (uint32_t)t;
    uint32_t where = inv32[v];
    if (where == 0xffffffffu) {
        uint32_t T = v;
        while (true) {
            const uint32_t q = prev[T];
            if (q != 0xffffffffu) { where = q; break; }     // the next smaller step targeting the slot it sits in
            const uint32_t slot = key[T];
            if (slot == T) { where = T; break; }            // H[T] == T: it never left
            T = slot;                                       // the slot's own step carries it on
        }
        inv32[v] = where;
    }
    if (inv64) inv64[v] = (int64_t)where;
    if (perm64) perm64[where] = (int64_t)v;
}

__global__ __launch_bounds__(256) void k8_identity(uint32_t n, uint32_t *inv32, int64_t *inv64, int64_t *perm64) {
    const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (k >= n) return;
    if (inv32) inv32[k] = (uint32_t)k;
    if (inv64) inv64[k] = k;
    if (perm64) perm64[k] = k;
}

// expected number of draws of a permutation of n (continuous model) and the octave table of the first guess
static double expected_draws(uint32_t n, K8Octaves *oc) {
    double t = 0.0;
    int cnt = 0;
    uint32_t i = n - 1;   // first step
    while (i >= 1) {
        uint32_t mask = i;
        mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
        const double M = (double)mask + 1.0;
        const uint32_t lo = (mask >> 1) + 1;   // the octave's last step: 2^(k-1)
        if (oc && cnt < 34) { oc->t_s[cnt] = t; oc->i_s1[cnt] = (double)i + 1.0; oc->m[cnt] = M; ++cnt; }
        t += M * log(((double)i + 1.5) / ((double)lo + 0.5));   // sum_{j=lo}^{i} M/(j+1)
        if (lo <= 1) break;
        i = lo - 1;
    }
    if (oc) oc->count = cnt;
    return t;
}

// The (partner, step) sort: keys of at most 30 bits.  rocprim's gfx950 default takes 8 bits per pass — four passes for the 27 bits of
// 82.5 M records, the last one for 3 bits; 9 bits per pass make it three.  Block size and items per thread do not matter (round 3
// sweep of <1024,8> / <1024,12> / <1024,16> / <512,8> / <512,16> at 9 bits, <1024,8> at 8 bits and rocprim's default: K8 + K6
// 16.4-18.3 ms, the kept <1024,8> 16.7): the three scatter passes run at ~1 TB/s of pair traffic whatever the tile.
#ifndef K8_SORT_BITS
#define K8_SORT_BITS 9
#endif
using K8SortConfig = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config,
                                                rocprim::radix_sort_onesweep_config<rocprim::kernel_config<1024, 8>, rocprim::kernel_config<1024, 8>,
                                                                                    K8_SORT_BITS, rocprim::block_radix_rank_algorithm::match>>;

struct PermScratch {
    uint32_t *d = nullptr;          // tempered stream
    int64_t n_draws = 0;            // words of d (multiple of 624)
    uint32_t seed = 0;
    bool valid = false;
};

// One permutation of n (2 <= n <= 2^30) from a stream that is already on the device.  Scratch layout inside `work`:
// cA, cB [draws] | key, hs, is, last [n] | rocprim temp.  Returns DYD_ERR_RANGE when the stream was too short (caller
// regenerates a longer one).
static size_t perm_work_bytes(uint32_t n, int64_t draws, size_t *tmp_bytes_out) {
    size_t scan_tmp = 0, sort_tmp = 0;
    K8Flag f{nullptr, nullptr, n};
    auto in = rocprim::make_transform_iterator(rocprim::make_counting_iterator<uint32_t>(0u), f);
    (void)rocprim::exclusive_scan(nullptr, scan_tmp, in, K8DiffOut{nullptr, nullptr, nullptr, 0u}, 0u, (size_t)draws, rocprim::plus<uint32_t>());
    (void)rocprim::radix_sort_pairs<K8SortConfig>(nullptr, sort_tmp, (uint32_t *)nullptr, (uint32_t *)nullptr, rocprim::make_counting_iterator<uint32_t>(0u),
                                    (uint32_t *)nullptr, (size_t)n, 0u, 32u);
    size_t tmp = scan_tmp > sort_tmp ? scan_tmp : sort_tmp;
    {   // the banded resolve: its scans (certain acceptances, final counts, the short recurrence) and the compaction
        size_t t1 = 0, t2 = 0, t3 = 0, t4 = 0;
        const auto cnt = rocprim::make_counting_iterator<uint32_t>(0u);
        (void)rocprim::exclusive_scan(nullptr, t1, rocprim::make_transform_iterator(cnt, K8Packed{nullptr, nullptr, n}),
                                      K8SplitOut{nullptr, nullptr, nullptr, nullptr, nullptr, n, 0u, 0u, 0u}, 0ull, (size_t)draws,
                                      rocprim::plus<unsigned long long>());
        (void)rocprim::exclusive_scan(nullptr, t3, rocprim::make_transform_iterator(cnt, K8FlagFinal{nullptr, nullptr, nullptr, n}),
                                      K8BandOut{nullptr, nullptr, nullptr, nullptr, nullptr, n, 0u}, 0u, (size_t)draws, rocprim::plus<uint32_t>());
        (void)rocprim::exclusive_scan(nullptr, t4, rocprim::make_transform_iterator(cnt, K8FlagU{nullptr, nullptr, nullptr, nullptr, n}),
                                      K8DiffOut{nullptr, nullptr, nullptr, 0u}, 0u, (size_t)draws, rocprim::plus<uint32_t>());
        for (size_t t : {t1, t2, t3, t4}) tmp = t > tmp ? t : tmp;
    }
    tmp = (tmp + 255) & ~(size_t)255;
    if (tmp_bytes_out) *tmp_bytes_out = tmp;
    const size_t a = (((size_t)draws * 4) + 255) & ~(size_t)255, b = (((size_t)n * 4) + 255) & ~(size_t)255;
    return 2 * a + 5 * b + tmp + 256;
}

constexpr uint32_t K8_BAND_MIN_N = 1u << 20;   // below, the uncertain share is large and the full-length rounds are cheap anyway
static bool g_k8_band = true;                    // A/B and test hook (dyd_set_option "k8_band")
void set_k8_band(int v) { g_k8_band = v != 0; }

// The banded resolve (see k8_classify).  On success with *resolved = true the exact counts are in cB and the partners in key.
// The list arrays live in the regions the sort phase uses later: mark | cntB in hs, pos_u in is, bu in last, cntA in pos.
static int resolve_banded(const uint32_t *d, int64_t draws, uint32_t n, uint32_t *cA, uint32_t *cB, uint32_t *hs, uint32_t *is, uint32_t *last,
                          uint32_t *pos, size_t region_bytes, void *tmp, size_t tmp_bytes, uint32_t *res, uint32_t *key, hipStream_t st,
                          int *rounds_out, bool *resolved) {
    *resolved = false;
    const size_t mark_bytes = (((size_t)draws) + 255) & ~(size_t)255;
    if (region_bytes < mark_bytes + 1024) return DYD_OK;
    uint8_t *mark = reinterpret_cast<uint8_t *>(hs);
    uint32_t *cntB0 = reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(hs) + mark_bytes);
    const size_t cap_b = (region_bytes - mark_bytes) / 4, cap = cap_b < (size_t)n ? cap_b : (size_t)n;
    uint32_t *pos_u = is, *bu = last, *cntA0 = pos;
    const auto cnt = rocprim::make_counting_iterator<uint32_t>(0u);
    size_t tb = tmp_bytes;
    // ONE scan: base[t] = certain acceptances before t (into cB) and the uncertain draws, in order, into the list
    uint32_t *n_sel = res + 4;
    DYD_HIP(hipMemsetAsync(n_sel, 0, 4, st));
    DYD_HIP(rocprim::exclusive_scan(tmp, tb, rocprim::make_transform_iterator(cnt, K8Packed{d, cA, n}),
                                    K8SplitOut{cB, pos_u, d, cA, n_sel, n, (uint32_t)cap, (uint32_t)(draws - 1), 0u}, 0ull, (size_t)draws,
                                    rocprim::plus<unsigned long long>(), st));
    uint32_t n_u = 0;
    DYD_HIP(hipMemcpyAsync(&n_u, n_sel, 4, hipMemcpyDeviceToHost, st));
    DYD_HIP(hipStreamSynchronize(st));
    if ((size_t)n_u > cap || (int64_t)n_u >= draws) return DYD_OK;   // not worth it (or no room): the full-length rounds
    DYD_HIP(hipMemsetAsync(mark, 0, (size_t)draws, st));
    int rounds = 0;
    if (n_u) {
        // list arrays: dvu in cntA0's place, the flag bytes in cntB0's (a quarter of it)
        uint32_t *dvu = cntA0;
        uint8_t *fu = reinterpret_cast<uint8_t *>(cntB0);
        hipLaunchKernelGGL(k8_list_init, dim3((unsigned)ceil_div((int64_t)n_u, 256)), dim3(256), 0, st, pos_u, d, cB, n_u, dvu, bu);
        DYD_HIP(hipGetLastError());
        hipLaunchKernelGGL(k8_list_resolve, dim3(1), dim3(K8L_THREADS), 0, st, dvu, bu, n_u, n, fu, res + 5);
        DYD_HIP(hipGetLastError());
        hipLaunchKernelGGL(k8_list_mark, dim3((unsigned)ceil_div((int64_t)n_u, 256)), dim3(256), 0, st, pos_u, fu, n_u, mark);
        DYD_HIP(hipGetLastError());
    }
    // exact counts (into cB, over the base counts that are no longer needed), checked against the band
    DYD_HIP(hipMemsetAsync(res + 2, 0, 4, st));
    tb = tmp_bytes;
    DYD_HIP(rocprim::exclusive_scan(tmp, tb, rocprim::make_transform_iterator(cnt, K8FlagFinal{d, cA, mark, n}), K8BandOut{cB, cA, d, key, res + 2, n, 0u}, 0u,
                                    (size_t)draws, rocprim::plus<uint32_t>(), st));
    uint32_t violated = 1, list_rounds = 0;
    DYD_HIP(hipMemcpyAsync(&violated, res + 2, 4, hipMemcpyDeviceToHost, st));
    if (n_u) DYD_HIP(hipMemcpyAsync(&list_rounds, res + 5, 4, hipMemcpyDeviceToHost, st));
    DYD_HIP(hipStreamSynchronize(st));
    rounds = (int)list_rounds;              // local rounds of the list walk, all tiles together
    if (rounds_out) *rounds_out = rounds;
    if (violated) return DYD_OK;   // a count left its band somewhere: nothing above is trusted (the full-length rounds rewrite every partner)
    *resolved = true;
    return DYD_OK;
}

static int perm_from_stream(const uint32_t *d, int64_t n_draws_avail, uint32_t n, void *work, uint32_t *inv32, int64_t *inv64,
                            int64_t *perm64, hipStream_t st, int *rounds_out) {
    K8Octaves oc;
    const double e = expected_draws(n, &oc);
    int64_t draws = (int64_t)(e * 1.002 + 8.0 * sqrt(2.0 * (double)n) + 4096.0);
    if (draws > n_draws_avail) draws = n_draws_avail;
    size_t tmp_bytes = 0;
    (void)perm_work_bytes(n, draws, &tmp_bytes);
    const size_t a = (((size_t)draws * 4) + 255) & ~(size_t)255, b = (((size_t)n * 4) + 255) & ~(size_t)255;
    char *w = static_cast<char *>(work);
    uint32_t *cA = reinterpret_cast<uint32_t *>(w), *cB = reinterpret_cast<uint32_t *>(w + a);
    uint32_t *key = reinterpret_cast<uint32_t *>(w + 2 * a), *hs = reinterpret_cast<uint32_t *>(w + 2 * a + b);
    uint32_t *is = reinterpret_cast<uint32_t *>(w + 2 * a + 2 * b), *last = reinterpret_cast<uint32_t *>(w + 2 * a + 3 * b);
    uint32_t *pos = reinterpret_cast<uint32_t *>(w + 2 * a + 4 * b);
    void *tmp = w + 2 * a + 5 * b;
    uint32_t *res = reinterpret_cast<uint32_t *>(w + 2 * a + 5 * b + tmp_bytes);
    const unsigned gn = (unsigned)ceil_div((int64_t)n, 256);

    hipLaunchKernelGGL(k8_guess, dim3((unsigned)ceil_div(draws, 256)), dim3(256), 0, st, oc, n, draws, cA);
    DYD_HIP(hipGetLastError());
    int rounds = 0;
    bool resolved = false;
    if (n >= K8_BAND_MIN_N && g_k8_band) {
        const int rc = resolve_banded(d, draws, n, cA, cB, hs, is, last, pos, b, tmp, tmp_bytes, res, key, st, &rounds, &resolved);
        if (rc) return rc;
    }
    // Picard rounds over the not yet final suffix [lo, draws): cB[t] = c_lo + sum of flags(cA) on [lo, t).  Whatever lies before
    // the first difference is final (its flags were computed from exact counts), so the partners of that stretch are written
    // at once and the next round starts there.
    int64_t lo = resolved ? draws : 0;
    uint32_t c_lo = 0;
    while (lo < draws) {
        K8Flag f{d, cA, n};
        auto in = rocprim::make_transform_iterator(rocprim::make_counting_iterator<uint32_t>((uint32_t)lo), f);
        size_t tb = tmp_bytes;
        DYD_HIP(hipMemsetAsync(res, 0xff, 4, st));
        DYD_HIP(rocprim::exclusive_scan(tmp, tb, in, K8DiffOut{cB + lo, cA + lo, res, (uint32_t)lo}, c_lo, (size_t)(draws - lo),
                                        rocprim::plus<uint32_t>(), st));
        hipLaunchKernelGGL(k8_pick, dim3(1), dim3(64), 0, st, cB, res);
        DYD_HIP(hipGetLastError());
        uint32_t host[2] = {0, 0};
        DYD_HIP(hipMemcpyAsync(host, res, 8, hipMemcpyDeviceToHost, st));
        DYD_HIP(hipStreamSynchronize(st));
        ++rounds;
        const int64_t first = (host[0] == 0xffffffffu) ? draws : (int64_t)host[0];
        if (first > lo) {   // cB is exact on [lo, first]: those draws' steps are known
            hipLaunchKernelGGL(k8_partners, dim3((unsigned)ceil_div(first - lo, 256)), dim3(256), 0, st, d, cB, n, lo, first, key);
            DYD_HIP(hipGetLastError());
        }
        if (first >= draws) break;
        lo = first;
        c_lo = host[1];
        uint32_t *t2 = cA; cA = cB; cB = t2;   // cA = the newest counts (exact up to lo, the best guess beyond)
        if (rounds > 100000) { set_error("K8: the rejection resolve did not settle"); return DYD_ERR_HIP; }
    }
    if (rounds_out) *rounds_out = rounds;
    // enough draws?  the last draw's count (+ its own flag) must reach n-1 accepted  (cB holds the final counts of the tail)
    uint32_t c_last = 0, d_last = 0;
    DYD_HIP(hipMemcpyAsync(&c_last, cB + (draws - 1), 4, hipMemcpyDeviceToHost, st));
    DYD_HIP(hipMemcpyAsync(&d_last, d + (draws - 1), 4, hipMemcpyDeviceToHost, st));
    DYD_HIP(hipStreamSynchronize(st));
    {
        uint32_t acc = c_last;
        if (c_last < n - 1u) {
            const uint32_t i = n - 1u - c_last;
            uint32_t mask = i;
            mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
            if ((d_last & mask) <= i) ++acc;
        }
        if (acc < n - 1u) return DYD_ERR_RANGE;
    }
    unsigned bits = 1;
    while (bits < 32 && (1ull << bits) < (unsigned long long)n) ++bits;
    size_t tb = tmp_bytes;
    DYD_HIP(rocprim::radix_sort_pairs<K8SortConfig>(tmp, tb, key, hs, rocprim::make_counting_iterator<uint32_t>(0u), is, (size_t)n, 0u, bits, st));
    uint32_t *invw = inv32 ? inv32 : last;               // the caller's 32-bit inverse, else scratch (the list arrays are done with)
    DYD_HIP(hipMemsetAsync(invw, 0xff, (size_t)n * 4, st));
    hipLaunchKernelGGL(k8_links, dim3(gn), dim3(256), 0, st, hs, is, n, pos, invw);
    DYD_HIP(hipGetLastError());
    hipLaunchKernelGGL(k8_inverse, dim3(gn), dim3(256), 0, st, key, pos, n, invw, inv64, perm64);
    DYD_HIP(hipGetLastError());
    return DYD_OK;
}

static int g_k8_last_rounds = 0;

// permutations of several sizes from ONE seed (every category of the split is shuffled with the same random_state,
// reference :800, so they share the stream).  inv32[c] / inv64[c] / perm64[c] may be null.
int k8_permutations(uint32_t seed, const int64_t *sizes, int n_sizes, uint32_t *const *inv32, int64_t *const *inv64,
                    int64_t *const *perm64, hipStream_t st) {
    int64_t n_max = 0;
    for (int c = 0; c < n_sizes; ++c) {
        if (sizes[c] < 0 || sizes[c] > (1LL << 30)) { set_error("K8: permutation size %lld outside [0, 2^30]", (long long)sizes[c]); return DYD_ERR_RANGE; }
        if (sizes[c] > n_max) n_max = sizes[c];
    }
    for (int c = 0; c < n_sizes; ++c)
        if (sizes[c] <= 1 && sizes[c] > 0) {
            hipLaunchKernelGGL(k8_identity, dim3(1), dim3(256), 0, st, (uint32_t)sizes[c], inv32 ? inv32[c] : nullptr,
                               inv64 ? inv64[c] : nullptr, perm64 ? perm64[c] : nullptr);
            DYD_HIP(hipGetLastError());
        }
    if (n_max <= 1) return DYD_OK;
    double margin = 1.004;
    for (int attempt = 0; attempt < 4; ++attempt, margin *= 1.5) {
        const double e = expected_draws((uint32_t)n_max, nullptr);
        int64_t draws = (int64_t)(e * margin + 16.0 * sqrt(2.0 * (double)n_max) + 8192.0);
        const int64_t n_blocks = ceil_div(draws, K8_MT_N);
        draws = n_blocks * K8_MT_N;
        const size_t d_bytes = ((((size_t)draws * 4) + 255) & ~(size_t)255) + K8_HEAD_BYTES;   // + the seeded state and the head's raw words
        // every category gets its own work area and (below) its own stream and host thread: the resolve is a chain of ~40
        // small dependent launches with an 8-byte read-back each, which leaves the device mostly idle — several categories
        // side by side cost little more than one
        std::vector<size_t> work_off((size_t)n_sizes + 1, 0);
        int n_big = 0;
        for (int c = 0; c < n_sizes; ++c) {
            size_t wb = 0;
            if (sizes[c] > 1) { wb = (perm_work_bytes((uint32_t)sizes[c], draws, nullptr) + 255) & ~(size_t)255; ++n_big; }
            work_off[(size_t)c + 1] = work_off[(size_t)c] + wb;
        }
        const size_t work = work_off[(size_t)n_sizes];
        void *scr = nullptr;
        int rc = get_scratch(d_bytes + work, &scr, st);
        if (rc) return rc;
        uint32_t *d = static_cast<uint32_t *>(scr);
        uint32_t *seeded = reinterpret_cast<uint32_t *>(static_cast<char *>(scr) + d_bytes - K8_HEAD_BYTES);   // 34 x 624 raw words behind the stream
        // the first 33 blocks by one workgroup, then every pass of K8_JUMP_W chunks in parallel (a pass reads the one before)
        hipLaunchKernelGGL(k8_mt_stream, dim3(1), dim3(K8_MT_THREADS), 0, st, seed, 0, 1, n_blocks, seeded, d);
        DYD_HIP(hipGetLastError());
        const int64_t n_chunks = ceil_div(n_blocks, (int64_t)K8_JUMP_BLOCKS);
        for (int64_t p = 0; p * K8_JUMP_W < n_chunks; ++p) {
            const int64_t left = n_chunks - p * K8_JUMP_W;
            hipLaunchKernelGGL(k8_mt_stream, dim3((unsigned)(left < K8_JUMP_W ? left : K8_JUMP_W)), dim3(K8_MT_THREADS), 0, st, seed, (int)p, 0,
                               n_blocks, seeded, d);
            DYD_HIP(hipGetLastError());
        }
        bool short_stream = false;
        char *work_base = static_cast<char *>(scr) + d_bytes;
        if (n_big <= 1) {
            for (int c = 0; c < n_sizes && !short_stream; ++c) {
                if (sizes[c] <= 1) continue;
                rc = perm_from_stream(d, draws, (uint32_t)sizes[c], work_base + work_off[(size_t)c], inv32 ? inv32[c] : nullptr,
                                      inv64 ? inv64[c] : nullptr, perm64 ? perm64[c] : nullptr, st, &g_k8_last_rounds);
                if (rc == DYD_ERR_RANGE) short_stream = true;
                else if (rc) { release_scratch(st); return rc; }
            }
        } else {
            hipEvent_t ready = nullptr;
            DYD_HIP(hipEventCreateWithFlags(&ready, hipEventDisableTiming));
            DYD_HIP(hipEventRecord(ready, st));      // the stream exists; the categories' streams start behind it
            const int device = ctx().device;
            std::vector<int> rcs((size_t)n_sizes, DYD_OK);
            std::vector<std::string> msgs((size_t)n_sizes);
            std::vector<std::thread> th;
            for (int c = 0; c < n_sizes; ++c) {
                if (sizes[c] <= 1) continue;
                th.emplace_back([&, c] {
                    int r = DYD_OK;
                    hipStream_t cs = nullptr;
                    if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&cs, hipStreamNonBlocking) != hipSuccess) {
                        rcs[(size_t)c] = DYD_ERR_HIP;
                        return;
                    }
                    if (hipStreamWaitEvent(cs, ready, 0) != hipSuccess) r = DYD_ERR_HIP;
                    int rounds = 0;
                    if (!r)
                        r = perm_from_stream(d, draws, (uint32_t)sizes[c], work_base + work_off[(size_t)c], inv32 ? inv32[c] : nullptr,
                                             inv64 ? inv64[c] : nullptr, perm64 ? perm64[c] : nullptr, cs, &rounds);
                    if (hipStreamSynchronize(cs) != hipSuccess && !r) r = DYD_ERR_HIP;
                    (void)hipStreamDestroy(cs);
                    if (r && r != DYD_ERR_RANGE) msgs[(size_t)c] = dyd_last_error();   // the message lives in this thread
                    rcs[(size_t)c] = r;
                    g_k8_last_rounds = rounds;
                });
            }
            for (auto &t : th) t.join();
            (void)hipEventDestroy(ready);
            for (int c = 0; c < n_sizes; ++c) {
                if (rcs[(size_t)c] == DYD_ERR_RANGE) short_stream = true;
                else if (rcs[(size_t)c]) {
                    set_error("%s", msgs[(size_t)c].empty() ? "K8: a category's permutation failed" : msgs[(size_t)c].c_str());
                    release_scratch(st);
                    return rcs[(size_t)c];
                }
            }
        }
        release_scratch(st);
        if (!short_stream) return DYD_OK;
    }
    set_error("K8: the generator stream stayed too short");
    return DYD_ERR_HIP;
}

int k8_last_rounds() { return g_k8_last_rounds; }

}  // namespace dyd

using namespace dyd;

extern "C" {

int dyd_mt19937_permutation_dev(uint32_t seed, int64_t n, int64_t *out_perm_or_null, int64_t *out_inverse_or_null, void *stream) {
    DYD_API_ENTER();
    DYD_REQUIRE(n >= 0, "n < 0");
    DYD_REQUIRE(n <= (1LL << 30), "n above 2^30: use dyd_mt19937_permutation");
    if (n == 0 || (!out_perm_or_null && !out_inverse_or_null)) return DYD_OK;
    int64_t *inv = out_inverse_or_null, *perm = out_perm_or_null;
    hipStream_t st = pick_stream(stream);
    const int rc = k8_permutations(seed, &n, 1, nullptr, &inv, &perm, st);
    if (rc) return rc;
    DYD_HIP(hipStreamSynchronize(st));   // "synchronous": the caller may read the arrays on any stream
    return DYD_OK;
}

}  // extern "C"
