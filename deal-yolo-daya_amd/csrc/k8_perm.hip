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
//             An analytic guess of c(t) is off by a few standard deviations of the rejection count at most, and for every count
//             inside that band most draws are decided the same way: two hand-written single-pass scans (decoupled look-back)
//             settle those, and the 1-3 % whose outcome depends on the exact count are listed and resolved by ONE workgroup
//             that walks the list tile by tile (k8_scan_classify -> k8_list_resolve -> k8_scan_final; "the banded resolve").
//             Short permutations, and a band that did not hold, take full-length rounds c <- scan(flags(c)) (rocPRIM);
//   shuffle   the FINAL position of every value follows from the partners H[] alone.  Slot j is touched by the steps that
//             target it, L_j = {i : H[i] = j} (all >= j), and by its own step j, which carries its content on to H[j]; after
//             step j the slot is final.  So value v either is fetched by the first step that targets slot v before step v
//             runs (max(L_v) > v: final position max(L_v)), or rides its own step to slot H[v], where the next smaller
//             member of L_{H[v]} fetches it for good — or, if there is none, that slot's own step carries it one hop
//             further, and so on (a hop survives with probability ~1/2: chains are a few hops long).  One stable radix sort
//             of (H[i], i) lays every L_j out in order; k8_links turns "next smaller member" into an array indexed by step,
//             so the chase (k8_inverse) reads its first hop in order.
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
// ---- the recurrence on the list of uncertain draws, in ONE launch ------------------------------------------------------------
// Round 2 iterated c <- scan(flags(c)) over the whole list (~1.2 M draws for 82 M steps) with rocPRIM, one 8-byte read-back per
// round to learn how far the exact prefix had grown: 45-47 rounds of ~90 us each per category, a fifth of K8.  The recurrence
// is sequential only from tile to tile: a single workgroup walks the list once, tile after tile of 8192 draws, carrying the EXACT
// count into each tile and iterating inside the tile until no flag changes — the counts inside a tile are off by at most the
// tile's own length at the start, far inside the band the draws were selected with, so three to five local rounds settle it
// (each one a ballot, a 16-entry LDS scan and two barriers).  No host round trip, no second workgroup to wait for.
constexpr int K8L_THREADS = 1024, K8L_EPT = 8, K8L_TILE = K8L_THREADS * K8L_EPT;

__device__ __forceinline__ uint32_t k8_accept(uint32_t dv, uint32_t c, uint32_t n) {
    if (c >= n - 1u) return 0u;
    const uint32_t i = n - 1u - c;
    return ((dv & k8_mask(i)) <= i) ? 1u : 0u;
}

__global__ __launch_bounds__(K8L_THREADS) void k8_list_resolve(const uint32_t *__restrict__ dvu, const uint32_t *__restrict__ bu,
                                                               uint32_t n_u, uint32_t n, uint8_t *__restrict__ fu,
                                                               uint32_t *__restrict__ rounds_out) {
    __shared__ uint32_t wsum[2][K8L_THREADS / 64];
    __shared__ int changed_any[2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint32_t c_in = 0;                      // accepted uncertain draws before the tile: exact
    uint32_t rounds = 0;
    uint32_t rate = 128;                    // accepted share of the last tile in 1/256: the first guess of the counts inside the next
    if (tid < 2) changed_any[tid] = 0;
    // the next tile's words are loaded while the current one is iterated on (a single workgroup has nobody to hide its latency)
    uint32_t ndv[K8L_EPT], nb[K8L_EPT];
    auto load = [&](uint32_t tile, uint32_t *dv, uint32_t *b) {
        const uint32_t u0 = tile + (uint32_t)tid * K8L_EPT;
        if (u0 + K8L_EPT <= n_u) {
            const uint4 a0 = reinterpret_cast<const uint4 *>(dvu + u0)[0], a1 = reinterpret_cast<const uint4 *>(dvu + u0)[1];
            const uint4 b0 = reinterpret_cast<const uint4 *>(bu + u0)[0], b1 = reinterpret_cast<const uint4 *>(bu + u0)[1];
            dv[0] = a0.x; dv[1] = a0.y; dv[2] = a0.z; dv[3] = a0.w; dv[4] = a1.x; dv[5] = a1.y; dv[6] = a1.z; dv[7] = a1.w;
            b[0] = b0.x; b[1] = b0.y; b[2] = b0.z; b[3] = b0.w; b[4] = b1.x; b[5] = b1.y; b[6] = b1.z; b[7] = b1.w;
        } else {
#pragma unroll
            for (int k = 0; k < K8L_EPT; ++k) {
                const bool in = u0 + k < n_u;
                dv[k] = in ? dvu[u0 + k] : 0u;
                b[k] = in ? bu[u0 + k] : 0xffffffffu;         // beyond the list: never accepted
            }
        }
    };
    load(0, ndv, nb);
    __syncthreads();
    for (uint32_t tile = 0; tile < n_u; tile += K8L_TILE) {
        const uint32_t u0 = tile + (uint32_t)tid * K8L_EPT;
        uint32_t dv[K8L_EPT], b[K8L_EPT], f[K8L_EPT];
#pragma unroll
        for (int k = 0; k < K8L_EPT; ++k) { dv[k] = ndv[k]; b[k] = nb[k]; }
        if (tile + K8L_TILE < n_u) load(tile + K8L_TILE, ndv, nb);
        {   // first guess of the count before each of this thread's draws: the last tile's acceptance share, spread evenly
            uint32_t c = c_in + (((uint32_t)tid * K8L_EPT * rate) >> 8);
#pragma unroll
            for (int k = 0; k < K8L_EPT; ++k) {
                f[k] = (b[k] != 0xffffffffu) ? k8_accept(dv[k], b[k] + c, n) : 0u;
                c += f[k];
            }
        }
        uint32_t total = 0;
        for (int par = 0;; par ^= 1) {
            uint32_t s = 0;
#pragma unroll
            for (int k = 0; k < K8L_EPT; ++k) s += f[k];
            uint32_t incl = s;                                  // inclusive scan of the threads' sums inside the wave
#pragma unroll
            for (int dlt = 1; dlt < 64; dlt <<= 1) {
                const uint32_t o = (uint32_t)__shfl_up((int)incl, dlt);
                if (lane >= dlt) incl += o;
            }
            if (lane == 63) wsum[par][wave] = incl;
            __syncthreads();
            uint32_t before = 0, all = 0;
#pragma unroll
            for (int w = 0; w < K8L_THREADS / 64; ++w) {
                const uint32_t v = wsum[par][w];
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
            if (changed) changed_any[par] = 1;
            ++rounds;
            __syncthreads();
            const int again = changed_any[par];
            if (tid == 0) changed_any[par ^ 1] = 0;             // the other parity's flag and sums are free again: two barriers a round
            total = all;
            if (!again) break;                                  // the flags reproduced themselves: `all` is their sum
        }
        if (u0 + K8L_EPT <= n_u) {
            const uint32_t lo = f[0] | (f[1] << 8) | (f[2] << 16) | (f[3] << 24), hi = f[4] | (f[5] << 8) | (f[6] << 16) | (f[7] << 24);
            reinterpret_cast<uint2 *>(fu + u0)[0] = make_uint2(lo, hi);
        } else {
#pragma unroll
            for (int k = 0; k < K8L_EPT; ++k)
                if (u0 + k < n_u) fu[u0 + k] = (uint8_t)f[k];
        }
        c_in += total;
        const uint32_t len = (n_u - tile < (uint32_t)K8L_TILE) ? n_u - tile : (uint32_t)K8L_TILE;
        rate = (total << 8) / len;
        __syncthreads();                                        // changed_any[par] of the last round is read by all before anybody resets it
        if (tid < 2) changed_any[tid] = 0;
        __syncthreads();
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
    double r[34];                      // exp(-1 / m): i+1 one draw later
};
__device__ __forceinline__ int k8_octave_of(const K8Octaves &oc, double t) {
    int k = 0;
    while (k + 1 < oc.count && t >= oc.t_s[k + 1]) ++k;
    return k;
}
__device__ __forceinline__ uint32_t k8_c0_from(double i1, uint32_t n) {
    if (i1 < 1.0) i1 = 1.0;
    double c = (double)n - i1;
    if (c < 0.0) c = 0.0;
    if (c > (double)(n - 1u)) c = (double)(n - 1u);
    return (uint32_t)c;
}
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

// ---- the two full-length passes of the banded resolve, hand-written (round 3) ------------------------------------------------
// Round 2 ran them as rocPRIM scans over transform iterators with work hidden in the output iterators: a guess kernel (115 M
// double exps, 0.58 ms), a 64-bit scan that re-read draw and guess in its output side and stored a base count per draw (1.16 ms),
// a final scan that stored the exact count of every draw (0.91 ms) — 2.65 ms per category for what is one read of the draws each
// time.  Now: single-pass scans with decoupled look-back (a workgroup takes a tile of 8192 draws by ticket, publishes its
// aggregate in one 8-byte word, sums the words of the tiles before it), everything else fused in:
//   pass A  reads d; computes the guess (ONE exp per 8 consecutive draws, the others by the octave's decay ratio), classifies,
//           scans (certain acceptances | uncertain draws) as one packed word; stores the guess (pass B checks the band with it), the
//           class as a byte, and — for the uncertain draws only — position, word and base count straight into the list;
//   pass B  reads d, the byte (by now 0 / 1 everywhere) and the guess; scans the acceptances; the exclusive count says which step
//           a draw belongs to, so the partner key[i] is written at once; no count is ever stored.
constexpr int K8S_THREADS = 1024, K8S_EPT = 8, K8S_TILE = K8S_THREADS * K8S_EPT;
constexpr unsigned long long K8S_AGG = 1ull << 62, K8S_PFX = 2ull << 62, K8S_VALUE = (1ull << 62) - 1;
constexpr int K8S_SPIN_LIMIT = 1 << 22;

// state[0] = ticket counter, state[1] = error word (a look-back gave up), state[2 + t] = look-back word of tile t.
// Returns the sum of the aggregates of every tile before `tile` (all threads get it); publishes this tile's words.
__device__ __forceinline__ unsigned long long k8s_lookback(unsigned long long *state, int64_t tile, unsigned long long mine,
                                                           unsigned long long *s_bcast) {
    unsigned long long *words = state + 2;
    const int tid = threadIdx.x, lane = tid & 63;
    if (tid == 0)
        __hip_atomic_store(&words[tile], (tile == 0 ? K8S_PFX : K8S_AGG) | mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (tid < 64) {
        unsigned long long base = 0;
        int64_t look = tile - 1;
        bool failed = false;
        while (look >= 0) {
            const int64_t t = look - lane;
            unsigned long long wv = K8S_PFX;      // lanes before tile 0 read as an empty prefix
            if (t >= 0) {
                int spins = 0;
                do {
                    wv = __hip_atomic_load(&words[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if ((wv >> 62) == 0 && ++spins > K8S_SPIN_LIMIT) { failed = true; break; }
                    if ((wv >> 62) == 0) __builtin_amdgcn_s_sleep(1);
                } while ((wv >> 62) == 0);
            }
            if (__any(failed)) { failed = true; break; }
            const unsigned long long has_pfx = __ballot((wv >> 62) == 2);
            const int first = has_pfx ? __ffsll((long long)has_pfx) - 1 : kWave;
            unsigned long long part = (lane <= first) ? (wv & K8S_VALUE) : 0ull;
#pragma unroll
            for (int dlt = 32; dlt >= 1; dlt >>= 1) {
                const unsigned int lo = (unsigned int)__shfl_xor((int)(unsigned int)part, dlt), hi = (unsigned int)__shfl_xor((int)(unsigned int)(part >> 32), dlt);
                part += ((unsigned long long)hi << 32) | lo;
            }
            base += part;
            if (has_pfx) break;
            look -= kWave;
        }
        if (lane == 0) {
            if (failed) { atomicExch(&state[1], 1ull); base = 0; }
            if (tile != 0)
                __hip_atomic_store(&words[tile], K8S_PFX | ((base + mine) & K8S_VALUE), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            *s_bcast = base;
        }
    }
    __syncthreads();
    return *s_bcast;
}

// exclusive scan of one packed 64-bit value per thread over the workgroup; `total` = the workgroup's sum
__device__ __forceinline__ unsigned long long k8s_block_scan(unsigned long long v, unsigned long long *s_wave, unsigned long long &total) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    unsigned long long incl = v;
#pragma unroll
    for (int dlt = 1; dlt < 64; dlt <<= 1) {
        const unsigned int lo = (unsigned int)__shfl_up((int)(unsigned int)incl, dlt), hi = (unsigned int)__shfl_up((int)(unsigned int)(incl >> 32), dlt);
        if (lane >= dlt) incl += ((unsigned long long)hi << 32) | lo;
    }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    unsigned long long before = 0;
    total = 0;
#pragma unroll
    for (int w = 0; w < K8S_THREADS / 64; ++w) {
        const unsigned long long x = s_wave[w];
        before += w < wave ? x : 0ull;
        total += x;
    }
    __syncthreads();                              // s_wave is reused by the caller's next scan
    return before + incl - v;
}

__global__ __launch_bounds__(K8S_THREADS) void k8_scan_classify(const uint32_t *__restrict__ d, K8Octaves oc, uint32_t n, int64_t draws, uint32_t cap,
                                                                uint32_t *__restrict__ c0_out, uint8_t *__restrict__ mark,
                                                                uint32_t *__restrict__ pos_u, uint32_t *__restrict__ dvu, uint32_t *__restrict__ bu,
                                                                unsigned long long *__restrict__ state, uint32_t *__restrict__ n_sel) {
    __shared__ unsigned long long s_wave[K8S_THREADS / 64];
    __shared__ unsigned long long s_bcast;
    __shared__ unsigned long long s_tile;
    const int tid = threadIdx.x;
    if (tid == 0) s_tile = atomicAdd(&state[0], 1ull);
    __syncthreads();
    const int64_t tile = (int64_t)s_tile;
    const int64_t t0 = tile * K8S_TILE + (int64_t)tid * K8S_EPT;
    uint32_t dv[K8S_EPT], c0[K8S_EPT], cls[K8S_EPT];
    const bool full = t0 + K8S_EPT <= draws;
    if (full) {
        const uint4 a = reinterpret_cast<const uint4 *>(d + t0)[0], b = reinterpret_cast<const uint4 *>(d + t0)[1];
        dv[0] = a.x; dv[1] = a.y; dv[2] = a.z; dv[3] = a.w; dv[4] = b.x; dv[5] = b.y; dv[6] = b.z; dv[7] = b.w;
    } else {
#pragma unroll
        for (int k = 0; k < K8S_EPT; ++k) dv[k] = (t0 + k < draws) ? d[t0 + k] : 0u;
    }
    {   // the guess for 8 consecutive draws: one exp, then the octave's decay ratio (a run that crosses into the next octave: one by one)
        const int k0 = k8_octave_of(oc, (double)t0);
        if (k0 + 1 >= oc.count || (double)(t0 + K8S_EPT - 1) < oc.t_s[k0 + 1]) {
            double i1 = oc.i_s1[k0] * exp(-((double)t0 - oc.t_s[k0]) / oc.m[k0]);
            const double r = oc.r[k0];
#pragma unroll
            for (int k = 0; k < K8S_EPT; ++k) { c0[k] = k8_c0_from(i1, n); i1 *= r; }
        } else {
#pragma unroll
            for (int k = 0; k < K8S_EPT; ++k) {
                const double t = (double)(t0 + k);
                const int kk = k8_octave_of(oc, t);
                c0[k] = k8_c0_from(oc.i_s1[kk] * exp(-(t - oc.t_s[kk]) / oc.m[kk]), n);
            }
        }
    }
    unsigned long long mine = 0;                  // certain acceptances | uncertain draws << 32
#pragma unroll
    for (int k = 0; k < K8S_EPT; ++k) {
        cls[k] = (t0 + k < draws) ? k8_classify(dv[k], (uint32_t)(t0 + k), c0[k], n) : 0u;
        mine += cls[k] == 1u ? 1ull : (cls[k] == 2u ? (1ull << 32) : 0ull);
    }
    unsigned long long tile_total;
    const unsigned long long in_tile = k8s_block_scan(mine, s_wave, tile_total);
    const unsigned long long before = k8s_lookback(state, tile, tile_total, &s_bcast) + in_tile;
    uint32_t cert = (uint32_t)before, unc = (uint32_t)(before >> 32);
#pragma unroll
    for (int k = 0; k < K8S_EPT; ++k) {
        if (cls[k] == 2u) {
            if (unc < cap) { pos_u[unc] = (uint32_t)(t0 + k); dvu[unc] = dv[k]; bu[unc] = cert; }
            ++unc;
        } else if (cls[k] == 1u) {
            ++cert;
        }
    }
    if (t0 <= draws - 1 && draws - 1 < t0 + K8S_EPT) *n_sel = unc;      // the thread holding the last draw: the list's length
    if (full) {
        reinterpret_cast<uint4 *>(c0_out + t0)[0] = make_uint4(c0[0], c0[1], c0[2], c0[3]);
        reinterpret_cast<uint4 *>(c0_out + t0)[1] = make_uint4(c0[4], c0[5], c0[6], c0[7]);
        const uint32_t lo = cls[0] | (cls[1] << 8) | (cls[2] << 16) | (cls[3] << 24), hi = cls[4] | (cls[5] << 8) | (cls[6] << 16) | (cls[7] << 24);
        reinterpret_cast<uint2 *>(mark + t0)[0] = make_uint2(lo, hi);
    } else {
#pragma unroll
        for (int k = 0; k < K8S_EPT; ++k)
            if (t0 + k < draws) { c0_out[t0 + k] = c0[k]; mark[t0 + k] = (uint8_t)cls[k]; }
    }
}

// res[2] = 1 when a count left the band it was classified under; res[6] = accepted draws in all
__global__ __launch_bounds__(K8S_THREADS) void k8_scan_final(const uint32_t *__restrict__ d, const uint8_t *__restrict__ mark,
                                                             const uint32_t *__restrict__ c0, uint32_t n, int64_t draws,
                                                             uint32_t *__restrict__ key, unsigned long long *__restrict__ state,
                                                             uint32_t *__restrict__ res) {
    __shared__ unsigned long long s_wave[K8S_THREADS / 64];
    __shared__ unsigned long long s_bcast;
    __shared__ unsigned long long s_tile;
    const int tid = threadIdx.x;
    if (tid == 0) s_tile = atomicAdd(&state[0], 1ull);
    __syncthreads();
    const int64_t tile = (int64_t)s_tile;
    const int64_t t0 = tile * K8S_TILE + (int64_t)tid * K8S_EPT;
    uint32_t dv[K8S_EPT], g[K8S_EPT], f[K8S_EPT];
    if (t0 + K8S_EPT <= draws) {
        const uint4 a = reinterpret_cast<const uint4 *>(d + t0)[0], b = reinterpret_cast<const uint4 *>(d + t0)[1];
        dv[0] = a.x; dv[1] = a.y; dv[2] = a.z; dv[3] = a.w; dv[4] = b.x; dv[5] = b.y; dv[6] = b.z; dv[7] = b.w;
        const uint4 ga = reinterpret_cast<const uint4 *>(c0 + t0)[0], gb = reinterpret_cast<const uint4 *>(c0 + t0)[1];
        g[0] = ga.x; g[1] = ga.y; g[2] = ga.z; g[3] = ga.w; g[4] = gb.x; g[5] = gb.y; g[6] = gb.z; g[7] = gb.w;
        const uint2 m = reinterpret_cast<const uint2 *>(mark + t0)[0];
#pragma unroll
        for (int k = 0; k < 4; ++k) { f[k] = (m.x >> (8 * k)) & 0xffu; f[4 + k] = (m.y >> (8 * k)) & 0xffu; }
    } else {
#pragma unroll
        for (int k = 0; k < K8S_EPT; ++k) {
            const bool in = t0 + k < draws;
            dv[k] = in ? d[t0 + k] : 0u; g[k] = in ? c0[t0 + k] : 0u; f[k] = in ? (uint32_t)mark[t0 + k] : 0u;
        }
    }
    unsigned long long mine = 0;
#pragma unroll
    for (int k = 0; k < K8S_EPT; ++k) mine += f[k] ? 1ull : 0ull;
    unsigned long long tile_total;
    const unsigned long long in_tile = k8s_block_scan(mine, s_wave, tile_total);
    const unsigned long long base = k8s_lookback(state, tile, tile_total, &s_bcast);
    uint32_t v = (uint32_t)(base + in_tile);      // accepted draws before draw t0: the exact count
    bool bad = false;
#pragma unroll
    for (int k = 0; k < K8S_EPT; ++k) {
        const int64_t t = t0 + k;
        if (t < draws) {
            const uint32_t K = k8_band((uint32_t)t, g[k]);
            bad |= (v > g[k] ? v - g[k] : g[k] - v) > K;
            if (t == 0) key[0] = 0u;                // step 0 does not exist: a no-op
            if (v < n - 1u) {
                const uint32_t i = n - 1u - v, h = dv[k] & k8_mask(i);
                if (h <= i) key[i] = h;
            }
            v += f[k] ? 1u : 0u;
        }
    }
    if (bad) res[2] = 1u;
    if (t0 <= draws - 1 && draws - 1 < t0 + K8S_EPT) res[6] = v;
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
        if (oc && cnt < 34) { oc->t_s[cnt] = t; oc->i_s1[cnt] = (double)i + 1.0; oc->m[cnt] = M; oc->r[cnt] = exp(-1.0 / M); ++cnt; }
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
    {   // the banded resolve's own scans keep one look-back word per tile of 8192 draws (+ ticket and error word) here
        const size_t words = 8 * ((size_t)ceil_div(draws, (int64_t)K8S_TILE) + 2);
        tmp = words > tmp ? words : tmp;
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
static int resolve_banded(const uint32_t *d, int64_t draws, uint32_t n, const K8Octaves &oc, uint32_t *cA, uint32_t *hs, uint32_t *is, uint32_t *last,
                          uint32_t *pos, size_t region_bytes, void *tmp, size_t tmp_bytes, uint32_t *res, uint32_t *key, hipStream_t st,
                          int *rounds_out, bool *resolved, uint32_t *accepted_out) {
    *resolved = false;
    const size_t mark_bytes = (((size_t)draws) + 255) & ~(size_t)255;
    const int64_t n_tiles = ceil_div(draws, (int64_t)K8S_TILE);
    const size_t state_bytes = 8 * ((size_t)n_tiles + 2);
    if (region_bytes < mark_bytes + 1024 || tmp_bytes < state_bytes) {   // no room for the list: the full-length rounds (they want a first guess)
        hipLaunchKernelGGL(k8_guess, dim3((unsigned)ceil_div(draws, 256)), dim3(256), 0, st, oc, n, draws, cA);
        DYD_HIP(hipGetLastError());
        return DYD_OK;
    }
    uint8_t *mark = reinterpret_cast<uint8_t *>(hs);
    uint8_t *fu = reinterpret_cast<uint8_t *>(hs) + mark_bytes;          // outcome of every uncertain draw, a byte each
    const size_t cap_b = region_bytes - mark_bytes, cap = cap_b < (size_t)n ? cap_b : (size_t)n;
    uint32_t *pos_u = is, *bu = last, *dvu = pos;                        // the list, in the regions the sort phase uses afterwards
    unsigned long long *state = static_cast<unsigned long long *>(tmp);
    uint32_t *n_sel = res + 4;
    // pass A: the guess (kept in cA), the class of every draw (a byte in mark) and the list of the uncertain draws
    DYD_HIP(hipMemsetAsync(state, 0, state_bytes, st));
    DYD_HIP(hipMemsetAsync(res, 0, 32, st));
    hipLaunchKernelGGL(k8_scan_classify, dim3((unsigned)n_tiles), dim3(K8S_THREADS), 0, st, d, oc, n, draws, (uint32_t)cap, cA, mark, pos_u, dvu, bu, state, n_sel);
    DYD_HIP(hipGetLastError());
    uint32_t n_u = 0;
    unsigned long long failed_a = 0;
    DYD_HIP(hipMemcpyAsync(&n_u, n_sel, 4, hipMemcpyDeviceToHost, st));
    DYD_HIP(hipMemcpyAsync(&failed_a, state + 1, 8, hipMemcpyDeviceToHost, st));
    DYD_HIP(hipStreamSynchronize(st));
    if (failed_a || (size_t)n_u > cap || (int64_t)n_u >= draws) return DYD_OK;   // not worth it (or no room): the full-length rounds, from the guess in cA
    if (n_u) {
        hipLaunchKernelGGL(k8_list_resolve, dim3(1), dim3(K8L_THREADS), 0, st, dvu, bu, n_u, n, fu, res + 5);
        DYD_HIP(hipGetLastError());
        hipLaunchKernelGGL(k8_list_mark, dim3((unsigned)ceil_div((int64_t)n_u, 256)), dim3(256), 0, st, pos_u, fu, n_u, mark);
        DYD_HIP(hipGetLastError());
    }
    // pass B: exact counts (never stored) -> partners, checked against the band the classes were derived under
    DYD_HIP(hipMemsetAsync(state, 0, state_bytes, st));
    hipLaunchKernelGGL(k8_scan_final, dim3((unsigned)n_tiles), dim3(K8S_THREADS), 0, st, d, mark, cA, n, draws, key, state, res);
    DYD_HIP(hipGetLastError());
    uint32_t back[8] = {0, 0, 1, 0, 0, 0, 0, 0};
    unsigned long long failed_b = 0;
    DYD_HIP(hipMemcpyAsync(back, res, 32, hipMemcpyDeviceToHost, st));
    DYD_HIP(hipMemcpyAsync(&failed_b, state + 1, 8, hipMemcpyDeviceToHost, st));
    DYD_HIP(hipStreamSynchronize(st));
    if (rounds_out) *rounds_out = (int)back[5];           // local rounds of the list walk, all tiles together
    if (back[2] || failed_b) return DYD_OK;   // a count left its band somewhere: nothing above is trusted (the full-length rounds rewrite every partner)
    *accepted_out = back[6];
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

    int rounds = 0;
    bool resolved = false;
    uint32_t accepted = 0;
    if (n >= K8_BAND_MIN_N && g_k8_band) {
        const int rc = resolve_banded(d, draws, n, oc, cA, hs, is, last, pos, b, tmp, tmp_bytes, res, key, st, &rounds, &resolved, &accepted);
        if (rc) return rc;
    } else {
        hipLaunchKernelGGL(k8_guess, dim3((unsigned)ceil_div(draws, 256)), dim3(256), 0, st, oc, n, draws, cA);
        DYD_HIP(hipGetLastError());
    }
    // Picard rounds over the not yet final suffix [lo, draws): cB[t] = c_lo + sum of flags(cA) on [lo, t).  Whatever lies before
    // the first difference is final (its flags were computed from exact counts), so the partners of that stretch are written
    // at once and the next round starts there.  (Short permutations, and the fallback of the banded resolve.)
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
    // enough draws?  n - 1 of them must have been accepted
    if (resolved) {
        if (accepted < n - 1u) return DYD_ERR_RANGE;
    } else {   // the full-length rounds: the last draw's count (+ its own flag); cB holds the final counts of the tail
        uint32_t c_last = 0, d_last = 0;
        DYD_HIP(hipMemcpyAsync(&c_last, cB + (draws - 1), 4, hipMemcpyDeviceToHost, st));
        DYD_HIP(hipMemcpyAsync(&d_last, d + (draws - 1), 4, hipMemcpyDeviceToHost, st));
        DYD_HIP(hipStreamSynchronize(st));
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
