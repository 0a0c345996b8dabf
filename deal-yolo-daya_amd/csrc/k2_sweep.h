// k2_sweep.h — K2 for a row of 40..1024 boxes: sort by the left edge, then only look at the partners that can still reach thr.
//
// The pair loops of k2_wave.h / k2_filter.h visit all n(n-1)/2 pairs of a row (reference core/processor.py:368-376,
// `any(calculate_iou(...) >= thr ...)` over i < j); for a row of 256 boxes that is 32 640 reject tests, 510 trips per lane,
// and the table of configs[4] (256 boxes per image) was bound by exactly that loop.  Most of those pairs cannot hit:
//   IoU >= thr  =>  inter >= thr * union >= thr * area_a  and  inter_h <= h_a  =>  inter_w >= thr * w_a
//   inter_w <= x2_a - x1_b                                                      =>  x1_b <= x2_a - thr * w_a =: lim_a
// for EITHER box of the pair as `a` (no assumption on which one lies further left).  So with the row's boxes ordered by x1,
// box a only has to meet the boxes after it while their x1 stays <= lim_a — for thr = 0.98 a window of 2 % of its width.
// The test is only a filter: what passes it (and a y-overlap test) is queued and decided by the exact f64 code
// (k2s_drain -> pair_hits), so rounding in the filter may only ever ADMIT pairs.  Hence: thr_lo = 0.999 thr instead of thr (the
// margin pair_hits already uses for its division shortcut), x1 bounded from below and lim from above in f32 (k2_filter.h), the low
// IB = 8 or 10 bits of the (order-preserving) key given to the box index — x1 is truncated DOWN by up to 2^IB - 1 ulp, lim pushed
// UP by 2^IB .. 2^(IB+1) - 1 ulp.  A row holding a corner that is not finite, or whose lim overflows, is left to the all-pairs
// code.  For the diagnostic maximum (WANT_MAX) the window is the overlap window, lim_a = x2_a: every pair with a non-empty
// intersection is evaluated, the others contribute the 0.0 the maximum starts from.
// tests/test_sweep_filter_cpu.py restates the filter in numpy and checks the argument; tests/test_gpu_sweep.py checks the kernels.
//
// Pieces: k2s_prepare (a box -> key, limit, y interval), k2s_sort_regs (bitonic network over 64 * E keys held E per lane, E = 1 .. 16;
// padding keys 0xffffffff sort last and pass no window), k2s_sweep_sorted (sort, sweep, exact tests; arrays through a K2sView of the
// caller's LDS), k2s_row (a row whose boxes are in memory: the tile kernels).  Callers that have the boxes in registers (k12_wave.h,
// DENSE) or more LDS (k2_big_rows_kernel: rows of up to 1024 boxes) use the pieces directly.
#pragma once

namespace dyd {

#ifdef K2S_DEBUG
static __device__ unsigned long long *g_k2s_dbg = nullptr;   // experiment counters (tools only; never defined in the product build)
#define K2S_DBG_ADD(i, v) do { if (g_k2s_dbg && (threadIdx.x & 63) == 0) atomicAdd(&g_k2s_dbg[i], (unsigned long long)(v)); } while (0)
#define K2S_DBG_MAX(i, v) do { if (g_k2s_dbg && (threadIdx.x & 63) == 0) atomicMax(&g_k2s_dbg[i], (unsigned long long)(v)); } while (0)
#else
#define K2S_DBG_ADD(i, v) do {} while (0)
#define K2S_DBG_MAX(i, v) do {} while (0)
#endif

#ifndef K2S_MIN_VALUE
#define K2S_MIN_VALUE 40
#endif
constexpr int32_t K2S_MIN = K2S_MIN_VALUE;   // rows from this many boxes on are swept (tools/dense_sweep.py at 40 / 48 / 64 / 80 boxes per row: 0.36 / 0.31 / 0.26 / 0.33 ms against 0.57 / 0.69 / 0.67 / 1.14 for the all-pairs tiles)

// f32 bit pattern -> u32 that orders like the value (-inf < ... < -0 < +0 < ... < +inf)
__device__ __forceinline__ uint32_t f32_order(float f) {
    const uint32_t b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

// median of three u32: with c = 0 it is min(a, b), with c = 0xffffffff max(a, b) — one instruction decides a compare-exchange
// whose direction differs from lane to lane
__device__ __forceinline__ uint32_t k2s_med3(uint32_t a, uint32_t b, uint32_t c) {
    uint32_t r;
    asm("v_med3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
// bit `bit` of the lane number spread over a word (0 or 0xffffffff).  volatile: the masks of the 21 cross-lane stages are loop
// invariants the compiler would otherwise keep in 27 registers across the whole kernel (85 -> 112 VGPRs, a wave less per SIMD)
template <int BIT>
__device__ __forceinline__ uint32_t k2s_lane_bit(int lane) {
    uint32_t r;
    asm volatile("v_bfe_i32 %0, %1, %2, 1" : "=v"(r) : "v"(lane), "n"(BIT));
    return r;
}
constexpr int k2s_log2(int x) { return x <= 1 ? 0 : 1 + k2s_log2(x >> 1); }

// Bitonic sort of 64 * E keys held E per lane (sorted position of v[r] afterwards: lane * E + r), ascending.  Strides below E
// are compare-exchanges between a lane's own registers, the others meet the partner lane through one cross-lane move per
// register; nothing touches LDS memory, and a stage costs ONE VALU instruction per key (v_med3_u32) instead of an LDS round trip.
// `down`: 0 where the merge of level K runs ascending for the lane's keys, all ones where descending (levels below E alternate
// inside the lane and are resolved at compile time).
template <int E, int K, int J>
__device__ __forceinline__ void k2s_sort_stage(uint32_t (&v)[E], int lane, uint32_t down) {
    if constexpr (J >= E) {
        constexpr int LJ = J / E;   // partner lane = lane ^ LJ, same register; the lower lane of an ascending pair keeps the minimum
        const uint32_t sel = k2s_lane_bit<k2s_log2(LJ)>(lane) ^ down;
#pragma unroll
        for (int r = 0; r < E; ++r) {
            const uint32_t o = (uint32_t)__shfl_xor((int)v[r], LJ);
            v[r] = k2s_med3(v[r], o, sel);
        }
    } else {
#pragma unroll
        for (int r = 0; r < E; ++r) {
            if ((r & J) == 0) {
                const int r2 = r | J;
                const uint32_t lo_sel = (K < E) ? ((r & K) == 0 ? 0u : 0xffffffffu) : down;
                const uint32_t a = v[r], b = v[r2];
                v[r] = k2s_med3(a, b, lo_sel);
                v[r2] = k2s_med3(a, b, ~lo_sel);
            }
        }
    }
    if constexpr (J > 1) k2s_sort_stage<E, K, J / 2>(v, lane, down);
}
template <int E, int K>
__device__ __forceinline__ void k2s_sort_level(uint32_t (&v)[E], int lane) {
    constexpr int P = 64 * E;
    uint32_t down = 0u;
    if constexpr (K >= E && K < P) down = k2s_lane_bit<k2s_log2(K / E)>(lane);
    k2s_sort_stage<E, K, K / 2>(v, lane, down);
    if constexpr (K < P) k2s_sort_level<E, 2 * K>(v, lane);
}
template <int E>
__device__ __forceinline__ void k2s_sort_regs(uint32_t (&v)[E], int lane) {
    k2s_sort_level<E, 2>(v, lane);
}

// Where the sweep keeps its arrays (all in the calling wave's LDS): sorted keys [cap], limits and y intervals by box index
// [cap], and the queue of candidate pairs (2 x 128 box indices).  cap = 64 * E of the instantiation.
struct K2sView {
    uint32_t *skey, *slim;
    float2 *syy;
    uint32_t *qa, *qb;
};

// One box -> its sweep record.  Corners must be normalised (x1 <= x2, y1 <= y2 unless a NaN is involved); returns false when
// one of them or the limit is not finite (the row must then take the all-pairs code).
// IB = bits of the key that carry the box index (8: rows of up to 256 boxes, 10: up to 1024): x1 is truncated DOWN by up to 2^IB - 1 ulp and
// the limit pushed UP by 2^IB .. 2^(IB+1) - 1 ulp.
// diag: sort axis = the top-left corner's diagonal, s = x1 + y1 against lim = (x2 - tl w) + (y2 - tl h) — the x bound and the same bound
// in y added up.  Valid for the same reason, but without the x test the x order gives for free (twice the exact tests on ordinary
// tables), so it is only the second attempt for a row whose x1 order degenerates (a column of boxes with equal x1, see k2s_retry_diag).
template <int IB = 8>
__device__ __forceinline__ bool k2s_prepare(const Corners &c, uint32_t k, double tl, uint32_t &key, uint32_t &lim, float2 &yy, bool diag = false) {
    constexpr uint32_t IM = (1u << IB) - 1u;
    const double lx = c.x2 - tl * (c.x2 - c.x1);
    const double sk = diag ? c.x1 + c.y1 : c.x1;
    const double l = diag ? lx + (c.y2 - tl * (c.y2 - c.y1)) : lx;
    // all finite?  one sum of magnitudes (NaN and inf propagate; a sum of huge finite values that overflows only sends the row to
    // the all-pairs code)
    const bool ok = __builtin_fabs(c.x1) + __builtin_fabs(c.y1) + __builtin_fabs(c.x2) + __builtin_fabs(c.y2) + __builtin_fabs(l) +
                    __builtin_fabs(sk) < __builtin_inf();
    key = (f32_order(f32_below(sk)) & ~IM) | k;
    lim = (f32_order(f32_above(l)) + (IM + 1u)) | IM;   // finite limit: at most 0xff7fffff + 2^IB, no wrap (padding keys stay above)
    yy = make_float2(f32_below(c.y1), f32_above(c.y2));
    return ok;
}

// exact f64 test of `cnt` (<= 64) queued pairs from queue slot `first` on; the diagnostic maximum stays in the lane's register
// (64 LDS atomics on one row's slot serialise: they cost more than the tests)
template <bool WANT_MAX>
__device__ __forceinline__ bool k2s_drain(const double *box4, int64_t base, const K2sView &V, int first, int cnt, double thr, double thr_lo,
                                          double &mxacc) {
    const int lane = threadIdx.x & 63;
    bool hit = false;
    if (lane < cnt) {
        const uint32_t a = V.qa[first + lane], b = V.qb[first + lane];
        const uint32_t lo = a < b ? a : b, hi = a < b ? b : a;   // the reference's (i < j) argument order
        const Corners p = load_corners(box4, base + lo), q = load_corners(box4, base + hi);
        double mx = 0.0;
        hit = pair_hits<WANT_MAX, true>(p, q, area_of(p), q, thr, thr_lo, false, mx);
        if (WANT_MAX && mx > mxacc) mxacc = mx;
    }
    return __any(hit);
}

// The keys of a row's n boxes are in the lanes' registers (any order, padding 0xffffffff), limits and y intervals in LDS under
// the box index: sort, sweep, exact tests.  Returns whether a pair reached thr; with WANT_MAX mxacc is every lane's running maximum.
// budget > 0: give up (*aborted = true, nothing decided unless a hit was already found) once the loop has made more trips than that —
// the sign of a row whose keys do not spread (every box in every other's window); the caller then tries the other sort axis.
template <bool WANT_MAX, int E, int IB = 8, bool BUDGET = false>
__device__ __forceinline__ bool k2s_sweep_sorted(const double *box4, int64_t base, int32_t n, const K2sView &V, uint32_t (&v)[E], double thr,
                                                 double thr_lo, double &mxacc, int32_t budget = 0, bool *aborted = nullptr) {
    constexpr int P = 64 * E;
    constexpr uint32_t IM = (1u << IB) - 1u;
    static_assert(P <= (1 << IB), "the box index must fit the key's low bits");
    const int lane = threadIdx.x & 63;
    const unsigned long long lt = lane ? (~0ull >> (64 - lane)) : 0ull;
    [[maybe_unused]] unsigned long long n_it = 0, n_cand = 0, n_drain = 0;
    k2s_sort_regs<E>(v, lane);
#pragma unroll
    for (int r = 0; r < E; ++r) V.skey[lane * E + r] = v[r];
    wave_sync();

    // ---- sweep: lane = sorted position p, partners p+1, p+2, ... while inside the window ---------------------------
    int qn = 0;
    int32_t trips = 0;
    bool any_hit = false, gave_up = false;
    for (int32_t p0 = 0; p0 < n - 1 && !gave_up; p0 += kWave) {
        const int32_t p = p0 + lane;
        const bool have = p < n - 1;
        const int ia = (int)(V.skey[have ? p : 0] & IM);
        const uint32_t lim = have ? V.slim[ia] : 0u;
        const float2 my = V.syy[ia];
        for (int32_t d = 1;; ++d) {
            const int32_t j = p + d;
            const uint32_t kj = (j < P) ? V.skey[j] : 0xffffffffu;
            const bool inwin = have && kj <= lim;
#ifdef K2S_DEBUG
            n_it += 1;
#endif
            if (!__any(inwin)) break;
            if (BUDGET && budget > 0 && ++trips > budget) {   // wave-uniform
                gave_up = true;
                break;
            }
            bool cand = false;
            const int ib = (int)(kj & IM);
            if (inwin) {
                const float2 o = V.syy[ib];
                cand = my.y > o.x && o.y > my.x;   // y intervals overlap (outward-rounded, so never a false reject)
            }
            const unsigned long long m = __ballot(cand);
            if (m) {
                if (cand) {
                    const int slot = qn + __popcll(m & lt);
                    V.qa[slot] = (uint32_t)ia;
                    V.qb[slot] = (uint32_t)ib;
                }
                qn += __popcll(m);
#ifdef K2S_DEBUG
                n_cand += __popcll(m);
#endif
                if (qn >= kWave) {
                    wave_sync();
#ifdef K2S_DEBUG
                    n_drain += 1;
#endif
                    any_hit |= k2s_drain<WANT_MAX>(box4, base, V, qn - kWave, kWave, thr, thr_lo, mxacc);
                    qn -= kWave;
                    wave_sync();
                    if (!WANT_MAX && any_hit) break;   // any() is decided
                }
            }
        }
        if (!WANT_MAX && any_hit) break;
    }
    if (BUDGET && gave_up && (WANT_MAX || !any_hit)) {   // the queued pairs are dropped: the second attempt meets them again
        if (aborted) *aborted = true;
        qn = 0;
    }
    if (qn > 0 && (WANT_MAX || !any_hit)) {
        wave_sync();
#ifdef K2S_DEBUG
        n_drain += 1;
#endif
        any_hit |= k2s_drain<WANT_MAX>(box4, base, V, 0, qn, thr, thr_lo, mxacc);
    }
    K2S_DBG_ADD(0, 1);
    K2S_DBG_ADD(2, n_it);
    K2S_DBG_ADD(3, n_cand);
    K2S_DBG_ADD(4, n_drain);
    K2S_DBG_MAX(5, n_it);
    wave_sync();
    return any_hit;
}

// Second attempt for a row that the x1 order could not spread (k2s_sweep_sorted gave up): the boxes are read back from memory, keyed by
// the diagonal and swept without a budget.  Should the diagonal sums overflow, the x1 order runs again to the end instead.
template <bool WANT_MAX, int E, int IB = 8>
__device__ __forceinline__ bool k2s_retry_diag(const double *box4, int64_t base, int32_t n, const K2sView &V, double tl, double thr, double thr_lo,
                                               double &mxacc) {
    const int lane = threadIdx.x & 63;
    K2S_DBG_ADD(9, 1);
    for (int attempt = 0; attempt < 2; ++attempt) {
        const bool diag = attempt == 0;
        uint32_t v[E];
        bool bad = false;
        wave_sync();
#pragma unroll
        for (int r = 0; r < E; ++r) {
            const int32_t k = lane + kWave * r;
            v[r] = 0xffffffffu;
            if (k < n) {
                const Corners c = load_corners(box4, base + k);
                uint32_t lim;
                float2 yy;
                bad |= !k2s_prepare<IB>(c, (uint32_t)k, tl, v[r], lim, yy, diag);
                V.slim[k] = lim;
                V.syy[k] = yy;
            }
        }
        if (diag && __any(bad)) continue;
        return k2s_sweep_sorted<WANT_MAX, E, IB>(box4, base, n, V, v, thr, thr_lo, mxacc);
    }
    return false;
}

// A row whose boxes are in memory (the tile kernels): load, prepare, sweep.  false = not finite, nothing was decided.
template <bool WANT_MAX, int E, int WROWS, int WCAP>
__device__ __forceinline__ bool k2s_row_e(const double *box4, int64_t base, int32_t n, int row, WaveLdsF<WROWS, WCAP> &S,
                                          int32_t min_boxes, double thr, double thr_lo) {
    static_assert(64 * E <= WCAP && WCAP <= 256, "box index lives in 8 key bits");
    static_assert(sizeof(S.cf) >= 16 * (size_t)WCAP, "keys + limits + y intervals alias the float4 tile");
    const int lane = threadIdx.x & 63;
    uint32_t *skey = reinterpret_cast<uint32_t *>(S.cf);
    const K2sView V = {skey, skey + WCAP, reinterpret_cast<float2 *>(skey + 2 * WCAP), S.qa, S.qb};
    const double tl = WANT_MAX ? 0.0 : thr_lo;
    bool bad = false;
    uint32_t v[E];
    wave_sync();
#pragma unroll
    for (int r = 0; r < E; ++r) {
        const int32_t k = lane + kWave * r;
        v[r] = 0xffffffffu;
        if (k < n) {
            // corners by IEEE min / max: they differ from the reference's first-wins min / max only for a NaN (the row then leaves
            // through `bad`) and in the sign of a zero (inside the filter's slack)
            const double2 *g = reinterpret_cast<const double2 *>(box4 + 4 * (base + k));
            const double2 pa = g[0], pb = g[1];
            const Corners c = {vmin(pa.x, pb.x), vmin(pa.y, pb.y), vmax(pa.x, pb.x), vmax(pa.y, pb.y)};
            uint32_t lim;
            float2 yy;
            // vmin / vmax drop a NaN operand: the finiteness test must see the raw corners as well
            bad |= !k2s_prepare(c, (uint32_t)k, tl, v[r], lim, yy) || !(pa.x + pa.y + pb.x + pb.y == pa.x + pa.y + pb.x + pb.y);
            V.slim[k] = lim;
            V.syy[k] = yy;
        }
    }
    if (__any(bad)) {
        wave_sync();
        K2S_DBG_ADD(1, 1);
        return false;
    }
    double mxacc = 0.0;
    const bool any_hit = k2s_sweep_sorted<WANT_MAX, E>(box4, base, n, V, v, thr, thr_lo, mxacc);
    if (any_hit && n >= min_boxes && lane == 0) S.flag[row] = 1;
    if (WANT_MAX) {
        unsigned long long bits = (unsigned long long)__double_as_longlong(mxacc);   // IoU >= 0: the bit patterns order like the values
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            const unsigned long long o = __shfl_xor(bits, d);
            bits = o > bits ? o : bits;
        }
        if (lane == 0 && bits > S.mx[row]) S.mx[row] = bits;
    }
    wave_sync();
    return true;
}

template <bool WANT_MAX, int WROWS, int WCAP>
__device__ __forceinline__ bool k2s_row(const double *box4, int64_t base, int32_t n, int row, WaveLdsF<WROWS, WCAP> &S,
                                        int32_t min_boxes, double thr, double thr_lo) {
    static_assert(WCAP == 128 || WCAP == 256, "one, two or four keys per lane");
    if constexpr (WCAP == 256) {
        if (n > 128) return k2s_row_e<WANT_MAX, 4>(box4, base, n, row, S, min_boxes, thr, thr_lo);
    }
    if (n > 64) return k2s_row_e<WANT_MAX, 2>(box4, base, n, row, S, min_boxes, thr, thr_lo);
    return k2s_row_e<WANT_MAX, 1>(box4, base, n, row, S, min_boxes, thr, thr_lo);
}

}  // namespace dyd
