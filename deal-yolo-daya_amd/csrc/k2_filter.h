// k2_filter.h — K2 with an f32 reject filter in front of the exact f64 IoU test.
//
// Same contract and wave-autonomous mapping as k2_wave.h (reference core/processor.py:328-339,
// :359-362, :368-376), but the all-pairs loop only asks "can these two boxes overlap at all?":
//   * every box is staged in LDS as ONE float4 = its corners rounded OUTWARD to f32
//     (x1, y1 down, x2, y2 up), so the f32 box contains the f64 box;
//   * two boxes have a positive-area intersection only if  x2a > x1b, x2b > x1a, y2a > y1b and
//     y2b > y1a; each f64 inequality implies its outward-rounded f32 counterpart, so four
//     v_cmp_gt_f32 reject a pair without a single f64 instruction.  A rejected pair has w <= 0 or
//     h <= 0, i.e. an intersection of 0 (or NaN from 0*inf): IoU 0.0 / NaN, never >= thr for thr > 0
//     and never the row's maximum;
//   * the few surviving pairs (~2-3 % on the synthetic tables) are appended to a per-wave queue with
//     ballot + popcount and evaluated EXACTLY 64 at a time, all lanes busy, with the f64 corners
//     re-read from global memory (L2-resident: the wave read or wrote them moments earlier).
// Rows that hold a NaN corner, and every row when thr <= 0 (an empty intersection then counts as a
// hit), skip the filter and run the ordered exact loop.
// LDS per wave: 16 B per box instead of 32 B (+1 KiB queue), so more waves fit a CU.
#pragma once

#include "k2_wave.h"

namespace dyd {

template <int WROWS, int WCAP>
struct alignas(16) WaveLdsF {
    float4 cf[WCAP];
    uint32_t qa[2 * kWave], qb[2 * kWave];  // queued candidate pairs (tile-local box indices)
    unsigned long long mx[WROWS];
    int32_t off[WROWS + 4];
    int32_t flag[WROWS];
    int32_t nan[WROWS];
    int32_t sst[WROWS];
    unsigned short row[WCAP];
    unsigned short perm[WCAP];
};

// f32 value <= v / >= v for every finite or infinite v (NaN stays NaN).  Not the neighbouring f32 — that costs a dozen instructions
// (convert back, compare, three sign cases) per corner — but the nearest f32 pushed outward by one part in 2^23 of itself plus
// 2^-120: the nearest f32 is within half an ulp of v and an ulp is at most 2^-23 of the value, so the push clears v whatever the
// fma's own rounding does; the constant covers the denormal range where the relative push vanishes.  |v| > FLT_MAX converts to
// +-inf: the bound on the far side is clamped to +-FLT_MAX first.  Four instructions; the filters only need SOME bound.
__device__ __forceinline__ float f32_below(double v) {
    const float f = fminf((float)v, 3.402823466e+38f);
    return fmaf(-fabsf(f), 0x1p-23f, f) - 0x1p-120f;
}
__device__ __forceinline__ float f32_above(double v) {
    const float f = fmaxf((float)v, -3.402823466e+38f);
    return fmaf(fabsf(f), 0x1p-23f, f) + 0x1p-120f;
}
// lane mask of a > b (false for NaN), kept in an SGPR pair: written as C++ the four tests of a pair are widened to
// integers and recombined with 16-bit VALU ops (17 VALU instructions per pair instead of 4 compares + scalar ANDs)
__device__ __forceinline__ unsigned long long gt_mask(float a, float b) {
    unsigned long long m;
    asm volatile("v_cmp_gt_f32 %0, %1, %2" : "=s"(m) : "v"(a), "v"(b));
    return m;
}
__device__ __forceinline__ unsigned long long overlap_mask(const float4 &a, const float4 &b) {
    return gt_mask(a.z, b.x) & gt_mask(b.z, a.x) & gt_mask(a.w, b.y) & gt_mask(b.w, a.y);
}

__device__ __forceinline__ float4 outward_f32(const Corners &c) {
    return make_float4(f32_below(c.x1), f32_below(c.y1), f32_above(c.x2), f32_above(c.y2));
}
__device__ __forceinline__ Corners load_corners(const double *box4, int64_t idx) {
    const double2 *g = reinterpret_cast<const double2 *>(box4 + 4 * idx);
    return normalise(g[0], g[1]);
}

// exact f64 test of `cnt` (<= 64) queued pairs starting at queue slot `first`; `row` < 0: look the row
// up through S.row[a]
template <bool WANT_MAX, int WROWS, int WCAP>
__device__ __forceinline__ void k2f_drain(const double *box4, int64_t base, WaveLdsF<WROWS, WCAP> &S, int first, int cnt,
                                          int row, int32_t min_boxes, double thr, double thr_lo) {
    const int lane = threadIdx.x & 63;
    if (lane < cnt) {
        const uint32_t a = S.qa[first + lane], b = S.qb[first + lane];
        const uint32_t lo = a < b ? a : b, hi = a < b ? b : a;
        const Corners p = load_corners(box4, base + lo), q = load_corners(box4, base + hi);
        const int lr = row >= 0 ? row : (int)S.row[a];
        double mx = 0.0;
        const bool hit = pair_hits<WANT_MAX, true>(p, q, area_of(p), q, thr, thr_lo, false, mx);
        const int32_t n = S.off[lr + 1] - S.off[lr];
        if (hit && n >= min_boxes) S.flag[lr] = 1;
        if (WANT_MAX) atomicMax(&S.mx[lr], (unsigned long long)__double_as_longlong(mx));
    }
}

}  // namespace dyd

#include "k2_sweep.h"

namespace dyd {

template <bool WANT_MAX, int WROWS, int WCAP>
__device__ __forceinline__ void k2f_wave_rows(const double *box4, const int32_t *__restrict__ row_off, int64_t r0,
                                              int nr, int32_t min_boxes, double thr, uint8_t *__restrict__ out_high,
                                              double *__restrict__ out_max, WaveLdsF<WROWS, WCAP> &S,
                                             unsigned long long *bigq = nullptr) {
    static_assert(WROWS < 63 && WCAP <= 65535, "rows map to lanes, box ids to 16 bits");
    const int lane = threadIdx.x & 63;
    const unsigned long long lt = lane ? (~0ull >> (64 - lane)) : 0ull;
    int32_t my_off = 0;
    if (lane <= nr) {
        my_off = row_off[r0 + lane];
        S.off[lane] = my_off;
    }
    if (lane < WROWS) {
        S.flag[lane] = 0;
        S.nan[lane] = 0;
        S.mx[lane] = 0ull;
    }
    const int32_t my_n = __shfl_down(my_off, 1) - my_off;
    wave_sync();
    const bool zero_hits = (0.0 >= thr);
    const double thr_lo = (thr > 0.0) ? thr * 0.999 : 0.0;

    int ra = 0;
    while (ra < nr) {  // every condition below is wave-uniform
        const int32_t base = __builtin_amdgcn_readlane(my_off, ra);
        {   // ---- a row of many boxes that fits the tile: sorted by x1 and swept (k2_sweep.h) -----
            const int32_t n0 = __builtin_amdgcn_readlane(my_off, ra + 1) - base;
            // (no trip budget here, unlike the fused wave kernels: the bookkeeping costs these kernels 5-10 % on ordinary dense tables —
            // 8 more VGPRs in the 256-box tiling — and a row the x1 order cannot spread is "only" back to all-pairs cost)
            if (n0 >= K2S_MIN && n0 <= WCAP && !zero_hits && (WANT_MAX || n0 >= min_boxes) &&
                k2s_row<WANT_MAX>(box4, (int64_t)base, n0, ra, S, min_boxes, thr, thr_lo)) {
                ra += 1;
                continue;
            }
        }
        const unsigned long long fits = __ballot(lane > ra && lane <= nr && my_off - base <= WCAP);
        const int taken = __popcll(fits);
        if (taken == 0) {
            // ---- one row larger than the LDS tile: partner tiles of f32 boxes stream through LDS -----
            const int32_t n = __builtin_amdgcn_readlane(my_off, ra + 1) - base;
            if (k2_defer_row<WANT_MAX>(bigq, r0 + ra, n, min_boxes, zero_hits)) {   // left to k2_big_rows_kernel
                ra += 1;
                continue;
            }
            const bool counted = WANT_MAX || n >= min_boxes;
            // does the row hold a NaN corner?  (one pass; also decides filter vs exact loop)
            bool nanrow = false;
            if (counted) {
                bool mine = false;
                for (int32_t k = lane; k < n; k += kWave) mine |= has_nan(load_corners(box4, (int64_t)base + k));
                nanrow = __any(mine);
            }
            if (counted && (nanrow || zero_hits)) {
                bool hit = false;
                double mx = 0.0;
                for (int32_t i = lane; i < n - 1; i += kWave) {
                    const Corners me = load_corners(box4, (int64_t)base + i);
                    const double me_ar = area_of(me);
                    for (int32_t j = i + 1; j < n; ++j) {
                        const Corners o = load_corners(box4, (int64_t)base + j);
                        hit |= pair_hits<WANT_MAX, false>(me, o, me_ar, o, thr, thr_lo, zero_hits, mx);  // i < j
                    }
                }
                if (hit && n >= min_boxes) S.flag[ra] = 1;
                if (WANT_MAX) atomicMax(&S.mx[ra], (unsigned long long)__double_as_longlong(mx));
            } else if (counted) {
                int qn = 0;
                for (int32_t tj = 0; tj < n; tj += WCAP) {
                    const int32_t tn = (n - tj < WCAP) ? n - tj : WCAP;
                    wave_sync();
                    for (int32_t k = lane; k < tn; k += kWave) S.cf[k] = outward_f32(load_corners(box4, (int64_t)base + tj + k));
                    wave_sync();
                    for (int32_t ib = 0; ib < tj + tn - 1; ib += kWave) {  // uniform outer loop over 64-box groups of i
                        const int32_t i = ib + lane;
                        const bool have = i < tj + tn - 1;
                        const float4 mef = have ? outward_f32(load_corners(box4, (int64_t)base + i)) : make_float4(0.f, 0.f, 0.f, 0.f);
                        const int32_t j0 = (ib + 1 > tj) ? ib + 1 : tj;  // smallest partner any lane of the group needs
                        for (int32_t j = j0; j < tj + tn; ++j) {
                            const float4 o = S.cf[j - tj];
                            const unsigned long long m = __builtin_amdgcn_ballot_w64(have && j > i) & overlap_mask(mef, o);
                            if (m) {
                                if ((m >> lane) & 1ull) {
                                    const int slot = qn + __popcll(m & lt);
                                    S.qa[slot] = (uint32_t)i;
                                    S.qb[slot] = (uint32_t)j;
                                }
                                qn += __popcll(m);
                                if (qn >= kWave) {
                                    wave_sync();
                                    k2f_drain<WANT_MAX>(box4, (int64_t)base, S, qn - kWave, kWave, ra, min_boxes, thr, thr_lo);
                                    qn -= kWave;
                                    wave_sync();
                                }
                            }
                        }
                    }
                }
                if (qn > 0) {
                    wave_sync();
                    k2f_drain<WANT_MAX>(box4, (int64_t)base, S, 0, qn, ra, min_boxes, thr, thr_lo);
                }
            }
            wave_sync();
            ra += 1;
            continue;
        }
        const int rb = ra + taken;
        const int32_t nb = __builtin_amdgcn_readlane(my_off, rb) - base;

        // ---- rank the sub-tile's rows by size (largest first, stable) in registers -------------
        int rank = 0, sorted_start = 0;
        for (int r2 = ra; r2 < rb; ++r2) {
            const int32_t n2 = __builtin_amdgcn_readlane(my_n, r2);
            rank += (n2 > my_n || (n2 == my_n && r2 < lane)) ? 1 : 0;
        }
        for (int r2 = ra; r2 < rb; ++r2) {
            const int32_t n2 = __builtin_amdgcn_readlane(my_n, r2);
            const int rank2 = __builtin_amdgcn_readlane(rank, r2);
            sorted_start += (rank2 < rank) ? n2 : 0;
        }
        if (lane >= ra && lane < rb) S.sst[lane] = sorted_start;
        wave_sync();

        // ---- stage the sub-tile: outward-rounded f32 boxes, row id, processing order ---------------
        for (int32_t k = lane; k < nb; k += kWave) {
            const Corners v = load_corners(box4, (int64_t)base + k);
            S.cf[k] = outward_f32(v);
            int lo = ra, hi = rb;
            while (hi - lo > 1) {
                const int mid = (lo + hi) >> 1;
                if (S.off[mid] - base <= k) lo = mid; else hi = mid;
            }
            S.row[k] = (unsigned short)lo;
            S.perm[S.sst[lo] + (k - (S.off[lo] - base))] = (unsigned short)k;
            if (has_nan(v)) S.nan[lo] = 1;
        }
        wave_sync();

        // ---- pairs: boxes in descending trip count, 64 per pass --------------------------------------
        int qn = 0;
        for (int32_t qb0 = 0; qb0 < nb; qb0 += kWave) {  // uniform loop over passes
            const int32_t q = qb0 + lane;
            int32_t k = 0, rs = 0, n = 1, i = 0, trips = 0;
            int lr = ra;
            bool plain = false, ordered = false;
            if (q < nb) {
                k = S.perm[q];
                lr = S.row[k];
                rs = S.off[lr] - base;
                n = S.off[lr + 1] - S.off[lr];
                if (n >= 2 && (WANT_MAX || n >= min_boxes)) {
                    i = k - rs;
                    const int32_t half = n >> 1;
                    trips = ((n & 1) == 0 && i >= half) ? half - 1 : half;
                    ordered = zero_hits || S.nan[lr] != 0;
                    plain = !ordered;
                }
            }
            if (__any(ordered)) {
                if (ordered) {  // exact loop straight from global memory: NaN corner in the row, or thr <= 0
                    const Corners me = load_corners(box4, (int64_t)base + k);
                    const double me_ar = area_of(me);
                    const bool nan_row = S.nan[lr] != 0;
                    bool hit = false;
                    double mx = 0.0;
                    for (int32_t d = 1; d <= trips; ++d) {
                        int32_t j = i + d;
                        if (j >= n) j -= n;
                        const Corners o = load_corners(box4, (int64_t)base + rs + j);
                        if (nan_row)  // keep the reference's (i < j) argument order
                            hit |= (j > i) ? pair_hits<WANT_MAX, false>(me, o, me_ar, o, thr, thr_lo, zero_hits, mx)
                                           : pair_hits<WANT_MAX, false>(o, me, me_ar, o, thr, thr_lo, zero_hits, mx);
                        else
                            hit |= pair_hits<WANT_MAX, true>(me, o, me_ar, o, thr, thr_lo, zero_hits, mx);
                    }
                    if (hit && n >= min_boxes) S.flag[lr] = 1;
                    if (WANT_MAX) atomicMax(&S.mx[lr], (unsigned long long)__double_as_longlong(mx));
                }
            }
            if (!plain) { rs = 0; n = 1; i = 0; trips = 0; }  // idle lanes read slot 0 and never qualify
            if (__any(plain)) {
                const float4 mef = S.cf[plain ? k : 0];
                int32_t j = (i + 1 >= n) ? i + 1 - n : i + 1;
                float4 nxt = S.cf[rs + j];
                for (int32_t d = 1; __any(d <= trips); ++d) {
                    const float4 o = nxt;
                    const int32_t kj = rs + j;  // partner of this trip
                    j = (j + 1 >= n) ? 0 : j + 1;
                    nxt = S.cf[rs + j];
                    const unsigned long long m = __builtin_amdgcn_ballot_w64(d <= trips) & overlap_mask(mef, o);
                    if (m) {  // wave-uniform
                        if ((m >> lane) & 1ull) {
                            const int slot = qn + __popcll(m & lt);
                            S.qa[slot] = (uint32_t)k;
                            S.qb[slot] = (uint32_t)kj;
                        }
                        qn += __popcll(m);
                        if (qn >= kWave) {
                            wave_sync();
                            k2f_drain<WANT_MAX>(box4, (int64_t)base, S, qn - kWave, kWave, -1, min_boxes, thr, thr_lo);
                            qn -= kWave;
                            wave_sync();
                        }
                    }
                }
            }
        }
        if (qn > 0) {
            wave_sync();
            k2f_drain<WANT_MAX>(box4, (int64_t)base, S, 0, qn, -1, min_boxes, thr, thr_lo);
        }
        wave_sync();
        ra = rb;
    }
    if (lane < nr) {
        out_high[r0 + lane] = (uint8_t)(S.flag[lane] != 0);
        if (WANT_MAX) out_max[r0 + lane] = __longlong_as_double((long long)S.mx[lane]);
    }
}

}  // namespace dyd
