// k12_wave.h — wave-autonomous fused K1+K2: poly -> bbox -> IoU flag with the boxes handed from
// K1 to K2 through LDS (no re-read from memory) and no workgroup barrier.
//
// Every 64-lane wave owns KW_ROWS consecutive image rows and a private 4.4 KiB LDS slice, so a
// CU holds 32 independent waves.  A wave walks its rows in tiles of whole rows with at most 64
// boxes, one box per lane:
//   K1  the tile's contiguous point range is streamed HBM -> LDS in KW_CHUNK-point pieces
//       (coalesced 16-B lanes); each lane walks its own box's points in order (first-wins
//       min/max, reference core/processor.py:252-260) and stores box4/arg4 to global memory;
//   K2  the same lanes write their corner-normalised box over the (now dead) point buffer and
//       run the all-pairs loop of k2_wave.h on it: lane i visits partners i+d (mod n), d = 1..n/2
//       (reference :328-339, :359-362, :368-376).
// A row with more than 64 boxes takes a slower general path: K1 in 64-box passes to global
// memory, then K2 streaming 64-box partner tiles through LDS.
// The DENSE instantiation (tables above 32 boxes per image on average) pairs rows of 40..256 boxes by sort and sweep
// (k2_sweep.h) instead: the sweep records (key, limit, y interval) are computed from the K1 accumulators while they are still in
// registers, the keys are sorted there, and only the exact tests read boxes back (from L2).  It costs ~20 VGPRs and 1 KiB of LDS
// per wave, which is why the kernel for sparse tables does not carry it.
//
// Chain semantics (reference :254-255 -> :364-365): a polygon without a valid point becomes a ptList of
// null coordinates in the replace step, and the IoU step's extract_boxes raises on it inside its blanket
// try — the row's box list is the PREFIX collected before that object.  K1 marks such a box with
// arg index -1; the pair stage below ends the row's list at the first of them (one ballot per tile, and
// nothing else unless a tile really holds an empty polygon).
#pragma once

#include "k1_tile.h"
#include "k2_filter.h"

namespace dyd {

#ifndef KW_ROWS_VALUE
#define KW_ROWS_VALUE 16
#endif
constexpr int KW_ROWS = KW_ROWS_VALUE;    // image rows per wave
constexpr int KW_CHUNK = 256;  // points per LDS piece (4 KiB)
// The sweep's trip budget (k2_sweep.h: give up on the x1 order after n trips, hand the row to the drain kernel's diagonal attempt) in the
// DENSE kernel: measured, not the default — it costs every dense table 2 % (configs[4]: 8.27-8.58 ms against 8.12-8.57, alternating on one
// box) to turn a single-column row's 640 trips into ~320.  The drain kernel, where registers are free, keeps it.  -DK12_BUDGET_ON to A/B.
#ifdef K12_BUDGET_ON
constexpr bool K12_BUDGET = true;
#else
constexpr bool K12_BUDGET = false;
#endif

struct alignas(16) WaveFuse {
    union {
        double2 pts[KW_CHUNK];
        struct {
            double x1[kWave], y1[kWave], x2[kWave], y2[kWave];
        } c;
    };
    int32_t off[KW_ROWS + 4];
    int32_t flag[KW_ROWS];
    int32_t nan[KW_ROWS];
};
// the DENSE kernel's slice: the same, plus the sweep's queue of candidate pairs; its sorted keys | limits | y intervals (4 KiB for 256
// boxes) lie over the point buffer, which is dead by then
struct alignas(16) WaveFuseDense : WaveFuse {
    uint32_t qa[2 * kWave], qb[2 * kWave];
};
static_assert(sizeof(double2) * KW_CHUNK >= 16 * 256, "sweep arrays of a 256-box row fit the point buffer");
__device__ __forceinline__ K2sView k12_sweep_view(WaveFuseDense &S) {
    uint32_t *skey = reinterpret_cast<uint32_t *>(S.pts);
    return K2sView{skey, skey + 256, reinterpret_cast<float2 *>(skey + 512), S.qa, S.qb};
}
__device__ __forceinline__ K2sView k12_sweep_view(WaveFuse &) { return K2sView{nullptr, nullptr, nullptr, nullptr, nullptr}; }

// K1 for up to 64 consecutive boxes [b0, b0 + cnt): lane l < cnt owns box b0 + l.  Returns the
// lane's accumulator (already stored to out_box4 / out_arg4).
__device__ __forceinline__ BoxAcc k12_wave_boxes(const double2 *__restrict__ xy, const int32_t *__restrict__ pt_off,
                                                 int64_t b0, int cnt, double *out_box4,
                                                 int32_t *__restrict__ out_arg4, WaveFuse &S) {
    const int lane = threadIdx.x & 63;
    int32_t s = 0, e = 0;
    if (lane < cnt) {
        s = pt_off[b0 + lane];
        e = pt_off[b0 + lane + 1];
    }
    const int32_t ts = __builtin_amdgcn_readlane(s, 0);
    const int32_t te = __builtin_amdgcn_readlane(e, cnt - 1);
    BoxAcc acc;
    acc.empty();
    for (int32_t cs = ts; cs < te; cs += KW_CHUNK) {
        const int32_t ce = (te - cs > KW_CHUNK) ? cs + KW_CHUNK : te;
        wave_sync();  // the buffer's previous content is fully consumed
        for (int32_t p = cs + lane; p < ce; p += kWave) S.pts[p - cs] = xy[p];
        wave_sync();
        int32_t lo = s > cs ? s : cs;
        const int32_t hi = e < ce ? e : ce;
        if (lo < hi) {
            if (lo == s) {
                const double2 v = S.pts[lo - cs];
                acc.first(v.x, v.y);
                ++lo;
            }
            for (int32_t p = lo; p < hi; ++p) {
                const double2 v = S.pts[p - cs];
                acc.next(v.x, v.y, p - s);
            }
        }
    }
    if (lane < cnt) acc.store<true>(out_box4, out_arg4, b0 + lane);
    return acc;
}

// BOXES_IN: the boxes already exist (K2 alone): out_box4 is then the INPUT and neither xy / pt_off nor out_arg4 are touched
template <bool BOXES_IN = false, bool DENSE = false, class Slice = WaveFuse>
__device__ __forceinline__ void k12_wave_rows(const double2 *__restrict__ xy, const int32_t *__restrict__ pt_off,
                                              const int32_t *__restrict__ box_off, int64_t r0, int nr,
                                              int32_t min_boxes, double thr, double *out_box4,
                                              int32_t *__restrict__ out_arg4, uint8_t *__restrict__ out_high,
                                              Slice &S, unsigned long long *bigq = nullptr) {
    const int lane = threadIdx.x & 63;
    int32_t my_off = 0;
    if (lane <= nr) {
        my_off = box_off[r0 + lane];
        S.off[lane] = my_off;
    }
    if (lane < KW_ROWS) {
        S.flag[lane] = 0;
        S.nan[lane] = 0;
    }
    wave_sync();
    const bool zero_hits = (0.0 >= thr);
    const double thr_lo = (thr > 0.0) ? thr * 0.999 : 0.0;
    double unused_mx = 0.0;

    int ra = 0;
    while (ra < nr) {  // every condition below is wave-uniform
        const int32_t base = __builtin_amdgcn_readlane(my_off, ra);
        const unsigned long long fits = __ballot(lane > ra && lane <= nr && my_off - base <= kWave);
        const int taken = __popcll(fits);
        if (taken == 0) {
            // ---- a row with more than 64 boxes: K1 in passes, then K2 over partner tiles ---------
            int32_t n = __builtin_amdgcn_readlane(my_off, ra + 1) - base;
            bool k1_done = false;
            if constexpr (DENSE) {
                if (n <= 256 && !zero_hits) {
                    // ---- 65..256 boxes: K1 passes with the sweep records taken from the accumulators, then sort and sweep ----
                    uint32_t vk[4], vl[4];
                    float2 vy[4];
                    unsigned badbits = 0u;
                    int32_t n_eff = n;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int32_t g = kWave * r;
                        vk[r] = 0xffffffffu;
                        vl[r] = 0u;
                        vy[r] = make_float2(0.f, 0.f);
                        if (g < n) {   // wave-uniform
                            const int cnt = (n - g < kWave) ? n - g : kWave;
                            Corners c = {0.0, 0.0, 0.0, 0.0};
                            if (BOXES_IN) {
                                if (lane < cnt) c = load_corners(out_box4, (int64_t)base + g + lane);
                            } else {
                                const BoxAcc a = k12_wave_boxes(xy, pt_off, (int64_t)base + g, cnt, out_box4, out_arg4, S);
                                const unsigned long long em = __ballot(lane < cnt && a.imnx < 0);
                                if (em != 0ull && g + (__ffsll((long long)em) - 1) < n_eff) n_eff = g + (__ffsll((long long)em) - 1);
                                c = normalise(make_double2(a.mnx, a.mny), make_double2(a.mxx, a.mxy));
                            }
                            if (lane < cnt && !k2s_prepare(c, (uint32_t)(g + lane), thr_lo, vk[r], vl[r], vy[r])) badbits |= 1u << r;
                        }
                    }
                    k1_done = true;
                    n = n_eff;   // the row's IoU list ends at its first empty polygon
                    if (n < 2 || n < min_boxes) {
                        ra += 1;
                        continue;
                    }
                    bool bad = false;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if (kWave * r + lane >= n) vk[r] = 0xffffffffu;
                        else bad |= ((badbits >> r) & 1u) != 0u;
                    }
                    if (!__any(bad)) {
                        // the exact tests re-read this wave's own box stores: wait for them and drop any stale L1 lines
                        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
                        const K2sView V = k12_sweep_view(S);
                        wave_sync();   // the point buffer is dead: limits and y intervals go over it
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int32_t k = kWave * r + lane;
                            if (k < n) {
                                V.slim[k] = vl[r];
                                V.syy[k] = vy[r];
                            }
                        }
                        // a budget of n trips: a row the x1 order cannot spread (a column of boxes) is handed to the drain kernel, which has
                        // the registers for a second attempt along the diagonal (k2s_retry_diag); a full queue: x1 order to the end
                        bool hit, ab = false;
                        const int32_t budget = bigq ? n : 0;
                        if (n <= 2 * kWave) {
                            uint32_t v2[2] = {vk[0], vk[1]};
                            hit = k2s_sweep_sorted<false, 2, 8, K12_BUDGET>(out_box4, (int64_t)base, n, V, v2, thr, thr_lo, unused_mx, budget, &ab);
                            if (ab && !midq_push(bigq, r0 + ra, n)) {
                                uint32_t w2[2] = {vk[0], vk[1]};
                                hit = k2s_sweep_sorted<false, 2>(out_box4, (int64_t)base, n, V, w2, thr, thr_lo, unused_mx);
                            }
                        } else {
                            uint32_t v4[4] = {vk[0], vk[1], vk[2], vk[3]};
                            hit = k2s_sweep_sorted<false, 4, 8, K12_BUDGET>(out_box4, (int64_t)base, n, V, v4, thr, thr_lo, unused_mx, budget, &ab);
                            if (ab && !midq_push(bigq, r0 + ra, n)) hit = k2s_sweep_sorted<false, 4>(out_box4, (int64_t)base, n, V, vk, thr, thr_lo, unused_mx);
                        }
                        if (hit && lane == 0) S.flag[ra] = 1;
                        wave_sync();
                        ra += 1;
                        continue;
                    }
                    // a corner that is not finite: the generic code below decides (K1 has run)
                }
            }
            if (!BOXES_IN && !k1_done) {
                int32_t n_eff = n;
                for (int32_t g = 0; g < n; g += kWave) {
                    const int cnt = (n - g < kWave) ? n - g : kWave;
                    const BoxAcc a = k12_wave_boxes(xy, pt_off, (int64_t)base + g, cnt, out_box4, out_arg4, S);
                    const unsigned long long em = __ballot(lane < cnt && a.imnx < 0);
                    if (em != 0ull && g + (__ffsll((long long)em) - 1) < n_eff) n_eff = g + (__ffsll((long long)em) - 1);
                }
                n = n_eff;  // the row's IoU list ends at its first empty polygon
            }
            if constexpr (!DENSE) {   // 65..256 boxes in the kernel for sparse tables: the drain kernel sorts and sweeps the row (k2_wave.h)
                if (n <= K2_BIG_ROW && n >= 2 && n >= min_boxes && !zero_hits && midq_push(bigq, r0 + ra, n)) {
                    ra += 1;
                    continue;
                }
            }
            if (k2_defer_row<false>(bigq, r0 + ra, n, min_boxes, zero_hits)) {   // hundreds or thousands of boxes: k2_big_rows_kernel (k2_wave.h)
                ra += 1;
                continue;
            }
            // this wave re-reads its own stores below: wait for them and drop any stale L1 lines
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
            bool hit = false;
            if (n >= min_boxes) {
                for (int32_t tj = 0; tj < n; tj += kWave) {
                    const int32_t tn = (n - tj < kWave) ? n - tj : kWave;
                    wave_sync();
                    if (lane < tn) {
                        const double2 *g2 = reinterpret_cast<const double2 *>(out_box4 + 4 * (int64_t)(base + tj + lane));
                        const Corners v = normalise(g2[0], g2[1]);
                        S.c.x1[lane] = v.x1; S.c.y1[lane] = v.y1; S.c.x2[lane] = v.x2; S.c.y2[lane] = v.y2;
                    }
                    wave_sync();
                    for (int32_t i = lane; i < tj + tn - 1; i += kWave) {
                        const double2 *g2 = reinterpret_cast<const double2 *>(out_box4 + 4 * (int64_t)(base + i));
                        const Corners me = normalise(g2[0], g2[1]);
                        const double me_ar = area_of(me);
                        for (int32_t j = (i + 1 > tj) ? i + 1 : tj; j < tj + tn; ++j) {
                            const int32_t k = j - tj;
                            const Corners o = {S.c.x1[k], S.c.y1[k], S.c.x2[k], S.c.y2[k]};
                            hit |= pair_hits<false, false>(me, o, me_ar, o, thr, thr_lo, zero_hits, unused_mx);
                        }
                    }
                }
            }
            if (hit) S.flag[ra] = 1;
            wave_sync();
            ra += 1;
            continue;
        }
        const int rb = ra + taken;
        const int32_t nb = __builtin_amdgcn_readlane(my_off, rb) - base;  // 0 .. 64 boxes in the tile
        if (nb > 0) {
            // ---- K1: one box per lane ---------------------------------------------------------
            BoxAcc acc;
            if (BOXES_IN) {
                acc.empty();
                if (lane < nb) {
                    const double2 *g2 = reinterpret_cast<const double2 *>(out_box4 + 4 * ((int64_t)base + lane));
                    const double2 lo2 = g2[0], hi2 = g2[1];
                    acc.mnx = lo2.x; acc.mny = lo2.y; acc.mxx = hi2.x; acc.mxy = hi2.y;
                }
            } else {
                acc = k12_wave_boxes(xy, pt_off, (int64_t)base, nb, out_box4, out_arg4, S);
            }
            // ---- hand-off: the lanes overwrite the dead point buffer with their normalised box ------
            int lr = ra;  // the tile row that holds box `lane`
            for (int r2 = ra + 1; r2 < rb; ++r2) lr += (__builtin_amdgcn_readlane(my_off, r2) - base <= lane) ? 1 : 0;
            const Corners me = normalise(make_double2(acc.mnx, acc.mny), make_double2(acc.mxx, acc.mxy));
            const unsigned long long em = BOXES_IN ? 0ull : __ballot(lane < nb && acc.imnx < 0);  // empty polygons of the tile
            if constexpr (DENSE) {
                if (taken == 1 && nb >= K2S_MIN && !zero_hits) {
                    // ---- one row of 40..64 boxes fills the tile: sort and sweep instead of 20..32 trips of all pairs ----
                    const int32_t n = (em != 0ull) ? (__ffsll((long long)em) - 1) : nb;   // the list ends at the first empty polygon
                    if (n < 2 || n < min_boxes) {
                        ra = rb;
                        continue;
                    }
                    uint32_t v1[1] = {0xffffffffu}, lim1 = 0u;
                    float2 yy1 = make_float2(0.f, 0.f);
                    const bool ok = lane >= n || k2s_prepare(me, (uint32_t)lane, thr_lo, v1[0], lim1, yy1);
                    if (lane >= n) v1[0] = 0xffffffffu;
                    if (!__any(!ok)) {
                        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");   // the exact tests read this wave's box stores back
                        const K2sView V = k12_sweep_view(S);
                        wave_sync();
                        if (lane < n) {
                            V.slim[lane] = lim1;
                            V.syy[lane] = yy1;
                        }
                        bool ab = false;
                        uint32_t w1[1] = {v1[0]};
                        bool hit = k2s_sweep_sorted<false, 1, 8, K12_BUDGET>(out_box4, (int64_t)base, n, V, v1, thr, thr_lo, unused_mx, bigq ? n : 0, &ab);
                        if (ab && !midq_push(bigq, r0 + ra, n)) hit = k2s_sweep_sorted<false, 1>(out_box4, (int64_t)base, n, V, w1, thr, thr_lo, unused_mx);
                        if (hit && lane == 0) S.flag[ra] = 1;
                        wave_sync();
                        ra = rb;
                        continue;
                    }
                }
            }
            wave_sync();
            if (lane < nb) {
                S.c.x1[lane] = me.x1; S.c.y1[lane] = me.y1; S.c.x2[lane] = me.x2; S.c.y2[lane] = me.y2;
                if (has_nan(me)) S.nan[lr] = 1;
            }
            wave_sync();
            // ---- K2: lane owns box i of its row, partners j = i+d (mod n), d = 1..n/2 ---------------
            if (lane < nb) {
                const int32_t rs = S.off[lr] - base;
                int32_t n = S.off[lr + 1] - S.off[lr];
                if (em != 0ull) {  // wave-uniform, rare: cut the row at its first empty polygon
                    const unsigned long long row_bits = ((n >= 64) ? ~0ull : ((1ull << n) - 1ull)) << rs;
                    const unsigned long long m = em & row_bits;
                    if (m != 0ull) n = (__ffsll((long long)m) - 1) - rs;
                }
                const int32_t i = lane - rs;
                if (n >= 2 && n >= min_boxes && i < n) {
                    const double me_ar = area_of(me);
                    const int32_t half = n >> 1;
                    const int32_t trips = ((n & 1) == 0 && i >= half) ? half - 1 : half;
                    bool hit = false;
                    if (S.nan[lr] == 0) {
                        // two partners per turn, each in its own registers: the next partner is loaded while the current one
                        // is tested, and no register copy is needed to hand it over (one load past the last trip at most; it
                        // stays inside the row)
                        int32_t j = (i + 1 >= n) ? i + 1 - n : i + 1;
                        Corners a = {S.c.x1[rs + j], S.c.y1[rs + j], S.c.x2[rs + j], S.c.y2[rs + j]};
                        int32_t d = 1;
                        for (; d < trips; d += 2) {
                            j = (j + 1 >= n) ? 0 : j + 1;
                            const int32_t kb = rs + j;
                            const Corners b = {S.c.x1[kb], S.c.y1[kb], S.c.x2[kb], S.c.y2[kb]};
                            hit |= pair_hits<false, true>(me, a, me_ar, a, thr, thr_lo, zero_hits, unused_mx);
                            j = (j + 1 >= n) ? 0 : j + 1;
                            const int32_t ka = rs + j;
                            a.x1 = S.c.x1[ka]; a.y1 = S.c.y1[ka]; a.x2 = S.c.x2[ka]; a.y2 = S.c.y2[ka];
                            hit |= pair_hits<false, true>(me, b, me_ar, b, thr, thr_lo, zero_hits, unused_mx);
                        }
                        if (d == trips) hit |= pair_hits<false, true>(me, a, me_ar, a, thr, thr_lo, zero_hits, unused_mx);
                    } else {  // a NaN in the row: keep the reference's (i < j) argument order
                        for (int32_t d = 1; d <= trips; ++d) {
                            int32_t j = i + d;
                            if (j >= n) j -= n;
                            const int32_t kj = rs + j;
                            const Corners o = {S.c.x1[kj], S.c.y1[kj], S.c.x2[kj], S.c.y2[kj]};
                            hit |= (j > i) ? pair_hits<false, false>(me, o, me_ar, o, thr, thr_lo, zero_hits, unused_mx)
                                           : pair_hits<false, false>(o, me, me_ar, o, thr, thr_lo, zero_hits, unused_mx);
                        }
                    }
                    if (hit) S.flag[lr] = 1;
                }
            }
            wave_sync();
        }
        ra = rb;
    }
    if (lane < nr) out_high[r0 + lane] = (uint8_t)(S.flag[lane] != 0);
}

}  // namespace dyd
