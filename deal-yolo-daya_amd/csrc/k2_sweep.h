// k2_sweep.h — K2 for a row of many boxes: sort by the left edge, then only look at the partners that can still reach thr.
//
// The pair loops of k2_wave.h / k2_filter.h visit all n(n-1)/2 pairs of a row (reference core/processor.py:368-376,
// `any(calculate_iou(...) >= thr ...)` over i < j); for a row of 256 boxes that is 32 640 reject tests, 510 trips per lane,
// and the table of configs[4] (256 boxes per image) is bound by exactly that loop.  Most of those pairs cannot hit:
//   IoU >= thr  =>  inter >= thr * union >= thr * area_a  and  inter_h <= h_a  =>  inter_w >= thr * w_a
//   inter_w <= x2_a - x1_b                                                      =>  x1_b <= x2_a - thr * w_a =: lim_a
// for EITHER box of the pair as `a` (no assumption on which one lies further left).  So with the row's boxes ordered by x1,
// box a only has to meet the boxes after it while their x1 stays <= lim_a — for thr = 0.98 a window of 2 % of its width.
// The test is only a filter: what passes it (and a y-overlap test) is queued and decided by the exact f64 code
// (k2f_drain), so rounding in the filter may only ever ADMIT pairs.  Hence: thr_lo = 0.999 thr instead of thr (the
// margin pair_hits already uses for its division shortcut), x1 rounded down and lim rounded up to f32, the low 8 bits
// of the (order-preserving) key given to the box index — x1 is truncated DOWN by up to 255 ulp, lim pushed UP by
// 256..511 ulp.  A row holding a corner that is not finite, or whose lim overflows, is left to the all-pairs code.
// For the diagnostic maximum (WANT_MAX) the window is the overlap window, lim_a = x2_a: every pair with a non-empty
// intersection is evaluated, the others contribute the 0.0 the maximum starts from.
//
// LDS: the wave's float4 tile (k2_filter.h) is reused as sorted keys u32[WCAP] | limits u32[WCAP] | (y1, y2) f32[WCAP][2].
// Sort: bitonic network over the next power of two (padding keys 0xffffffff sort last and pass no window).
#pragma once

namespace dyd {

constexpr int32_t K2S_MIN = 96;   // rows from this many boxes on are swept (below, the sort costs more than the pairs)

// f32 bit pattern -> u32 that orders like the value (-inf < ... < -0 < +0 < ... < +inf)
__device__ __forceinline__ uint32_t f32_order(float f) {
    const uint32_t b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

template <bool WANT_MAX, int WROWS, int WCAP>
__device__ __forceinline__ bool k2s_row(const double *box4, int64_t base, int32_t n, int row, WaveLdsF<WROWS, WCAP> &S,
                                        int32_t min_boxes, double thr, double thr_lo) {
    static_assert(WCAP <= 256 && WCAP >= 64 && (WCAP & (WCAP - 1)) == 0, "box index lives in 8 key bits; bitonic size");
    static_assert(sizeof(S.cf) >= 16 * (size_t)WCAP, "keys + limits + y intervals alias the float4 tile");
    const int lane = threadIdx.x & 63;
    const unsigned long long lt = lane ? (~0ull >> (64 - lane)) : 0ull;
    uint32_t *skey = reinterpret_cast<uint32_t *>(S.cf);
    uint32_t *slim = skey + WCAP;
    float2 *syy = reinterpret_cast<float2 *>(skey + 2 * WCAP);
    int P = 64;
    while (P < n) P <<= 1;   // n <= WCAP, a power of two
    const double tl = WANT_MAX ? 0.0 : thr_lo;

    // ---- keys, limits, y intervals ------------------------------------------------------------------------------
    bool bad = false;
    wave_sync();
    for (int32_t k = lane; k < P; k += kWave) {
        uint32_t key = 0xffffffffu;
        if (k < n) {
            const Corners c = load_corners(box4, base + k);
            const double lim = c.x2 - tl * (c.x2 - c.x1);
            const double probe = (c.x1 - c.x1) + (c.y1 - c.y1) + (c.x2 - c.x2) + (c.y2 - c.y2) + (lim - lim);   // 0 iff all are finite
            bad |= !(probe == 0.0);
            key = (f32_order(f32_below(c.x1)) & ~0xffu) | (uint32_t)k;
            slim[k] = (f32_order(f32_above(lim)) + 256u) | 0xffu;   // finite lim: at most 0xff7fffff + 256, no wrap
            syy[k] = make_float2(f32_below(c.y1), f32_above(c.y2));
        }
        skey[k] = key;
    }
    if (__any(bad)) {
        wave_sync();
        return false;
    }

    // ---- bitonic sort of the keys ------------------------------------------------------------------------------
    for (int k = 2; k <= P; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            wave_sync();
            for (int t = lane; t < (P >> 1); t += kWave) {
                const int a = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                const int b = a | j;
                const uint32_t ka = skey[a], kb = skey[b];
                const uint32_t lo = ka < kb ? ka : kb, hi = ka < kb ? kb : ka;
                const bool up = (a & k) == 0;
                skey[a] = up ? lo : hi;
                skey[b] = up ? hi : lo;
            }
        }
    }
    wave_sync();

    // ---- sweep: lane = sorted position p, partners p+1, p+2, ... while inside the window ---------------------------
    int qn = 0;
    bool done = false;
    for (int32_t p0 = 0; p0 < n - 1 && !done; p0 += kWave) {
        const int32_t p = p0 + lane;
        const bool have = p < n - 1;
        const int ia = (int)(skey[have ? p : 0] & 0xffu);
        const uint32_t lim = have ? slim[ia] : 0u;
        const float2 my = syy[ia];
        for (int32_t d = 1;; ++d) {
            const int32_t j = p + d;
            const uint32_t kj = (j < P) ? skey[j] : 0xffffffffu;
            const bool inwin = have && kj <= lim;
            if (!__any(inwin)) break;
            bool cand = false;
            const int ib = (int)(kj & 0xffu);
            if (inwin) {
                const float2 o = syy[ib];
                cand = my.y > o.x && o.y > my.x;   // y intervals overlap (outward-rounded, so never a false reject)
            }
            const unsigned long long m = __ballot(cand);
            if (m) {
                if (cand) {
                    const int slot = qn + __popcll(m & lt);
                    S.qa[slot] = (uint32_t)ia;
                    S.qb[slot] = (uint32_t)ib;
                }
                qn += __popcll(m);
                if (qn >= kWave) {
                    wave_sync();
                    k2f_drain<WANT_MAX>(box4, base, S, qn - kWave, kWave, row, min_boxes, thr, thr_lo);
                    qn -= kWave;
                    wave_sync();
                    if (!WANT_MAX && __builtin_amdgcn_readfirstlane(S.flag[row]) != 0) {   // any() is decided
                        done = true;
                        qn = 0;
                        break;
                    }
                }
            }
        }
    }
    if (qn > 0) {
        wave_sync();
        k2f_drain<WANT_MAX>(box4, base, S, 0, qn, row, min_boxes, thr, thr_lo);
    }
    wave_sync();
    return true;
}

}  // namespace dyd
