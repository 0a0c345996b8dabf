// k2_wave.h — device code of K2 (per-image all-pairs IoU >= thr flag), shared by k2_iou.hip and
// the fused K1+K2 kernel.  See k2_iou.hip for the mapping and the reference semantics
// (reference core/processor.py:328-339, :359-362, :368-376).
#pragma once

#include "dyd_common.h"

namespace dyd {

constexpr int K2_BLOCK = 256;
constexpr int K2_WAVES = K2_BLOCK / kWave;
constexpr int K2_WROWS = 16;   // image rows per wave
constexpr int K2_WCAP = 256;   // boxes staged in LDS per sub-tile (4 f64 columns = 8 KiB per wave)

struct Corners {
    double x1, y1, x2, y2;
};

// Rows of thousands of boxes.  The tile kernels keep a row on ONE wave: 1000 boxes take 1.4 ms there, 10 000 boxes 120 ms,
// 50 000 boxes 3 s.  A wave that meets a row of more than K2_BIG_ROW boxes therefore pushes it onto a small queue in device
// memory — q[0] = number of pushes, then (row, boxes to pair) per entry — and moves on; k2_big_rows_kernel, launched behind the
// main kernel on the same stream, spreads every queued row over the whole grid.  When the queue is full (K2_BIG_LIST rows) the
// row stays with its wave as before: a table with thousands of such rows keeps the tile kernel's waves busy anyway.
constexpr int32_t K2_BIG_ROW = 256;      // rows above this many boxes (the tile kernels hold 128 / 256 per tile)
constexpr int32_t K2_BIG_LIST = 2048;    // queue capacity
// The same queue carries a second list: rows of 65..256 boxes met by the SPARSE wave kernel (k12_wave.h), which has neither the
// registers nor the LDS to sort and sweep them (k2_sweep.h) and would pair them all against all from memory — two orders of magnitude
// above a row's fair share, so that 2 % of such rows in a table of small ones add 60 % to the launch.  The drain kernel sweeps them,
// one row per wave.  Layout of ONE queue (u64 words): [0] big pushes, [1] mid pushes, then (row, boxes) pairs of either list.
constexpr int32_t K2_MID_LIST = 1 << 20;   // 16 MB per queue; a table with more such rows than this is not sparse
constexpr size_t K2_BIGQ_MID0 = 2 + 2 * (size_t)K2_BIG_LIST;                       // first word of the mid list
constexpr size_t K2_BIGQ_BYTES = 8 * (K2_BIGQ_MID0 + 2 * (size_t)K2_MID_LIST);     // ONE queue (the context holds two, taking turns)

__device__ __forceinline__ bool k2_queue_push(unsigned long long *q, int which, int64_t row, int32_t n) {   // wave-uniform call
    if (!q) return false;
    unsigned long long idx = 0;
    if ((threadIdx.x & 63) == 0) idx = atomicAdd(&q[which], 1ull);
    idx = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(idx >> 32)) << 32) |
          (unsigned)__builtin_amdgcn_readfirstlane((int)idx);
    if (idx >= (unsigned long long)(which ? K2_MID_LIST : K2_BIG_LIST)) return false;
    if ((threadIdx.x & 63) == 0) {
        unsigned long long *e = q + (which ? K2_BIGQ_MID0 : 2) + 2 * idx;
        e[0] = (unsigned long long)row;
        e[1] = (unsigned long long)(unsigned)n;
    }
    return true;
}
__device__ __forceinline__ bool bigq_push(unsigned long long *q, int64_t row, int32_t n) { return k2_queue_push(q, 0, row, n); }
__device__ __forceinline__ bool midq_push(unsigned long long *q, int64_t row, int32_t n) { return k2_queue_push(q, 1, row, n); }
// A row of more than K2_BIG_ROW boxes leaves the main kernels: up to K2_MID_ROW boxes onto the mid list (sorted and swept by one
// wave of the drain kernel: 16 keys per lane), beyond onto the big list (all pairs spread over the grid).  true = the row is
// taken care of (queued, or nothing to do: too few boxes for a flag and no maximum wanted).
constexpr int32_t K2_MID_ROW = 1024;
template <bool WANT_MAX>
__device__ __forceinline__ bool k2_defer_row(unsigned long long *q, int64_t row, int32_t n, int32_t min_boxes, bool zero_hits) {
    if (n <= K2_BIG_ROW || !q) return false;
    if (n <= K2_MID_ROW && !zero_hits) {
        if (!WANT_MAX && (n < 2 || n < min_boxes)) return true;
        if (midq_push(q, row, n)) return true;
    }
    return bigq_push(q, row, n);
}

// per-wave LDS slice: WROWS image rows, at most WCAP boxes staged at a time
template <int WROWS, int WCAP>
struct alignas(16) WaveLdsT {
    double x1[WCAP], y1[WCAP], x2[WCAP], y2[WCAP];
    unsigned long long mx[WROWS];
    int32_t off[WROWS + 4];
    int32_t flag[WROWS];
    int32_t nan[WROWS];
    int32_t sst[WROWS];
    unsigned short row[WCAP];
    unsigned short perm[WCAP];
};
using WaveLds = WaveLdsT<K2_WROWS, K2_WCAP>;

// LDS hand-off between lanes of ONE wave: the hardware executes a wave's LDS operations in
// order, so only the compiler must be kept from moving accesses across this point.
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// extract_boxes :359-362 — builtin two-argument min/max: first argument unless the second is
// strictly better.
__device__ __forceinline__ Corners normalise(double2 a, double2 b) {
    Corners o;
    o.x1 = (b.x < a.x) ? b.x : a.x;
    o.y1 = (b.y < a.y) ? b.y : a.y;
    o.x2 = (b.x > a.x) ? b.x : a.x;
    o.y2 = (b.y > a.y) ? b.y : a.y;
    return o;
}
// the area expression of :336-337
__device__ __forceinline__ double area_of(const Corners &c) { return (c.x2 - c.x1) * (c.y2 - c.y1); }
__device__ __forceinline__ bool has_nan(const Corners &c) {
    return c.x1 != c.x1 || c.y1 != c.y1 || c.x2 != c.x2 || c.y2 != c.y2;
}

// Single-instruction IEEE maxNum / minNum.  __builtin_fmax would do, but hipcc (ROCm 7.2) puts a
// canonicalising v_max_f64 x,x in front of every operand that comes from memory, doubling the
// instruction count of the pair loop; the asm form issues exactly one VALU op.
__device__ __forceinline__ double vmax(double a, double b) {
    double r;
    asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ double vmin(double a, double b) {
    double r;
    asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ double vmax0(double a) {
    double r;
    asm("v_max_f64 %0, %1, 0" : "=v"(r) : "v"(a));
    return r;
}

// calculate_iou :328-339 for p = lower-index box, q = higher-index box.  `me_ar` is the area of
// the lane's own box (either p or q), `oth` the partner whose area is only needed once the
// boxes intersect.  Returns true iff IoU >= thr; when WANT_MAX also folds the exact IoU into mx.
//
// NO_NAN: none of the eight corners is NaN.  Then CPython's first-wins max(a, b) / min(a, b)
// and max(0, v) return the same VALUE as IEEE maxNum / minNum (they can differ only in the sign
// of a zero, which cannot change w*h == 0, the sums or the quotient), so one v_max_f64 /
// v_min_f64 replaces each compare + select pair; max(0, NaN) = 0 also holds for maxNum when
// inf - inf produces a NaN width.  With a NaN corner the exact compare/select order is kept.
template <bool WANT_MAX, bool NO_NAN>
__device__ __forceinline__ bool pair_hits(const Corners &p, const Corners &q, double me_ar,
                                          const Corners &oth, double thr, double thr_lo, bool zero_hits,
                                          double &mx) {
    double ix1, iy1, ix2, iy2, w, h;
    if (NO_NAN) {
        ix1 = vmax(p.x1, q.x1);
        iy1 = vmax(p.y1, q.y1);
        ix2 = vmin(p.x2, q.x2);
        iy2 = vmin(p.y2, q.y2);
        w = vmax0(ix2 - ix1);
        h = vmax0(iy2 - iy1);
    } else {
        ix1 = (q.x1 > p.x1) ? q.x1 : p.x1;
        iy1 = (q.y1 > p.y1) ? q.y1 : p.y1;
        ix2 = (q.x2 < p.x2) ? q.x2 : p.x2;
        iy2 = (q.y2 < p.y2) ? q.y2 : p.y2;
        w = ix2 - ix1;
        h = iy2 - iy1;
        w = (w > 0.0) ? w : 0.0;
        h = (h > 0.0) ? h : 0.0;
    }
    const double inter = w * h;
    if (inter == 0.0) return zero_hits;  // :334-335 -> 0.0
    const double uni = me_ar + area_of(oth) - inter;  // area1 + area2 - inter; IEEE + commutes
    if (!WANT_MAX) {
        // certainly below the threshold: skip the division.  thr_lo = thr * 0.999 (0 when thr <= 0,
        // which disables the shortcut); the 1e-3 margin dwarfs every rounding involved.
        if (inter < thr_lo * uni) return false;
    }
    const double iou = (uni != 0.0) ? inter / uni : 0.0;
    if (WANT_MAX) {
        if (iou > mx) mx = iou;
    }
    return iou >= thr;
}

// One wave, rows [r0, r0 + nr) (nr <= K2_WROWS), private LDS slice S.  No workgroup barrier.
// `box4` may have been written earlier in the same kernel by this workgroup (fused path), so it
// is not declared __restrict__/const-cached here.
template <bool WANT_MAX, int WROWS = K2_WROWS, int WCAP = K2_WCAP>
__device__ __forceinline__ void k2_wave_rows(const double *box4, const int32_t *__restrict__ row_off, int64_t r0,
                                             int nr, int32_t min_boxes, double thr, uint8_t *__restrict__ out_high,
                                             double *__restrict__ out_max, WaveLdsT<WROWS, WCAP> &S,
                                             unsigned long long *bigq = nullptr) {
    static_assert(WROWS < 63 && WCAP <= 65535, "rows map to lanes, box ids to 16 bits");
    const int lane = threadIdx.x & 63;
    // lane L (L <= nr) keeps row_off[r0 + L] in a register and in LDS
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
    const int32_t my_n = __shfl_down(my_off, 1) - my_off;  // size of row L for L < nr
    wave_sync();
    const bool zero_hits = (0.0 >= thr);  // an empty intersection yields IoU 0.0 (:334-335)
    const double thr_lo = (thr > 0.0) ? thr * 0.999 : 0.0;

    int ra = 0;
    while (ra < nr) {  // every condition below is wave-uniform
        const int32_t base = __builtin_amdgcn_readlane(my_off, ra);
        // rows ra .. rb-1 fit the LDS tile together: offsets are monotone, so the qualifying lanes
        // are contiguous and their count is the number of rows taken
        const unsigned long long fits = __ballot(lane > ra && lane <= nr && my_off - base <= WCAP);
        const int taken = __popcll(fits);
        if (taken == 0) {
            // ---- one row larger than the LDS tile: stream partner tiles through LDS ----------
            const int32_t n = __builtin_amdgcn_readlane(my_off, ra + 1) - base;
            if (k2_defer_row<WANT_MAX>(bigq, r0 + ra, n, min_boxes, zero_hits)) {   // left to k2_big_rows_kernel
                ra += 1;
                continue;
            }
            const bool counted = WANT_MAX || n >= min_boxes;
            bool hit = false;
            double mx = 0.0;
            for (int32_t tj = 0; tj < n && counted; tj += WCAP) {
                const int32_t tn = (n - tj < WCAP) ? n - tj : WCAP;
                wave_sync();
                for (int32_t k = lane; k < tn; k += kWave) {
                    const double2 *g = reinterpret_cast<const double2 *>(box4 + 4 * (int64_t)(base + tj + k));
                    const Corners v = normalise(g[0], g[1]);
                    S.x1[k] = v.x1; S.y1[k] = v.y1; S.x2[k] = v.x2; S.y2[k] = v.y2;
                }
                wave_sync();
                for (int32_t i = lane; i < tj + tn - 1; i += kWave) {
                    const double2 *g = reinterpret_cast<const double2 *>(box4 + 4 * (int64_t)(base + i));
                    const Corners me = normalise(g[0], g[1]);
                    const double me_ar = area_of(me);
                    for (int32_t j = (i + 1 > tj) ? i + 1 : tj; j < tj + tn; ++j) {
                        const int32_t k = j - tj;
                        const Corners o = {S.x1[k], S.y1[k], S.x2[k], S.y2[k]};
                        hit |= pair_hits<WANT_MAX, false>(me, o, me_ar, o, thr, thr_lo, zero_hits, mx);  // i < j
                    }
                }
            }
            if (hit && n >= min_boxes) S.flag[ra] = 1;
            if (WANT_MAX) atomicMax(&S.mx[ra], (unsigned long long)__double_as_longlong(mx));
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

        // ---- stage the sub-tile's boxes: normalised corners as SoA columns in LDS -------------
        for (int32_t k = lane; k < nb; k += kWave) {
            const double2 *g = reinterpret_cast<const double2 *>(box4 + 4 * (int64_t)(base + k));
            const Corners v = normalise(g[0], g[1]);
            S.x1[k] = v.x1; S.y1[k] = v.y1; S.x2[k] = v.x2; S.y2[k] = v.y2;
            int lo = ra, hi = rb;  // the row r in [ra, rb) with off[r] - base <= k < off[r+1] - base
            while (hi - lo > 1) {
                const int mid = (lo + hi) >> 1;
                if (S.off[mid] - base <= k) lo = mid; else hi = mid;
            }
            S.row[k] = (unsigned short)lo;
            S.perm[S.sst[lo] + (k - (S.off[lo] - base))] = (unsigned short)k;
            if (has_nan(v)) S.nan[lo] = 1;
        }
        wave_sync();

        // ---- pairs: boxes are taken in descending trip count, 64 per pass -----------------------
        for (int32_t q = lane; q < nb; q += kWave) {
            const int32_t k = S.perm[q];
            const int lr = S.row[k];
            const int32_t rs = S.off[lr] - base;
            const int32_t n = S.off[lr + 1] - S.off[lr];
            if (n < 2 || (!WANT_MAX && n < min_boxes)) continue;
            const int32_t i = k - rs;
            const Corners me = {S.x1[k], S.y1[k], S.x2[k], S.y2[k]};
            const double me_ar = area_of(me);
            const int32_t half = n >> 1;
            const int32_t trips = ((n & 1) == 0 && i >= half) ? half - 1 : half;
            bool hit = false;
            double mx = 0.0;
            if (S.nan[lr] == 0) {
                int32_t j = (i + 1 >= n) ? i + 1 - n : i + 1;
                Corners nxt = {S.x1[rs + j], S.y1[rs + j], S.x2[rs + j], S.y2[rs + j]};
                for (int32_t d = 1; d <= trips; ++d) {
                    const Corners o = nxt;
                    j = (j + 1 >= n) ? 0 : j + 1;
                    const int32_t kj = rs + j;
                    nxt.x1 = S.x1[kj]; nxt.y1 = S.y1[kj]; nxt.x2 = S.x2[kj]; nxt.y2 = S.y2[kj];
                    hit |= pair_hits<WANT_MAX, true>(me, o, me_ar, o, thr, thr_lo, zero_hits, mx);
                }
            } else {  // a NaN in the row: keep the reference's (i < j) argument order
                for (int32_t d = 1; d <= trips; ++d) {
                    int32_t j = i + d;
                    if (j >= n) j -= n;
                    const int32_t kj = rs + j;
                    const Corners o = {S.x1[kj], S.y1[kj], S.x2[kj], S.y2[kj]};
                    hit |= (j > i) ? pair_hits<WANT_MAX, false>(me, o, me_ar, o, thr, thr_lo, zero_hits, mx)
                                   : pair_hits<WANT_MAX, false>(o, me, me_ar, o, thr, thr_lo, zero_hits, mx);
                }
            }
            if (hit && n >= min_boxes) S.flag[lr] = 1;
            if (WANT_MAX) atomicMax(&S.mx[lr], (unsigned long long)__double_as_longlong(mx));
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
