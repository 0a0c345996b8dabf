// k2_iou.hip — K2: per-image box-count + all-pairs IoU >= threshold flag.
//
// Replaces meet_conditions (reference core/processor.py:368-376), calculate_iou (:328-339) and
// the corner normalisation of extract_boxes (:359-362).
//
// Layout in HBM: box4 = B x (p1x,p1y,p2x,p2y) f64 (32 B, 16-B aligned), row_off = N+1 int32 box
// offsets per image row; out_high = N bytes.  Algorithmic bytes per launch:
// 32*B + 4*(N+1) + N; algorithmic flops 22 * sum n_i(n_i-1)/2 (f64).  Bound: f64 VALU issue
// (compare / select / divide per pair — not a contraction, so no MFMA), HBM only as a floor.
//
// Mapping (wave-autonomous, no workgroup barrier anywhere): every 64-lane wave owns K2_WROWS
// consecutive image rows and a private 9.6 KiB LDS slice, so 16 waves per CU run independently
// and one wave's HBM/LDS latency is covered by the others' arithmetic.  A wave walks its rows in
// sub-tiles of whole rows holding <= K2_WCAP boxes:
//   1. boxes are loaded once (coalesced 16-B lanes), corner-normalised with first-wins min/max
//      and staged as four SoA columns in LDS;
//   2. the sub-tile's rows are ranked by size with v_readlane compares (no LDS, no barrier) and
//      boxes are handed to lanes in descending trip count, so the 64 lanes of a pass run the
//      same number of trips;
//   3. a lane owns box i and visits partners j = i+d (mod n), d = 1..n/2 — every unordered pair
//      exactly once — reading the partner's corners from LDS (consecutive lanes -> consecutive
//      addresses, conflict-free); the partner of trip d+1 is fetched before trip d is evaluated.
// calculate_iou is symmetric in its two boxes unless a coordinate is NaN (first-wins max/min
// keep a NaN only from the first argument), so rows holding a NaN take an ordered loop that
// always evaluates (lower index, higher index); all other rows use v_max_f64 / v_min_f64.
// Arithmetic is f64 in the reference's operation order, built with -ffp-contract=off; the IEEE
// division is only executed when a conservative bound (inter < 0.999*thr*union) cannot already
// rule the pair out.
#include "dyd_common.h"

namespace dyd {

constexpr int K2_BLOCK = 256;
constexpr int K2_WAVES = K2_BLOCK / kWave;
constexpr int K2_WROWS = 16;   // image rows per wave
constexpr int K2_WCAP = 256;   // boxes staged in LDS per sub-tile (4 f64 columns = 8 KiB per wave)

struct Corners {
    double x1, y1, x2, y2;
};

struct alignas(16) WaveLds {
    double x1[K2_WCAP], y1[K2_WCAP], x2[K2_WCAP], y2[K2_WCAP];
    unsigned long long mx[K2_WROWS];
    int32_t off[K2_WROWS + 4];
    int32_t flag[K2_WROWS];
    int32_t nan[K2_WROWS];
    int32_t sst[K2_WROWS];
    unsigned short row[K2_WCAP];
    unsigned short perm[K2_WCAP];
};

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

template <bool WANT_MAX>
__global__ __launch_bounds__(K2_BLOCK) void k2_iou_kernel(const double *__restrict__ box4,
                                                          const int32_t *__restrict__ row_off,
                                                          int64_t n_rows, int32_t min_boxes, double thr,
                                                          uint8_t *__restrict__ out_high,
                                                          double *__restrict__ out_max) {
    __shared__ WaveLds s_all[K2_WAVES];
    const int lane = threadIdx.x & 63;
    WaveLds &S = s_all[threadIdx.x >> 6];
    const int64_t r0 = ((int64_t)blockIdx.x * K2_WAVES + (threadIdx.x >> 6)) * K2_WROWS;
    if (r0 >= n_rows) return;  // whole wave leaves; no workgroup barrier exists in this kernel
    const int nr = (n_rows - r0 < K2_WROWS) ? (int)(n_rows - r0) : K2_WROWS;

    // lane L (L <= nr) keeps row_off[r0 + L] in a register and in LDS
    int32_t my_off = 0;
    if (lane <= nr) {
        my_off = row_off[r0 + lane];
        S.off[lane] = my_off;
    }
    if (lane < K2_WROWS) {
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
        const unsigned long long fits = __ballot(lane > ra && lane <= nr && my_off - base <= K2_WCAP);
        const int taken = __popcll(fits);
        if (taken == 0) {
            // ---- one row larger than the LDS tile: stream partner tiles through LDS ----------
            const int32_t n = __builtin_amdgcn_readlane(my_off, ra + 1) - base;
            const bool counted = WANT_MAX || n >= min_boxes;
            bool hit = false;
            double mx = 0.0;
            for (int32_t tj = 0; tj < n && counted; tj += K2_WCAP) {
                const int32_t tn = (n - tj < K2_WCAP) ? n - tj : K2_WCAP;
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

int launch_k2(const double *box4, const int32_t *row_off, int64_t n_rows, int32_t min_boxes, double thr,
              uint8_t *out_high, double *out_max, hipStream_t st) {
    if (n_rows == 0) return DYD_OK;
    const int64_t blocks = ceil_div(n_rows, (int64_t)K2_WAVES * K2_WROWS);
    if (blocks > 0x7fffffffLL) {
        set_error("n_rows=%lld exceeds one launch", (long long)n_rows);
        return DYD_ERR_RANGE;
    }
    if (out_max)
        hipLaunchKernelGGL(k2_iou_kernel<true>, dim3((unsigned)blocks), dim3(K2_BLOCK), 0, st, box4, row_off,
                           n_rows, min_boxes, thr, out_high, out_max);
    else
        hipLaunchKernelGGL(k2_iou_kernel<false>, dim3((unsigned)blocks), dim3(K2_BLOCK), 0, st, box4, row_off,
                           n_rows, min_boxes, thr, out_high, out_max);
    DYD_HIP(hipGetLastError());
    return DYD_OK;
}

}  // namespace dyd

using namespace dyd;

extern "C" {

int dyd_iou_any_ge_dev(const double *box4, const int32_t *row_off, int64_t n_rows, int32_t min_boxes,
                       double thr, uint8_t *out_high, double *out_max_iou_or_null, void *stream) {
    DYD_API_ENTER();
    DYD_REQUIRE(n_rows >= 0, "n_rows < 0");
    if (n_rows == 0) return DYD_OK;
    DYD_REQUIRE(row_off && out_high, "null pointer");
    DYD_REQUIRE((reinterpret_cast<uintptr_t>(box4) & 15) == 0, "box4 must be 16-byte aligned");
    return launch_k2(box4, row_off, n_rows, min_boxes, thr, out_high, out_max_iou_or_null, pick_stream(stream));
}

int dyd_iou_any_ge(const double *box4, const int32_t *row_off, int64_t n_rows, int32_t min_boxes, double thr,
                   uint8_t *out_high, double *out_max_iou_or_null) {
    DYD_API_ENTER();
    DYD_REQUIRE(n_rows >= 0, "n_rows < 0");
    if (n_rows == 0) return DYD_OK;
    DYD_REQUIRE(row_off && out_high, "null pointer");
    DYD_REQUIRE(row_off[0] == 0, "row_off[0] != 0");
    for (int64_t i = 0; i < n_rows; ++i) DYD_REQUIRE(row_off[i + 1] >= row_off[i], "row_off not monotone");
    const int64_t nb = row_off[n_rows];
    DYD_REQUIRE(nb == 0 || box4, "box4 is null");
    DevBuf d_box, d_off, d_high, d_max;
    int rc;
    if ((rc = d_box.alloc(32 * (size_t)nb)) || (rc = d_off.alloc(4 * (size_t)(n_rows + 1))) ||
        (rc = d_high.alloc((size_t)n_rows)))
        return rc;
    if (out_max_iou_or_null && (rc = d_max.alloc(8 * (size_t)n_rows))) return rc;
    hipStream_t st = ctx().stream;
    if (nb) DYD_HIP(hipMemcpyAsync(d_box.p, box4, 32 * (size_t)nb, hipMemcpyHostToDevice, st));
    DYD_HIP(hipMemcpyAsync(d_off.p, row_off, 4 * (size_t)(n_rows + 1), hipMemcpyHostToDevice, st));
    KernelTimer t(st);
    rc = launch_k2(d_box.as<double>(), d_off.as<int32_t>(), n_rows, min_boxes, thr, d_high.as<uint8_t>(),
                   out_max_iou_or_null ? d_max.as<double>() : nullptr, st);
    if (rc) return rc;
    t.finish();
    DYD_HIP(hipMemcpyAsync(out_high, d_high.p, (size_t)n_rows, hipMemcpyDeviceToHost, st));
    if (out_max_iou_or_null)
        DYD_HIP(hipMemcpyAsync(out_max_iou_or_null, d_max.p, 8 * (size_t)n_rows, hipMemcpyDeviceToHost, st));
    DYD_HIP(hipStreamSynchronize(st));
    return DYD_OK;
}

}  // extern "C"
