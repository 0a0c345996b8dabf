// k2_iou.hip — K2: per-image box-count + all-pairs IoU >= threshold flag.
//
// Replaces meet_conditions (reference core/processor.py:368-376), calculate_iou (:328-339) and
// the corner normalisation of extract_boxes (:359-362).
//
// Layout in HBM: box4 = B x (p1x,p1y,p2x,p2y) f64 (32 B, 16-B aligned), row_off = N+1 int32 box
// offsets per image row; out_high = N bytes.  Algorithmic bytes per launch:
// 32*B + 4*(N+1) + N; algorithmic flops 22 * sum n_i(n_i-1)/2 (f64).  Bound: HBM for
// n_i <~ 32, f64 VALU for dense rows (256 boxes/row: ~80 flop/B) — no MFMA: the pair test is
// compare/select/divide, not a contraction.
//
// Mapping: a 256-thread workgroup owns K2_ROWS consecutive rows.  It walks them in sub-tiles of
// whole rows holding at most K2_CAP boxes; a sub-tile's boxes are loaded once (coalesced 16-B
// lanes), corner-normalised, and staged with their areas as SoA columns in LDS.  Lane t then
// owns box i of its row and visits partners j = i+d (mod n), d = 1..n/2 — every unordered pair
// exactly once, the same trip count for all lanes of a row — reading the partner's five columns
// from LDS (consecutive lanes -> consecutive addresses, conflict-free).  calculate_iou is
// symmetric in its two boxes unless a coordinate is NaN (first-wins max/min keep a NaN only
// from the first argument), so rows holding a NaN take an ordered loop that always evaluates
// (lower index, higher index); all other rows skip the operand swap.  Arithmetic is f64 in the reference's operation order, built with
// -ffp-contract=off; the IEEE division is only executed when a conservative bound
// (inter < thr*union*0.999) cannot already rule the pair out.
#include "dyd_common.h"

namespace dyd {

constexpr int K2_BLOCK = 256;
constexpr int K2_ROWS = 32;    // rows per workgroup
constexpr int K2_CAP = 1024;   // boxes staged in LDS per sub-tile (5 f64 columns = 40 KiB)

struct NBox {
    double x1, y1, x2, y2, ar;
};

// extract_boxes :359-362 — builtin two-argument min/max: first argument unless the second is
// strictly better — followed by the area expression of :336.
__device__ __forceinline__ NBox normalise(double2 a, double2 b) {
    NBox o;
    o.x1 = (b.x < a.x) ? b.x : a.x;
    o.y1 = (b.y < a.y) ? b.y : a.y;
    o.x2 = (b.x > a.x) ? b.x : a.x;
    o.y2 = (b.y > a.y) ? b.y : a.y;
    o.ar = (o.x2 - o.x1) * (o.y2 - o.y1);
    return o;
}

// calculate_iou :328-339 for p = lower-index box, q = higher-index box.  Returns true iff
// IoU >= thr; when WANT_MAX also folds the exact IoU into mx.
template <bool WANT_MAX>
__device__ __forceinline__ bool pair_hits(const NBox &p, const NBox &q, double thr, bool zero_hits,
                                          double &mx) {
    const double ix1 = (q.x1 > p.x1) ? q.x1 : p.x1;
    const double iy1 = (q.y1 > p.y1) ? q.y1 : p.y1;
    const double ix2 = (q.x2 < p.x2) ? q.x2 : p.x2;
    const double iy2 = (q.y2 < p.y2) ? q.y2 : p.y2;
    double w = ix2 - ix1, h = iy2 - iy1;
    w = (w > 0.0) ? w : 0.0;
    h = (h > 0.0) ? h : 0.0;
    const double inter = w * h;
    if (inter == 0.0) return zero_hits;  // :334-335 -> 0.0
    const double uni = p.ar + q.ar - inter;
    if (!WANT_MAX) {
        // certainly below the threshold: skip the division (margin 1e-3 >> any rounding)
        if (uni > 0.0 && thr > 0.0 && inter < thr * uni * 0.999) return false;
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
    __shared__ double sx1[K2_CAP], sy1[K2_CAP], sx2[K2_CAP], sy2[K2_CAP], sar[K2_CAP];
    __shared__ unsigned long long s_max[K2_ROWS];
    __shared__ int32_t s_off[K2_ROWS + 1];
    __shared__ int32_t s_flag[K2_ROWS];
    __shared__ int32_t s_nan[K2_ROWS];
    __shared__ unsigned short s_row[K2_CAP];

    const int tid = threadIdx.x;
    const int64_t r0 = (int64_t)blockIdx.x * K2_ROWS;
    const int nr = (n_rows - r0 < K2_ROWS) ? (int)(n_rows - r0) : K2_ROWS;
    if (tid <= nr) s_off[tid] = row_off[r0 + tid];
    if (tid < K2_ROWS) {
        s_flag[tid] = 0;
        s_nan[tid] = 0;
        s_max[tid] = 0ull;
    }
    __syncthreads();
    const bool zero_hits = (0.0 >= thr);  // an empty intersection yields IoU 0.0 (:334-335)

    int ra = 0;
    while (ra < nr) {  // all conditions below are workgroup-uniform
        const int32_t base = s_off[ra];
        int rb = ra + 1;
        const int32_t n_first = s_off[rb] - base;
        if (n_first > K2_CAP) {
            // ---- one row larger than the LDS tile: stream partner tiles through LDS ----------
            const int32_t n = n_first;
            const bool counted = WANT_MAX || n >= min_boxes;
            bool hit = false;
            double mx = 0.0;
            for (int32_t tj = 0; tj < n && counted; tj += K2_CAP) {
                const int32_t tn = (n - tj < K2_CAP) ? n - tj : K2_CAP;
                __syncthreads();
                for (int32_t k = tid; k < tn; k += K2_BLOCK) {
                    const double2 *p = reinterpret_cast<const double2 *>(box4 + 4 * (int64_t)(base + tj + k));
                    const NBox v = normalise(p[0], p[1]);
                    sx1[k] = v.x1; sy1[k] = v.y1; sx2[k] = v.x2; sy2[k] = v.y2; sar[k] = v.ar;
                }
                __syncthreads();
                for (int32_t i = tid; i < tj + tn - 1; i += K2_BLOCK) {
                    const double2 *p = reinterpret_cast<const double2 *>(box4 + 4 * (int64_t)(base + i));
                    const NBox me = normalise(p[0], p[1]);
                    for (int32_t j = (i + 1 > tj) ? i + 1 : tj; j < tj + tn; ++j) {
                        const int32_t k = j - tj;
                        const NBox o = {sx1[k], sy1[k], sx2[k], sy2[k], sar[k]};
                        hit |= pair_hits<WANT_MAX>(me, o, thr, zero_hits, mx);
                    }
                }
            }
            if (hit && n >= min_boxes) s_flag[ra] = 1;
            if (WANT_MAX) atomicMax(&s_max[ra], (unsigned long long)__double_as_longlong(mx));
            __syncthreads();
            ra = rb;
            continue;
        }
        while (rb < nr && s_off[rb + 1] - base <= K2_CAP) ++rb;
        const int32_t nb = s_off[rb] - base;

        // ---- stage the sub-tile's boxes: normalised corners + area, SoA in LDS --------------
        for (int32_t k = tid; k < nb; k += K2_BLOCK) {
            const double2 *p = reinterpret_cast<const double2 *>(box4 + 4 * (int64_t)(base + k));
            const NBox v = normalise(p[0], p[1]);
            sx1[k] = v.x1; sy1[k] = v.y1; sx2[k] = v.x2; sy2[k] = v.y2; sar[k] = v.ar;
            int lo = ra, hi = rb;  // largest row r in [ra, rb) with s_off[r] - base <= k
            while (hi - lo > 1) {
                const int mid = (lo + hi) >> 1;
                if (s_off[mid] - base <= k) lo = mid; else hi = mid;
            }
            s_row[k] = (unsigned short)lo;
            if (v.x1 != v.x1 || v.y1 != v.y1 || v.x2 != v.x2 || v.y2 != v.y2) s_nan[lo] = 1;
        }
        __syncthreads();

        // ---- pairs: lane owns box i, partners j = i+d (mod n), d = 1..n/2 ---------------------
        for (int32_t k = tid; k < nb; k += K2_BLOCK) {
            const int lr = s_row[k];
            const int32_t rs = s_off[lr] - base;
            const int32_t n = s_off[lr + 1] - s_off[lr];
            if (n < 2 || (!WANT_MAX && n < min_boxes)) continue;
            const int32_t i = k - rs;
            const NBox me = {sx1[k], sy1[k], sx2[k], sy2[k], sar[k]};
            const int32_t half = n >> 1;
            const int32_t trips = ((n & 1) == 0 && i >= half) ? half - 1 : half;
            bool hit = false;
            double mx = 0.0;
            if (s_nan[lr] == 0) {
                for (int32_t d = 1; d <= trips; ++d) {
                    int32_t j = i + d;
                    if (j >= n) j -= n;
                    const int32_t kj = rs + j;
                    const NBox o = {sx1[kj], sy1[kj], sx2[kj], sy2[kj], sar[kj]};
                    hit |= pair_hits<WANT_MAX>(me, o, thr, zero_hits, mx);
                }
            } else {  // a NaN in the row: keep the reference's (i < j) argument order
                for (int32_t d = 1; d <= trips; ++d) {
                    int32_t j = i + d;
                    if (j >= n) j -= n;
                    const int32_t kj = rs + j;
                    const NBox o = {sx1[kj], sy1[kj], sx2[kj], sy2[kj], sar[kj]};
                    hit |= (j > i) ? pair_hits<WANT_MAX>(me, o, thr, zero_hits, mx)
                                   : pair_hits<WANT_MAX>(o, me, thr, zero_hits, mx);
                }
            }
            if (hit && n >= min_boxes) s_flag[lr] = 1;
            if (WANT_MAX) atomicMax(&s_max[lr], (unsigned long long)__double_as_longlong(mx));
        }
        __syncthreads();
        ra = rb;
    }
    if (tid < nr) {
        out_high[r0 + tid] = (uint8_t)(s_flag[tid] != 0);
        if (WANT_MAX) out_max[r0 + tid] = __longlong_as_double((long long)s_max[tid]);
    }
}

int launch_k2(const double *box4, const int32_t *row_off, int64_t n_rows, int32_t min_boxes, double thr,
              uint8_t *out_high, double *out_max, hipStream_t st) {
    if (n_rows == 0) return DYD_OK;
    const int64_t blocks = ceil_div(n_rows, K2_ROWS);
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
