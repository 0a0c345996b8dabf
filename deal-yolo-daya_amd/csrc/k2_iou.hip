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
#include "k2_filter.h"

namespace dyd {

template <bool WANT_MAX, int WROWS, int WCAP>
__global__ __launch_bounds__(K2_BLOCK) void k2_iou_kernel(const double *__restrict__ box4,
                                                          const int32_t *__restrict__ row_off,
                                                          int64_t n_rows, int32_t min_boxes, double thr,
                                                          uint8_t *__restrict__ out_high,
                                                          double *__restrict__ out_max) {
    __shared__ WaveLdsT<WROWS, WCAP> s_all[K2_WAVES];
    const int wave = threadIdx.x >> 6;
    const int64_t r0 = ((int64_t)blockIdx.x * K2_WAVES + wave) * WROWS;
    if (r0 >= n_rows) return;  // whole wave leaves; no workgroup barrier exists in this kernel
    const int nr = (n_rows - r0 < WROWS) ? (int)(n_rows - r0) : WROWS;
    k2_wave_rows<WANT_MAX, WROWS, WCAP>(box4, row_off, r0, nr, min_boxes, thr, out_high, out_max, s_all[wave]);
}

// the same kernel with the f32 reject filter in front of the exact test (k2_filter.h)
template <bool WANT_MAX, int WROWS, int WCAP>
__global__ __launch_bounds__(K2_BLOCK) void k2f_iou_kernel(const double *__restrict__ box4,
                                                           const int32_t *__restrict__ row_off,
                                                           int64_t n_rows, int32_t min_boxes, double thr,
                                                           uint8_t *__restrict__ out_high,
                                                           double *__restrict__ out_max) {
    __shared__ WaveLdsF<WROWS, WCAP> s_all[K2_WAVES];
    const int wave = threadIdx.x >> 6;
    const int64_t r0 = ((int64_t)blockIdx.x * K2_WAVES + wave) * WROWS;
    if (r0 >= n_rows) return;
    const int nr = (n_rows - r0 < WROWS) ? (int)(n_rows - r0) : WROWS;
    k2f_wave_rows<WANT_MAX, WROWS, WCAP>(box4, row_off, r0, nr, min_boxes, thr, out_high, out_max, s_all[wave]);
}

template <int WROWS, int WCAP>
static int launch_k2f_t(const double *box4, const int32_t *row_off, int64_t n_rows, int32_t min_boxes, double thr,
                        uint8_t *out_high, double *out_max, hipStream_t st) {
    const int64_t blocks = ceil_div(n_rows, (int64_t)K2_WAVES * WROWS);
    if (blocks > 0x7fffffffLL) {
        set_error("n_rows=%lld exceeds one launch", (long long)n_rows);
        return DYD_ERR_RANGE;
    }
    if (out_max)
        hipLaunchKernelGGL((k2f_iou_kernel<true, WROWS, WCAP>), dim3((unsigned)blocks), dim3(K2_BLOCK), 0, st, box4,
                           row_off, n_rows, min_boxes, thr, out_high, out_max);
    else
        hipLaunchKernelGGL((k2f_iou_kernel<false, WROWS, WCAP>), dim3((unsigned)blocks), dim3(K2_BLOCK), 0, st, box4,
                           row_off, n_rows, min_boxes, thr, out_high, out_max);
    DYD_HIP(hipGetLastError());
    return DYD_OK;
}

template <int WROWS, int WCAP>
static int launch_k2_t(const double *box4, const int32_t *row_off, int64_t n_rows, int32_t min_boxes, double thr,
                       uint8_t *out_high, double *out_max, hipStream_t st) {
    const int64_t blocks = ceil_div(n_rows, (int64_t)K2_WAVES * WROWS);
    if (blocks > 0x7fffffffLL) {
        set_error("n_rows=%lld exceeds one launch", (long long)n_rows);
        return DYD_ERR_RANGE;
    }
    if (out_max)
        hipLaunchKernelGGL((k2_iou_kernel<true, WROWS, WCAP>), dim3((unsigned)blocks), dim3(K2_BLOCK), 0, st, box4,
                           row_off, n_rows, min_boxes, thr, out_high, out_max);
    else
        hipLaunchKernelGGL((k2_iou_kernel<false, WROWS, WCAP>), dim3((unsigned)blocks), dim3(K2_BLOCK), 0, st, box4,
                           row_off, n_rows, min_boxes, thr, out_high, out_max);
    DYD_HIP(hipGetLastError());
    return DYD_OK;
}

// tile variant: 0 = 16 rows / 256 boxes per wave (16 waves per CU), 1 = 8 rows / 128 boxes (32 waves per CU),
// 2 / 3 = the same two tilings with the f32 reject filter (k2_filter.h)
static int g_k2_variant = 3;  // default: 8-row tiles + f32 reject filter (best on dense rows, on par on sparse ones)
void set_k2_variant(int v) { g_k2_variant = v; }

int launch_k2(const double *box4, const int32_t *row_off, int64_t n_rows, int32_t min_boxes, double thr,
              uint8_t *out_high, double *out_max, hipStream_t st) {
    if (n_rows == 0) return DYD_OK;
    if (g_k2_variant == 0) return launch_k2_t<K2_WROWS, K2_WCAP>(box4, row_off, n_rows, min_boxes, thr, out_high, out_max, st);
    if (g_k2_variant == 1) return launch_k2_t<8, 128>(box4, row_off, n_rows, min_boxes, thr, out_high, out_max, st);
    if (g_k2_variant == 2) return launch_k2f_t<16, 256>(box4, row_off, n_rows, min_boxes, thr, out_high, out_max, st);
    if (g_k2_variant == 3) return launch_k2f_t<8, 128>(box4, row_off, n_rows, min_boxes, thr, out_high, out_max, st);
    return launch_k2f_t<8, 128>(box4, row_off, n_rows, min_boxes, thr, out_high, out_max, st);
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
