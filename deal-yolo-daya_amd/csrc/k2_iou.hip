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
#include <vector>

#include "k2_filter.h"

namespace dyd {

template <bool WANT_MAX, int WROWS, int WCAP>
__global__ __launch_bounds__(K2_BLOCK) void k2_iou_kernel(const double *__restrict__ box4,
                                                          const int32_t *__restrict__ row_off,
                                                          int64_t n_rows, int32_t min_boxes, double thr,
                                                          uint8_t *__restrict__ out_high,
                                                          double *__restrict__ out_max, unsigned long long *bigq) {
    __shared__ WaveLdsT<WROWS, WCAP> s_all[K2_WAVES];
    const int wave = threadIdx.x >> 6;
    const int64_t r0 = ((int64_t)blockIdx.x * K2_WAVES + wave) * WROWS;
    if (r0 >= n_rows) return;  // whole wave leaves; no workgroup barrier exists in this kernel
    const int nr = (n_rows - r0 < WROWS) ? (int)(n_rows - r0) : WROWS;
    k2_wave_rows<WANT_MAX, WROWS, WCAP>(box4, row_off, r0, nr, min_boxes, thr, out_high, out_max, s_all[wave], bigq);
}

// the same kernel with the f32 reject filter in front of the exact test (k2_filter.h)
template <bool WANT_MAX, int WROWS, int WCAP>
__global__ __launch_bounds__(K2_BLOCK) void k2f_iou_kernel(const double *__restrict__ box4,
                                                           const int32_t *__restrict__ row_off,
                                                           int64_t n_rows, int32_t min_boxes, double thr,
                                                           uint8_t *__restrict__ out_high,
                                                           double *__restrict__ out_max, unsigned long long *bigq) {
    __shared__ WaveLdsF<WROWS, WCAP> s_all[K2_WAVES];
    const int wave = threadIdx.x >> 6;
    const int64_t r0 = ((int64_t)blockIdx.x * K2_WAVES + wave) * WROWS;
    if (r0 >= n_rows) return;
    const int nr = (n_rows - r0 < WROWS) ? (int)(n_rows - r0) : WROWS;
    k2f_wave_rows<WANT_MAX, WROWS, WCAP>(box4, row_off, r0, nr, min_boxes, thr, out_high, out_max, s_all[wave], bigq);
}

template <int WROWS, int WCAP>
static int launch_k2f_t(const double *box4, const int32_t *row_off, int64_t n_rows, int32_t min_boxes, double thr,
                        uint8_t *out_high, double *out_max, unsigned long long *bigq, hipStream_t st) {
    const int64_t blocks = ceil_div(n_rows, (int64_t)K2_WAVES * WROWS);
    if (blocks > 0x7fffffffLL) {
        set_error("n_rows=%lld exceeds one launch", (long long)n_rows);
        return DYD_ERR_RANGE;
    }
    if (out_max)
        hipLaunchKernelGGL((k2f_iou_kernel<true, WROWS, WCAP>), dim3((unsigned)blocks), dim3(K2_BLOCK), 0, st, box4,
                           row_off, n_rows, min_boxes, thr, out_high, out_max, bigq);
    else
        hipLaunchKernelGGL((k2f_iou_kernel<false, WROWS, WCAP>), dim3((unsigned)blocks), dim3(K2_BLOCK), 0, st, box4,
                           row_off, n_rows, min_boxes, thr, out_high, out_max, bigq);
    DYD_HIP(hipGetLastError());
    return DYD_OK;
}

template <int WROWS, int WCAP>
static int launch_k2_t(const double *box4, const int32_t *row_off, int64_t n_rows, int32_t min_boxes, double thr,
                       uint8_t *out_high, double *out_max, unsigned long long *bigq, hipStream_t st) {
    const int64_t blocks = ceil_div(n_rows, (int64_t)K2_WAVES * WROWS);
    if (blocks > 0x7fffffffLL) {
        set_error("n_rows=%lld exceeds one launch", (long long)n_rows);
        return DYD_ERR_RANGE;
    }
    if (out_max)
        hipLaunchKernelGGL((k2_iou_kernel<true, WROWS, WCAP>), dim3((unsigned)blocks), dim3(K2_BLOCK), 0, st, box4,
                           row_off, n_rows, min_boxes, thr, out_high, out_max, bigq);
    else
        hipLaunchKernelGGL((k2_iou_kernel<false, WROWS, WCAP>), dim3((unsigned)blocks), dim3(K2_BLOCK), 0, st, box4,
                           row_off, n_rows, min_boxes, thr, out_high, out_max, bigq);
    DYD_HIP(hipGetLastError());
    return DYD_OK;
}

// tile variant: 0 = 16 rows / 256 boxes per wave (16 waves per CU), 1 = 8 rows / 128 boxes (32 waves per CU),
// 2 / 3 = the same two tilings with the f32 reject filter (k2_filter.h), 4 = the fused wave kernel's pair stage alone,
// 5 = 8 rows / 256 boxes with the filter: rows of up to 256 boxes fit the tile and are swept in x1 order (k2_sweep.h)
// -1 (default): by the table's shape — the wave kernel's pair stage for sparse tables (variant 4), 8-row tiles + f32 filter (3)
// up to 128 boxes per image on average, 256-box tiles (5) beyond (tools/dense_sweep.py: at 128 boxes per row 0.46 vs 0.51 ms, at 256 4.6 vs 0.54)
static int g_k2_variant = -1;
void set_k2_variant(int v) { g_k2_variant = v; }
#ifdef K2S_DEBUG
int set_k2s_debug(void *p) {   // experiment builds only: device buffer of 16 u64 counters (k2_sweep.h)
    DYD_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_k2s_dbg), &p, sizeof(p)));
    return DYD_OK;
}
#endif

int launch_k2_wave64(const double *box4, const int32_t *row_off, int64_t n_rows, int32_t min_boxes, double thr, uint8_t *out_high,
                     unsigned long long *bigq, hipStream_t st);   // k12_fused.hip
int launch_k2_big_rows(const double *box4, const int32_t *row_off, unsigned long long *bigq, int32_t min_boxes, double thr,
                       uint8_t *out_high, double *out_max, hipStream_t st);

// the pair stage alone: main kernel (rows of thousands of boxes go to `bigq`) — the caller launches k2_big_rows behind it
static int launch_k2_main(const double *box4, const int32_t *row_off, int64_t n_rows, int32_t min_boxes, double thr,
                          uint8_t *out_high, double *out_max, hipStream_t st, unsigned long long *bigq, int64_t n_boxes) {
    // sparse tables (at most 32 boxes per image on average, no diagnostic maximum): the wave kernel's pair stage
    if ((g_k2_variant < 0 || g_k2_variant == 4) && !out_max && (g_k2_variant == 4 || (n_boxes >= 0 && n_boxes <= 32 * n_rows)))
        return launch_k2_wave64(box4, row_off, n_rows, min_boxes, thr, out_high, bigq, st);
    if (g_k2_variant == 0) return launch_k2_t<K2_WROWS, K2_WCAP>(box4, row_off, n_rows, min_boxes, thr, out_high, out_max, bigq, st);
    if (g_k2_variant == 1) return launch_k2_t<8, 128>(box4, row_off, n_rows, min_boxes, thr, out_high, out_max, bigq, st);
    if (g_k2_variant == 2) return launch_k2f_t<16, 256>(box4, row_off, n_rows, min_boxes, thr, out_high, out_max, bigq, st);
    if (g_k2_variant == 5 || (g_k2_variant < 0 && n_boxes > 128 * n_rows))
        return launch_k2f_t<8, 256>(box4, row_off, n_rows, min_boxes, thr, out_high, out_max, bigq, st);
    return launch_k2f_t<8, 128>(box4, row_off, n_rows, min_boxes, thr, out_high, out_max, bigq, st);
}

// The context's queue for rows of thousands of boxes (zeroed once; k2_big_rows_kernel leaves it empty again).  A launch on
// another stream than the previous user's waits for that user's big-row kernel first: the queue is one per context.
int acquire_bigq(unsigned long long **q, hipStream_t st) {
    Context &c = ctx();
    if (!c.bigq) {
        DYD_HIP(hipMalloc(&c.bigq, 2 * K2_BIGQ_BYTES));
        DYD_HIP(hipMemset(c.bigq, 0, 2 * K2_BIGQ_BYTES));
        DYD_HIP(hipEventCreateWithFlags(&c.bigq_ev, hipEventDisableTiming));
    }
    if (c.bigq_busy && c.bigq_stream != st) DYD_HIP(hipStreamWaitEvent(st, c.bigq_ev, 0));
    if (c.bigq_dirty) {   // a launch between acquire and drain failed last time: start from two empty queues
        DYD_HIP(hipMemsetAsync(c.bigq, 0, 2 * K2_BIGQ_BYTES, st));
        c.bigq_dirty = false;
    }
    c.bigq_turn ^= 1;
    c.bigq_dirty = true;   // until release_bigq: the drain kernel is what leaves the queues clean
    *q = static_cast<unsigned long long *>(c.bigq) + (c.bigq_turn ? K2_BIGQ_BYTES / 8 : 0);
    return DYD_OK;
}
void release_bigq(hipStream_t st) {
    Context &c = ctx();
    c.bigq_stream = st;
    c.bigq_busy = true;
    c.bigq_dirty = false;
    (void)hipEventRecord(c.bigq_ev, st);
}

// K2 over device arrays: main kernel, then the queued big rows spread over the grid
int launch_k2(const double *box4, const int32_t *row_off, int64_t n_rows, int32_t min_boxes, double thr,
              uint8_t *out_high, double *out_max, hipStream_t st, int64_t n_boxes) {
    if (n_rows == 0) return DYD_OK;
    unsigned long long *q = nullptr;
    int rc = acquire_bigq(&q, st);
    if (rc) return rc;
    rc = launch_k2_main(box4, row_off, n_rows, min_boxes, thr, out_high, out_max, st, q, n_boxes);
    if (!rc) rc = launch_k2_big_rows(box4, row_off, q, min_boxes, thr, out_high, out_max, st);
    if (!rc) release_bigq(st);
    return rc;
}

// ---- rows of thousands of boxes ---------------------------------------------------------------------------------
// The rows the main kernel queued (k2_wave.h): a row's pairs are cut into items of 64 boxes x K2_BIG_CHUNK partners, the items of
// all queued rows are numbered in one sequence, and wave w of the grid takes the items w, w + W, ...  Within an item lane l holds
// box i0 + l and meets the partners j > i of its chunk, 64 at a time through LDS, always as (lower index, higher index) — the
// reference's argument order, so rows with NaN corners need no special path.  An empty queue costs one load per wave.
constexpr int32_t K2_BIG_CHUNK = 4096;

struct alignas(16) BigRowLds {
    union {
        struct {
            double x1[kWave], y1[kWave], x2[kWave], y2[kWave];
        };
        uint32_t sweep[4 * K2_MID_ROW];   // a queued row of up to 1024 boxes: sorted keys | limits | y intervals (k2_sweep.h), 16 KB
    };
    uint32_t qa[2 * kWave], qb[2 * kWave];
};

template <bool WANT_MAX>
__global__ __launch_bounds__(K2_BLOCK) void k2_big_rows_kernel(const double *__restrict__ box4,
                                                               const int32_t *__restrict__ row_off,
                                                               const unsigned long long *__restrict__ bigq,
                                                               unsigned long long *__restrict__ bigq_other,
                                                               int32_t min_boxes, double thr,
                                                               uint8_t *__restrict__ out_high,
                                                               unsigned long long *__restrict__ out_max_bits) {
    __shared__ BigRowLds s_all[K2_WAVES];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    BigRowLds &S = s_all[wave];
    const int64_t n_waves = (int64_t)gridDim.x * K2_WAVES, me_wave = (int64_t)blockIdx.x * K2_WAVES + wave;
    const bool zero_hits = (0.0 >= thr);
    const double thr_lo = (thr > 0.0) ? thr * 0.999 : 0.0;
    int64_t item = 0;   // running number of the items, the same in every wave
    const unsigned long long pushed = bigq[0], pushed_mid = bigq[1];
    const int32_t n_big = pushed < (unsigned long long)K2_BIG_LIST ? (int32_t)pushed : K2_BIG_LIST;
    const int32_t n_mid = pushed_mid < (unsigned long long)K2_MID_LIST ? (int32_t)pushed_mid : K2_MID_LIST;
    // two queues take turns: this launch empties the OTHER one (its last user's drain has finished — same stream, or waited for),
    // so the next launch finds an empty queue without a memset in front of it
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        bigq_other[0] = 0ull;
        bigq_other[1] = 0ull;
    }
    // ---- the mid list: rows of 65..256 boxes deferred by the sparse wave kernel and rows of 257..1024 boxes deferred by every
    //      main kernel — one row per wave, sorted by x1 and swept (k2_sweep.h; up to 16 keys per lane) ----
    {
        unsigned long long *mx_out = WANT_MAX ? out_max_bits : nullptr;
        for (int64_t k = me_wave; k < n_mid; k += n_waves) {
            const unsigned long long *e = bigq + K2_BIGQ_MID0 + 2 * k;
            const int64_t r = (int64_t)e[0];
            const int32_t n = (int32_t)e[1];   // 2 .. 1024 (or fewer with the maximum wanted), thr > 0: the pusher checked
            const int64_t base = row_off[r];
            const K2sView V = {S.sweep, S.sweep + K2_MID_ROW, reinterpret_cast<float2 *>(S.sweep + 2 * K2_MID_ROW), S.qa, S.qb};
            const double tl = WANT_MAX ? 0.0 : thr_lo;
            uint32_t vk[16];
            bool bad = false;
            wave_sync();
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int32_t b = kWave * q + lane;
                vk[q] = 0xffffffffu;
                if (kWave * q < n) {   // wave-uniform
                    if (b < n) {
                        const Corners c = load_corners(box4, base + b);
                        uint32_t lim;
                        float2 yy;
                        bad |= !(n > 256 ? k2s_prepare<10>(c, (uint32_t)b, tl, vk[q], lim, yy) : k2s_prepare<8>(c, (uint32_t)b, tl, vk[q], lim, yy));
                        V.slim[b] = lim;
                        V.syy[b] = yy;
                    }
                }
            }
            bool hit = false;
            double mx = 0.0;
            if (n < 2) {
            } else if (__any(bad)) {   // a corner that is not finite: all pairs in the reference's (i < j) order
                for (int32_t i = lane; i < n - 1; i += kWave) {
                    const Corners me = load_corners(box4, base + i);
                    const double me_ar = area_of(me);
                    for (int32_t j = i + 1; j < n; ++j) {
                        const Corners o = load_corners(box4, base + j);
                        hit |= pair_hits<WANT_MAX, false>(me, o, me_ar, o, thr, thr_lo, false, mx);
                    }
                }
                hit = __any(hit);
            } else {   // a budget of n trips: a row the x1 order cannot spread is swept along the diagonal instead (k2_sweep.h)
                bool ab = false;
                if (n <= kWave) {
                    uint32_t v1[1] = {vk[0]};
                    hit = k2s_sweep_sorted<WANT_MAX, 1, 8, true>(box4, base, n, V, v1, thr, thr_lo, mx, n, &ab);
                    if (ab) hit = k2s_retry_diag<WANT_MAX, 1>(box4, base, n, V, tl, thr, thr_lo, mx);
                } else if (n <= 2 * kWave) {
                    uint32_t v2[2] = {vk[0], vk[1]};
                    hit = k2s_sweep_sorted<WANT_MAX, 2, 8, true>(box4, base, n, V, v2, thr, thr_lo, mx, n, &ab);
                    if (ab) hit = k2s_retry_diag<WANT_MAX, 2>(box4, base, n, V, tl, thr, thr_lo, mx);
                } else if (n <= 4 * kWave) {
                    uint32_t v4[4] = {vk[0], vk[1], vk[2], vk[3]};
                    hit = k2s_sweep_sorted<WANT_MAX, 4, 8, true>(box4, base, n, V, v4, thr, thr_lo, mx, n, &ab);
                    if (ab) hit = k2s_retry_diag<WANT_MAX, 4>(box4, base, n, V, tl, thr, thr_lo, mx);
                } else if (n <= 8 * kWave) {
                    uint32_t v8[8] = {vk[0], vk[1], vk[2], vk[3], vk[4], vk[5], vk[6], vk[7]};
                    hit = k2s_sweep_sorted<WANT_MAX, 8, 10, true>(box4, base, n, V, v8, thr, thr_lo, mx, n, &ab);
                    if (ab) hit = k2s_retry_diag<WANT_MAX, 8, 10>(box4, base, n, V, tl, thr, thr_lo, mx);
                } else {
                    hit = k2s_sweep_sorted<WANT_MAX, 16, 10, true>(box4, base, n, V, vk, thr, thr_lo, mx, n, &ab);
                    if (ab) hit = k2s_retry_diag<WANT_MAX, 16, 10>(box4, base, n, V, tl, thr, thr_lo, mx);
                }
            }
            if (hit && lane == 0 && n >= min_boxes) out_high[r] = 1;
            if (WANT_MAX) {
                unsigned long long bits = (unsigned long long)__double_as_longlong(mx);   // IoU >= 0: the bit patterns order like the values
#pragma unroll
                for (int dd = 32; dd >= 1; dd >>= 1) {
                    const unsigned long long o = __shfl_xor(bits, dd);
                    bits = o > bits ? o : bits;
                }
                if (lane == 0 && bits) atomicMax(&mx_out[r], bits);
            }
            wave_sync();
        }
    }
    if (n_big == 0) return;
    for (int32_t k = 0; k < n_big; ++k) {
        const int64_t r = (int64_t)bigq[2 + 2 * k];
        const int64_t base = row_off[r];
        const int32_t n = (int32_t)bigq[3 + 2 * k];   // boxes to pair: the row's, or its prefix before an empty polygon (fused path)
        if (!WANT_MAX && n < min_boxes) continue;
        bool hit = false;
        double mx = 0.0;
        for (int32_t i0 = 0; i0 < n - 1; i0 += kWave) {
            const int32_t first_j = i0 + 1;
            const int32_t n_chunks = (n - first_j + K2_BIG_CHUNK - 1) / K2_BIG_CHUNK;
            // the chunks of this i-tile that are this wave's: item + c == me_wave (mod n_waves)
            int64_t c = ((me_wave - item) % n_waves + n_waves) % n_waves;
            item += n_chunks;
            if (c >= n_chunks) continue;
            const int32_t i = i0 + lane;
            const bool have = i < n - 1;
            Corners me = {0.0, 0.0, 0.0, 0.0};
            double me_ar = 0.0;
            if (have) {
                me = load_corners(box4, base + i);
                me_ar = area_of(me);
            }
            for (; c < n_chunks; c += n_waves) {
                const int32_t jlo = first_j + (int32_t)c * K2_BIG_CHUNK;
                const int32_t jhi = (n - jlo > K2_BIG_CHUNK) ? jlo + K2_BIG_CHUNK : n;
                for (int32_t tj = jlo; tj < jhi; tj += kWave) {
                    const int32_t tn = (jhi - tj < kWave) ? jhi - tj : kWave;
                    wave_sync();
                    if (lane < tn) {
                        const Corners v = load_corners(box4, base + tj + lane);
                        S.x1[lane] = v.x1; S.y1[lane] = v.y1; S.x2[lane] = v.x2; S.y2[lane] = v.y2;
                    }
                    wave_sync();
                    if (have) {
                        for (int32_t q = (i + 1 > tj) ? i + 1 - tj : 0; q < tn; ++q) {
                            const Corners o = {S.x1[q], S.y1[q], S.x2[q], S.y2[q]};
                            hit |= pair_hits<WANT_MAX, false>(me, o, me_ar, o, thr, thr_lo, zero_hits, mx);   // i < j
                        }
                    }
                }
            }
        }
        if (__any(hit) && lane == 0 && n >= min_boxes) out_high[r] = 1;
        if (WANT_MAX) {
            unsigned long long bits = (unsigned long long)__double_as_longlong(mx);   // IoU >= 0: the bit patterns order like the values
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) {
                const unsigned long long o = __shfl_xor(bits, d);
                bits = o > bits ? o : bits;
            }
            if (lane == 0 && bits) atomicMax(&out_max_bits[r], bits);
        }
    }
}

int launch_k2_big_rows(const double *box4, const int32_t *row_off, unsigned long long *bigq, int32_t min_boxes, double thr,
                       uint8_t *out_high, double *out_max, hipStream_t st) {
    const unsigned blocks = (unsigned)ctx().num_cu * 2;   // 8 waves per CU, striding over the items (an empty queue is the usual case)
    unsigned long long *base = static_cast<unsigned long long *>(ctx().bigq);
    unsigned long long *other = (bigq == base) ? base + K2_BIGQ_BYTES / 8 : base;
    if (out_max)
        hipLaunchKernelGGL(k2_big_rows_kernel<true>, dim3(blocks), dim3(K2_BLOCK), 0, st, box4, row_off, bigq, other, min_boxes, thr,
                           out_high, reinterpret_cast<unsigned long long *>(out_max));
    else
        hipLaunchKernelGGL(k2_big_rows_kernel<false>, dim3(blocks), dim3(K2_BLOCK), 0, st, box4, row_off, bigq, other, min_boxes, thr,
                           out_high, (unsigned long long *)nullptr);
    DYD_HIP(hipGetLastError());
    return DYD_OK;
}

}  // namespace dyd

using namespace dyd;

extern "C" {

int dyd_iou_any_ge_dev(const double *box4, const int32_t *row_off, int64_t n_rows, int64_t n_boxes, int32_t min_boxes,
                       double thr, uint8_t *out_high, double *out_max_iou_or_null, void *stream) {
    DYD_API_ENTER();
    DYD_REQUIRE(n_rows >= 0, "n_rows < 0");
    if (n_rows == 0) return DYD_OK;
    DYD_REQUIRE(row_off && out_high, "null pointer");
    DYD_REQUIRE((reinterpret_cast<uintptr_t>(box4) & 15) == 0, "box4 must be 16-byte aligned");
    return launch_k2(box4, row_off, n_rows, min_boxes, thr, out_high, out_max_iou_or_null, pick_stream(stream), n_boxes);
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
                   out_max_iou_or_null ? d_max.as<double>() : nullptr, st, nb);
    if (rc) return rc;
    t.finish();
    DYD_HIP(hipMemcpyAsync(out_high, d_high.p, (size_t)n_rows, hipMemcpyDeviceToHost, st));
    if (out_max_iou_or_null)
        DYD_HIP(hipMemcpyAsync(out_max_iou_or_null, d_max.p, 8 * (size_t)n_rows, hipMemcpyDeviceToHost, st));
    DYD_HIP(hipStreamSynchronize(st));
    return DYD_OK;
}

}  // extern "C"
