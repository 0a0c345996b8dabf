// k12_fused.hip — K1+K2 fused: poly -> bbox -> IoU flag in ONE launch, plus the tuning hook.
//
// reference ui/pages/processing.py:580-598 runs process_csv_replace_ptlist and
// filter_by_box_count_and_iou back to back on the same rows; K1's out_box4 =
// (min_x, min_y, max_x, max_y) is exactly the two-point ptList the IoU step reads back
// (processor.py:260 -> :354-362), so K2 can consume it directly.
//
// Mapping: a 256-thread workgroup owns K2_WAVES * K2_WROWS = 64 consecutive image rows, i.e. one
// contiguous box range and one contiguous point range.
//   phase 1 (K1, HBM-bound): the box range is processed in tiles of 256 boxes exactly as
//     k1_bbox_lds does (points streamed HBM -> LDS, lane-per-box first-wins scan), results go to
//     out_box4 / out_arg4;
//   phase 2 (K2, VALU-bound): after one workgroup barrier each wave runs the wave-autonomous IoU
//     code on its 16 rows, reading the boxes the workgroup has just written — they are still in
//     the XCD's L2, so K2's 32 B/box never come from HBM again.
// The two phases use the same LDS bytes (point chunk, then the waves' box columns), so the
// occupancy is that of K2 alone (4 workgroups per CU), and while one workgroup's waves are busy
// with pair arithmetic the other workgroups of the CU keep the memory pipeline streaming:
// the VALU-bound part hides under the HBM-bound part.
// Algorithmic bytes per launch: 16*P + 4*(B+1) + 48*B + 4*(N+1) + N.
#include <cstring>
#include <type_traits>

#include "k12_wave.h"
#include "k2_filter.h"

namespace dyd {

// CHUNK = K1 point-chunk size, (WROWS, WCAP) = K2 per-wave tile.  The two phases alias one LDS
// buffer, so its size — and the number of workgroups a CU can hold — is the larger of the two.
template <int CHUNK, int WROWS, int WCAP, bool FILTER = false>
__global__ __launch_bounds__(K1_BLOCK, (FILTER && WCAP == 256 && WROWS == 8) ? 6 : 1) void k12_fused_kernel(const double2 *__restrict__ xy,
                                                             const int32_t *__restrict__ pt_off,
                                                             const int32_t *__restrict__ box_off, int64_t n_rows,
                                                             int32_t min_boxes, double thr, double *out_box4,
                                                             int32_t *__restrict__ out_arg4,
                                                             uint8_t *__restrict__ out_high, unsigned long long *bigq) {
    using Slice = typename std::conditional<FILTER, WaveLdsF<WROWS, WCAP>, WaveLdsT<WROWS, WCAP>>::type;
    constexpr size_t kLds = sizeof(double2) * CHUNK > sizeof(Slice) * K2_WAVES ? sizeof(double2) * CHUNK
                                                                              : sizeof(Slice) * K2_WAVES;
    constexpr int kRows = K2_WAVES * WROWS;
    __shared__ __attribute__((aligned(16))) unsigned char s_raw[kLds];
    const int64_t row0 = (int64_t)blockIdx.x * kRows;
    const int64_t row1 = (row0 + kRows < n_rows) ? row0 + kRows : n_rows;
    const int64_t bA = box_off[row0], bB = box_off[row1];  // workgroup-uniform
    // ---- phase 1: K1 over the workgroup's boxes ------------------------------------------------
    double2 *s_pts = reinterpret_cast<double2 *>(s_raw);
    for (int64_t b0 = bA; b0 < bB; b0 += K1_BLOCK)
        k1_process_tile<CHUNK>(xy, pt_off, b0, bB, out_box4, out_arg4, s_pts);
    // every wave's box stores are complete and visible to the workgroup; LDS may be reused
    __syncthreads();
    // ---- phase 2: K2, one wave per WROWS rows, no further workgroup barrier ------------------------
    const int wave = threadIdx.x >> 6;
    const int64_t r0 = row0 + (int64_t)wave * WROWS;
    if (r0 >= n_rows) return;
    const int nr = (n_rows - r0 < WROWS) ? (int)(n_rows - r0) : WROWS;
    Slice *S = reinterpret_cast<Slice *>(s_raw);
    if constexpr (FILTER)
        k2f_wave_rows<false, WROWS, WCAP>(out_box4, box_off, r0, nr, min_boxes, thr, out_high, nullptr, S[wave], bigq);
    else
        k2_wave_rows<false, WROWS, WCAP>(out_box4, box_off, r0, nr, min_boxes, thr, out_high, nullptr, S[wave], bigq);
}

// wave-autonomous variant: boxes go from K1 to K2 through LDS, no workgroup barrier (k12_wave.h)
template <int WPB>   // waves per workgroup: the waves are autonomous, so this only sets the dispatch granularity
__global__ __launch_bounds__(64 * WPB) void k12_wave_kernel(const double2 *__restrict__ xy,
                                                            const int32_t *__restrict__ pt_off,
                                                            const int32_t *__restrict__ box_off, int64_t n_rows,
                                                            int32_t min_boxes, double thr, double *out_box4,
                                                            int32_t *__restrict__ out_arg4,
                                                            uint8_t *__restrict__ out_high, unsigned long long *bigq) {
    __shared__ WaveFuse s_all[WPB];
    const int wave = threadIdx.x >> 6;
    // experiment (-DK12_XCD_REMAP; workgroups are dealt round-robin to the 8 XCDs): every XCD streams one contiguous eighth of the table
    // instead of every eighth workgroup-sized piece.  tools/xcd_ab.sh, 10 M rows: 1-2 % ahead on one box (5.44 / 5.54 / 5.44 ms against
    // 5.53 / 5.55 / 5.54), 2-4 % behind and erratic on the next (5.77 / 5.78 / 5.46 against 5.54 / 5.57 / 5.53): not the default
#ifdef K12_XCD_REMAP
    const int64_t per = gridDim.x / 8;
    const int64_t blk = (int64_t)(blockIdx.x % 8) * per + blockIdx.x / 8;
#else
    const int64_t blk = blockIdx.x;
#endif
    const int64_t r0 = (blk * WPB + wave) * KW_ROWS;
    if (r0 >= n_rows) return;
    const int nr = (n_rows - r0 < KW_ROWS) ? (int)(n_rows - r0) : KW_ROWS;
    k12_wave_rows(xy, pt_off, box_off, r0, nr, min_boxes, thr, out_box4, out_arg4, out_high, s_all[wave], bigq);
}

// the same kernel for tables above 32 boxes per image on average: rows of 40..256 boxes are sorted and swept (k12_wave.h, k2_sweep.h)
// WPE = waves per SIMD the register allocation aims at: 7 (72 VGPRs, 16 bytes of scratch) wins up to 128 boxes per image, 6 (80, none)
// beyond, where the four-keys-per-lane sort is what runs (tools/dense_sweep.py: 2.13 / 2.25 / 2.54 ms against 2.19 / 2.35 / 2.43 at
// 64 / 128 / 256 boxes per row)
constexpr int KWD_ROWS = 8;   // image rows per wave: dense rows are long, half of KW_ROWS keeps twice as many waves in flight on small tables
template <int WPE>
__global__ __launch_bounds__(256, WPE) void k12_wave_dense_kernel(const double2 *__restrict__ xy, const int32_t *__restrict__ pt_off,
                                                             const int32_t *__restrict__ box_off, int64_t n_rows, int32_t min_boxes,
                                                             double thr, double *out_box4, int32_t *__restrict__ out_arg4,
                                                             uint8_t *__restrict__ out_high, unsigned long long *bigq) {
    __shared__ WaveFuseDense s_all[4];
    const int wave = threadIdx.x >> 6;
#ifdef K12_XCD_REMAP   // as in k12_wave_kernel (experiment)
    const int64_t per = gridDim.x / 8;
    const int64_t blk = (int64_t)(blockIdx.x % 8) * per + blockIdx.x / 8;
#else
    const int64_t blk = blockIdx.x;
#endif
    const int64_t r0 = (blk * 4 + wave) * KWD_ROWS;
    if (r0 >= n_rows) return;
    const int nr = (n_rows - r0 < KWD_ROWS) ? (int)(n_rows - r0) : KWD_ROWS;
    k12_wave_rows<false, true>(xy, pt_off, box_off, r0, nr, min_boxes, thr, out_box4, out_arg4, out_high, s_all[wave], bigq);
}

// K2 alone with the wave kernel's pair stage: a wave owns 16 rows, walks them in tiles of whole rows with at most 64 boxes (one
// box per lane, loaded 32 bytes per lane), and runs the same all-pairs loop — none of the tile kernels' bookkeeping (row ranks,
// row search, permutation): 0.245 -> 0.196 ms on the bench table.  Rows beyond 64 boxes take the slow partner-tile route,
// so tables with many of them stay with the tile kernels (launch_k2 decides by the mean).
template <int WPB>
__global__ __launch_bounds__(64 * WPB) void k2_wave64_kernel(const double *box4, const int32_t *__restrict__ box_off, int64_t n_rows,
                                                             int32_t min_boxes, double thr, uint8_t *__restrict__ out_high,
                                                             unsigned long long *bigq) {
    __shared__ WaveFuse s_all[WPB];
    const int wave = threadIdx.x >> 6;
    const int64_t r0 = ((int64_t)blockIdx.x * WPB + wave) * KW_ROWS;
    if (r0 >= n_rows) return;
    const int nr = (n_rows - r0 < KW_ROWS) ? (int)(n_rows - r0) : KW_ROWS;
    k12_wave_rows<true>(nullptr, nullptr, box_off, r0, nr, min_boxes, thr, const_cast<double *>(box4), nullptr, out_high, s_all[wave], bigq);
}

// Chain semantics for the variants whose pair stage reads the boxes back from memory (workgroup tiles, the
// two-launch route of long polygons): a polygon without a valid point (arg index -1) ends its row's IoU box list
// (reference processor.py:254-255 -> :364-365, see k12_wave.h).  This pass runs after them, looks for such boxes —
// 16 bytes per box, which is noise for tables of dense rows or long polygons — and recomputes the flag of the few
// rows that hold one from the prefix before it, pairs in the reference's (i < j) order.
__global__ __launch_bounds__(256) void k12_null_fix_kernel(const int32_t *__restrict__ arg4, const double *__restrict__ box4,
                                                           const int32_t *__restrict__ box_off, int64_t n_rows, int64_t n_boxes,
                                                           int32_t min_boxes, double thr, uint8_t *__restrict__ out_high) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t stride = ((int64_t)gridDim.x * blockDim.x >> 6) * kWave;
    const bool zero_hits = (0.0 >= thr);
    const double thr_lo = (thr > 0.0) ? thr * 0.999 : 0.0;
    double unused_mx = 0.0;
    for (int64_t b0 = wave * kWave; b0 < n_boxes; b0 += stride) {
        const int64_t b = b0 + lane;
        unsigned long long em = __ballot(b < n_boxes && arg4[4 * b] < 0);
        while (em != 0ull) {  // wave-uniform
            const int64_t be = b0 + (__ffsll((long long)em) - 1);
            em &= em - 1ull;
            int64_t lo = 0, hi = n_rows;  // the row r with box_off[r] <= be < box_off[r + 1]
            while (hi - lo > 1) {
                const int64_t mid = (lo + hi) >> 1;
                if ((int64_t)box_off[mid] <= be) lo = mid; else hi = mid;
            }
            const int64_t s = box_off[lo];
            bool earlier = false;  // an earlier empty polygon of the same row decides instead
            for (int64_t j = s + lane; j < be; j += kWave) earlier |= arg4[4 * j] < 0;
            if (__ballot(earlier) != 0ull) continue;
            const int64_t n = be - s;
            bool hit = false;
            if (n >= 2 && n >= min_boxes) {
                for (int64_t i = lane; i < n - 1; i += kWave) {
                    const double2 *gi = reinterpret_cast<const double2 *>(box4 + 4 * (s + i));
                    const Corners me = normalise(gi[0], gi[1]);
                    const double me_ar = area_of(me);
                    for (int64_t j = i + 1; j < n; ++j) {
                        const double2 *gj = reinterpret_cast<const double2 *>(box4 + 4 * (s + j));
                        const Corners o = normalise(gj[0], gj[1]);
                        hit |= pair_hits<false, false>(me, o, me_ar, o, thr, thr_lo, zero_hits, unused_mx);
                    }
                }
            }
            const bool any = __ballot(hit) != 0ull;
            if (lane == 0) out_high[lo] = (uint8_t)any;
        }
    }
}

static int launch_null_fix(const int32_t *arg4, const double *box4, const int32_t *box_off, int64_t n_rows, int64_t n_boxes,
                           int32_t min_boxes, double thr, uint8_t *out_high, hipStream_t st) {
    if (n_boxes == 0 || n_rows == 0) return DYD_OK;
    const int64_t want = ceil_div(n_boxes, 256);
    const int64_t cap = (int64_t)ctx().num_cu * 8;
    hipLaunchKernelGGL(k12_null_fix_kernel, dim3((unsigned)(want < cap ? want : cap)), dim3(256), 0, st, arg4, box4, box_off, n_rows,
                       n_boxes, min_boxes, thr, out_high);
    DYD_HIP(hipGetLastError());
    return DYD_OK;
}

int launch_k2_wave64(const double *box4, const int32_t *row_off, int64_t n_rows, int32_t min_boxes, double thr, uint8_t *out_high,
                     unsigned long long *bigq, hipStream_t st) {
    const int64_t blocks = ceil_div(n_rows, (int64_t)4 * KW_ROWS);
    if (blocks > 0x7fffffffLL) {
        set_error("n_rows=%lld exceeds one launch", (long long)n_rows);
        return DYD_ERR_RANGE;
    }
    hipLaunchKernelGGL(k2_wave64_kernel<4>, dim3((unsigned)blocks), dim3(256), 0, st, box4, row_off, n_rows, min_boxes, thr, out_high, bigq);
    DYD_HIP(hipGetLastError());
    return DYD_OK;
}

template <int CHUNK, int WROWS, int WCAP, bool FILTER = false>
static int launch_fused(const double *xy, const int32_t *pt_off, const int32_t *box_off, int64_t n_rows,
                        int32_t min_boxes, double thr, double *out_box4, int32_t *out_arg4, uint8_t *out_high,
                        unsigned long long *bigq, hipStream_t st) {
    const int64_t blocks = ceil_div(n_rows, (int64_t)K2_WAVES * WROWS);
    if (blocks > 0x7fffffffLL) {
        set_error("n_rows=%lld exceeds one launch", (long long)n_rows);
        return DYD_ERR_RANGE;
    }
    hipLaunchKernelGGL((k12_fused_kernel<CHUNK, WROWS, WCAP, FILTER>), dim3((unsigned)blocks), dim3(K1_BLOCK), 0, st,
                       reinterpret_cast<const double2 *>(xy), pt_off, box_off, n_rows, min_boxes, thr, out_box4,
                       out_arg4, out_high, bigq);
    DYD_HIP(hipGetLastError());
    return DYD_OK;
}

int launch_k1(const double *xy, const int32_t *pt_off, int64_t n_boxes, int64_t n_points, double *out_box4, int32_t *out_arg4,
              hipStream_t st);
bool k1_wants_groups(int64_t n_boxes, int64_t n_points);
int launch_k2(const double *box4, const int32_t *row_off, int64_t n_rows, int32_t min_boxes, double thr,
              uint8_t *out_high, double *out_max, hipStream_t st, int64_t n_boxes = -1);
int acquire_bigq(unsigned long long **q, hipStream_t st);
void release_bigq(hipStream_t st);
int launch_k2_big_rows(const double *box4, const int32_t *row_off, unsigned long long *bigq, int32_t min_boxes, double thr,
                       uint8_t *out_high, double *out_max, hipStream_t st);
void set_k1_variant(int v);
void set_k2_variant(int v);
void set_k7_variant(int v);
void set_k6_variant(int v);
void set_k4_capacity_shift(int v);
void set_k8_band(int v);
void set_k7_trace(void *p);
#ifdef K2S_DEBUG
int set_k2s_debug(void *p);
#endif

// -1 = by the table's shape (below); 1 = K1 launch then K2 launch (polygons of 48 points and more), 4 = wave-autonomous fused
// kernel (LDS hand-off; up to 32 boxes per image on average), 10 = its DENSE instantiation (rows of 40..256 boxes sorted and swept,
// k2_sweep.h), 6 / 9 = workgroup-level fusion <1024,8,128> / <1024,8,256> with the f32 reject filter (polygons of 20..48 points).
// The other tilings of rounds 1-2 (0, 2, 3, 5 and the 1- and 2-wave workgroups 7, 8) were A/B residue and are gone.
static int g_fused_variant = -1;

}  // namespace dyd

using namespace dyd;

extern "C" {

int dyd_bbox_iou_fused_dev(const double *xy, const int32_t *pt_off, const int32_t *box_off, int64_t n_rows,
                           int64_t n_boxes, int64_t n_points, int32_t min_boxes, double thr, double *out_box4,
                           int32_t *out_arg4, uint8_t *out_high, void *stream) {
    DYD_API_ENTER();
    DYD_REQUIRE(n_rows >= 0 && n_boxes >= 0, "negative size");
    if (n_rows == 0) return DYD_OK;
    DYD_REQUIRE(pt_off && box_off && out_box4 && out_arg4 && out_high, "null pointer");
    DYD_REQUIRE(((reinterpret_cast<uintptr_t>(xy) | reinterpret_cast<uintptr_t>(out_box4) |
                  reinterpret_cast<uintptr_t>(out_arg4)) & 15) == 0,
                "xy / out_box4 / out_arg4 must be 16-byte aligned");
    hipStream_t st = pick_stream(stream);
    // long polygons: a lane per box would walk alone (k1_bbox.hip): the group kernel, then K2 (the boxes are 32 of the table's
    // 400+ bytes per box, so reading them back costs little)
    if (g_fused_variant == 1 || (g_fused_variant < 0 && k1_wants_groups(n_boxes, n_points))) {
        int rc = launch_k1(xy, pt_off, n_boxes, n_points, out_box4, out_arg4, st);
        if (rc) return rc;
        rc = launch_k2(out_box4, box_off, n_rows, min_boxes, thr, out_high, nullptr, st, n_boxes);
        if (rc) return rc;
        return launch_null_fix(out_arg4, out_box4, box_off, n_rows, n_boxes, min_boxes, thr, out_high, st);
    }
    // rows of thousands of boxes are queued by the main kernel and paired by k2_big_rows_kernel behind it (k2_wave.h)
    unsigned long long *bigq = nullptr;
    {
        const int rcq = acquire_bigq(&bigq, st);
        if (rcq) return rcq;
    }
    int v = g_fused_variant;
    // sparse rows (<= 32 boxes per image on average): the wave-autonomous kernel (32 waves per CU, boxes handed to K2 through
    // LDS); denser tables: its DENSE instantiation (rows of 40..256 boxes sorted and swept; tools/fused_sweep.py and
    // tools/dense_sweep.py: ahead of the workgroup kernels from 24 boxes per row on, 2.15-2.4 ms against 2.4-2.9 per 64 M boxes);
    // polygons of 20..48 points: workgroup-level fusion (its tiles keep more lanes walking than a wave's 64-box tile does)
    if (v < 0) v = (n_points > 20 * n_boxes) ? (n_boxes > 128 * n_rows ? 9 : 6) : (n_boxes <= 32 * n_rows ? 4 : 10);   // tools/fused_sweep.py: fixed 32 boxes per row 0.146 vs 0.171 ms, 48: equal, 96: 0.60 vs 0.32
    if (v == 10) {   // the wave kernel with sort and sweep for rows of 40..256 boxes
        int64_t blocks = ceil_div(n_rows, (int64_t)4 * KWD_ROWS);
#ifdef K12_XCD_REMAP
        blocks = ceil_div(blocks, 8) * 8;
#endif
        if (blocks > 0x7fffffffLL) {
            set_error("n_rows=%lld exceeds one launch", (long long)n_rows);
            return DYD_ERR_RANGE;
        }
        if (n_boxes > 128 * n_rows)
            hipLaunchKernelGGL(k12_wave_dense_kernel<6>, dim3((unsigned)blocks), dim3(256), 0, st, reinterpret_cast<const double2 *>(xy),
                               pt_off, box_off, n_rows, min_boxes, thr, out_box4, out_arg4, out_high, bigq);
        else
            hipLaunchKernelGGL(k12_wave_dense_kernel<7>, dim3((unsigned)blocks), dim3(256), 0, st, reinterpret_cast<const double2 *>(xy),
                               pt_off, box_off, n_rows, min_boxes, thr, out_box4, out_arg4, out_high, bigq);
        DYD_HIP(hipGetLastError());
        const int rcb = launch_k2_big_rows(out_box4, box_off, bigq, min_boxes, thr, out_high, nullptr, st);
        if (!rcb) release_bigq(st);
        return rcb;
    }
    if (v == 4) {
        int64_t blocks = ceil_div(n_rows, (int64_t)4 * KW_ROWS);   // 4 waves per workgroup (1, 2 and 4 time the same: 0.598 ms back to back at 1 M rows)
#ifdef K12_XCD_REMAP
        blocks = ceil_div(blocks, 8) * 8;   // the grid is a multiple of 8: workgroups beyond the table leave at once
#endif
        if (blocks > 0x7fffffffLL) {
            set_error("n_rows=%lld exceeds one launch", (long long)n_rows);
            return DYD_ERR_RANGE;
        }
        hipLaunchKernelGGL(k12_wave_kernel<4>, dim3((unsigned)blocks), dim3(256), 0, st, reinterpret_cast<const double2 *>(xy), pt_off, box_off,
                           n_rows, min_boxes, thr, out_box4, out_arg4, out_high, bigq);
        DYD_HIP(hipGetLastError());
        const int rcb = launch_k2_big_rows(out_box4, box_off, bigq, min_boxes, thr, out_high, nullptr, st);
        if (!rcb) release_bigq(st);
        return rcb;
    }
    int rc;
    if (v == 9)
        rc = launch_fused<1024, 8, 256, true>(xy, pt_off, box_off, n_rows, min_boxes, thr, out_box4, out_arg4, out_high, bigq, st);
    else if (v == 6)
        rc = launch_fused<1024, 8, 128, true>(xy, pt_off, box_off, n_rows, min_boxes, thr, out_box4, out_arg4, out_high, bigq, st);
    else {
        release_bigq(st);
        set_error("invalid argument: fused_variant %d does not exist (1, 4, 6, 9, 10 or -1)", v);
        return DYD_ERR_INVALID;
    }
    if (!rc) rc = launch_k2_big_rows(out_box4, box_off, bigq, min_boxes, thr, out_high, nullptr, st);
    if (rc) return rc;
    release_bigq(st);
    return launch_null_fix(out_arg4, out_box4, box_off, n_rows, n_boxes, min_boxes, thr, out_high, st);
}

// Host-pointer twin: stages the three input arrays, runs the fused launch and copies back the arg indices (what the JSON emitter
// needs), the flags and — only when asked for — the boxes.  It works on a stream of its own and holds the library's lock only
// while the kernels are queued, so the worker threads of the native replace -> IoU pipeline (host_json.cpp: one call per
// thread's share of the cells) stage and copy side by side.
// ---- host-pointer passes through a staging slot (dyd_common.h) ------------------------------------------------------------
struct dyd_stage {
    StageSlot *slot;
};

int dyd_stage_acquire(size_t pinned_bytes, dyd_stage **out, void **pinned, size_t *pinned_cap) {
    if (!out) return DYD_ERR_INVALID;
    StageSlot *slot = nullptr;
    const int rc = stage_acquire(pinned_bytes, &slot);
    if (rc) return rc;
    dyd_stage *h = new (std::nothrow) dyd_stage{slot};
    if (!h) { stage_release(slot); return DYD_ERR_OOM; }
    *out = h;
    if (pinned) *pinned = slot->pin;
    if (pinned_cap) *pinned_cap = slot->pin_cap;
    return DYD_OK;
}

void dyd_stage_release(dyd_stage *h) {
    if (!h) return;
    stage_release(h->slot);
    delete h;
}

int dyd_bbox_iou_fused_staged(dyd_stage *h, const double *xy, const int32_t *pt_off, const int32_t *box_off, int64_t n_rows,
                              int32_t min_boxes, double thr, double *out_box4_or_null, int32_t *out_arg4, uint8_t *out_high) {
    DYD_REQUIRE(h && h->slot, "no staging slot");
    if (n_rows < 0) { set_error("invalid argument: n_rows < 0"); return DYD_ERR_INVALID; }
    if (n_rows == 0) return DYD_OK;
    DYD_REQUIRE(pt_off && box_off && out_high, "null pointer");
    DYD_REQUIRE(box_off[0] == 0 && pt_off[0] == 0, "offsets must start at 0");
    for (int64_t i = 0; i < n_rows; ++i) DYD_REQUIRE(box_off[i + 1] >= box_off[i], "box_off not monotone");
    const int64_t n_boxes = box_off[n_rows];
    for (int64_t i = 0; i < n_boxes; ++i) DYD_REQUIRE(pt_off[i + 1] >= pt_off[i], "pt_off not monotone");
    const int64_t n_pts = pt_off[n_boxes];
    DYD_REQUIRE(n_pts == 0 || xy, "xy is null");
    DYD_REQUIRE(n_boxes == 0 || out_arg4, "out_arg4 is null");
    StageSlot *slot = h->slot;
    // one device arena: xy | pt_off | box_off | box4 | arg4 | high, every piece on a 256-byte boundary
    auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const size_t o_xy = 0, o_po = o_xy + up(16 * (size_t)n_pts), o_bo = o_po + up(4 * (size_t)(n_boxes + 1)),
                 o_box = o_bo + up(4 * (size_t)(n_rows + 1)), o_arg = o_box + up(32 * (size_t)n_boxes),
                 o_high = o_arg + up(16 * (size_t)n_boxes), total = o_high + up((size_t)n_rows);
    const int rcd = stage_device(slot, total);
    if (rcd) return rcd;
    char *d = static_cast<char *>(slot->dev);
    hipStream_t st = slot->s;
    if (n_pts) DYD_HIP(hipMemcpyAsync(d + o_xy, xy, 16 * (size_t)n_pts, hipMemcpyHostToDevice, st));
    DYD_HIP(hipMemcpyAsync(d + o_po, pt_off, 4 * (size_t)(n_boxes + 1), hipMemcpyHostToDevice, st));
    DYD_HIP(hipMemcpyAsync(d + o_bo, box_off, 4 * (size_t)(n_rows + 1), hipMemcpyHostToDevice, st));
    DYD_HIP(hipEventRecord(slot->e0, st));
    int rc = dyd_bbox_iou_fused_dev(reinterpret_cast<const double *>(d + o_xy), reinterpret_cast<const int32_t *>(d + o_po),
                                    reinterpret_cast<const int32_t *>(d + o_bo), n_rows, n_boxes, n_pts, min_boxes, thr,
                                    reinterpret_cast<double *>(d + o_box), reinterpret_cast<int32_t *>(d + o_arg),
                                    reinterpret_cast<uint8_t *>(d + o_high), st);   // takes the lock while it queues
    if (rc) { (void)hipStreamSynchronize(st); return rc; }
    DYD_HIP(hipEventRecord(slot->e1, st));
    if (n_boxes) DYD_HIP(hipMemcpyAsync(out_arg4, d + o_arg, 16 * (size_t)n_boxes, hipMemcpyDeviceToHost, st));
    if (n_boxes && out_box4_or_null) DYD_HIP(hipMemcpyAsync(out_box4_or_null, d + o_box, 32 * (size_t)n_boxes, hipMemcpyDeviceToHost, st));
    DYD_HIP(hipMemcpyAsync(out_high, d + o_high, (size_t)n_rows, hipMemcpyDeviceToHost, st));
    DYD_HIP(hipStreamSynchronize(st));
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, slot->e0, slot->e1) == hipSuccess) set_last_kernel_ms(ms);
    return DYD_OK;
}

int dyd_bbox_iou_fused(const double *xy, const int32_t *pt_off, const int32_t *box_off, int64_t n_rows, int32_t min_boxes,
                       double thr, double *out_box4_or_null, int32_t *out_arg4, uint8_t *out_high) {
    if (n_rows < 0) { set_error("invalid argument: n_rows < 0"); return DYD_ERR_INVALID; }
    dyd_stage *h = nullptr;
    int rc = dyd_stage_acquire(0, &h, nullptr, nullptr);
    if (rc) return rc;
    rc = dyd_bbox_iou_fused_staged(h, xy, pt_off, box_off, n_rows, min_boxes, thr, out_box4_or_null, out_arg4, out_high);
    dyd_stage_release(h);
    return rc;
}

}  // extern "C"

extern "C" {

// Tuning / A-B hook (not part of the reference-facing ABI): selects kernel variants.
int dyd_set_option(const char *key, int64_t value) {
    std::lock_guard<std::recursive_mutex> lock(api_mutex());
    if (!key) return DYD_ERR_INVALID;
    if (!strcmp(key, "k1_variant")) {
        set_k1_variant((int)value);
        return DYD_OK;
    }
    if (!strcmp(key, "k2_variant")) {
        set_k2_variant((int)value);
        return DYD_OK;
    }
    if (!strcmp(key, "k6_variant")) {
        set_k6_variant((int)value);
        return DYD_OK;
    }
    if (!strcmp(key, "k4_capacity_shift")) {   // test hook: undersized hash table (the failure path must surface)
        set_k4_capacity_shift((int)value);
        return DYD_OK;
    }
    if (!strcmp(key, "k8_band")) {   // 0 = K8 resolves the rejections by full-length rounds only (A/B, tests)
        set_k8_band((int)value);
        return DYD_OK;
    }
    if (!strcmp(key, "k7_variant")) {
        set_k7_variant((int)value);
        return DYD_OK;
    }
    if (!strcmp(key, "k7_trace_ptr")) {   // device buffer of 8 x n_tiles u64 (0 = off)
        set_k7_trace(reinterpret_cast<void *>(static_cast<intptr_t>(value)));
        return DYD_OK;
    }
#ifdef K2S_DEBUG
    if (!strcmp(key, "k2s_debug_ptr")) {
        return set_k2s_debug(reinterpret_cast<void *>(static_cast<intptr_t>(value)));
    }
#endif
    if (!strcmp(key, "fused_variant")) {
        g_fused_variant = (int)value;
        return DYD_OK;
    }
    set_error("unknown option %s", key);
    return DYD_ERR_INVALID;
}

}  // extern "C"
