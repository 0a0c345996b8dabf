// k1_bbox.hip — K1: polygon ptList -> bounding box (first-wins min/max + arg indices).
//
// Replaces get_bbox_points, reference core/processor.py:252-260 (called per object at :273).
//
// Layout in HBM: xy = P interleaved (x,y) f64 points (16 B each, 16-B aligned), pt_off =
// B+1 int32 point offsets; out_box4 = B x (min_x,min_y,max_x,max_y) f64, out_arg4 = B x 4
// int32 indices inside the box.  Algorithmic bytes per launch: 16*P + 4*(B+1) + 48*B.
// Bound: HBM bandwidth (≈0.1 flop/B) — no MFMA, nothing here is a contraction.
//
// Mapping: one 256-thread workgroup owns a tile of 256 consecutive boxes, i.e. ONE contiguous
// point range.  The range is streamed HBM -> LDS with fully coalesced 16-B-per-lane loads in
// chunks of K1_CHUNK points; then lane t walks box t's points in LDS in their original order
// with strict </> compares — exactly CPython's sequential builtin min/max, so the first
// extremal element wins, -0.0/0.0 and int/float ties keep the lower index and a NaN survives
// only from position 0.  A box longer than a chunk simply spans several chunks.
#include "k1_tile.h"

namespace dyd {

// LDS-staged tile kernel (the product path): one tile of K1_BLOCK boxes per workgroup.
__global__ __launch_bounds__(K1_BLOCK) void k1_bbox_lds(const double2 *__restrict__ xy,
                                                        const int32_t *__restrict__ pt_off,
                                                        int64_t n_boxes,
                                                        double *__restrict__ out_box4,
                                                        int32_t *__restrict__ out_arg4) {
    __shared__ double2 s_pts[K1_CHUNK];
    k1_process_tile(xy, pt_off, (int64_t)blockIdx.x * K1_BLOCK, n_boxes, out_box4, out_arg4, s_pts);
}

// Direct variant (no LDS): lane t reads box t's points straight from global memory.  Kept for
// A/B measurement (dyd_set_option("k1_variant", 1)); same results by construction.
__global__ __launch_bounds__(K1_BLOCK) void k1_bbox_direct(const double2 *__restrict__ xy,
                                                           const int32_t *__restrict__ pt_off,
                                                           int64_t n_boxes,
                                                           double *__restrict__ out_box4,
                                                           int32_t *__restrict__ out_arg4) {
    const int64_t b = (int64_t)blockIdx.x * K1_BLOCK + threadIdx.x;
    if (b >= n_boxes) return;
    const int32_t s = pt_off[b], e = pt_off[b + 1];
    BoxAcc acc;
    acc.empty();
    if (s < e) {
        double2 v = xy[s];
        acc.first(v.x, v.y);
        for (int32_t p = s + 1; p < e; ++p) {
            v = xy[p];
            acc.next(v.x, v.y, p - s);
        }
    }
    acc.store(out_box4, out_arg4, b);
}

// Long polygons (segmentation-style outlines of tens to thousands of points): a lane per box walks its polygon alone while
// the other lanes of its tile wait — measured with 64 points per box 1.2 TB/s, with 256 0.09 (tools/points_sweep.py).
// Here GL = 16 lanes share a box: lane g takes the points g, g + 16, ... (a wave reads four boxes' 256-byte pieces per
// load), keeps its own first-wins extremes, and the sixteen partial results are folded towards lane 0 with the index as
// the tie-break — the first extremal point of the whole polygon wins, as in the sequential walk.  CPython's NaN rule
// (only a NaN at position 0 sticks) becomes: a lane seeds itself with its first non-NaN value, lane 0 with the polygon's
// first value whatever it is; lane 0 is always the receiving side of the fold, so its NaN survives.
constexpr int K1_GROUP = 16;

struct GroupAcc {   // one coordinate: running min and max over the lane's points, seeded together
    double mn, mx;
    int32_t imn, imx;
    bool has;
    __device__ __forceinline__ void take(double v, int32_t k) {
        if (!has) {
            if (v == v || k == 0) {
                mn = mx = v;
                imn = imx = k;
                has = true;
            }
        } else {
            if (v < mn) { mn = v; imn = k; }
            if (v > mx) { mx = v; imx = k; }
        }
    }
    // fold the partner d lanes up into this lane: strictly better wins, equal values keep the lower index
    __device__ __forceinline__ void fold(int d) {
        const double omn = __shfl_down(mn, d, K1_GROUP), omx = __shfl_down(mx, d, K1_GROUP);
        const int32_t oimn = __shfl_down(imn, d, K1_GROUP), oimx = __shfl_down(imx, d, K1_GROUP);
        const bool ohas = __shfl_down((int)has, d, K1_GROUP) != 0;
        if (!ohas) return;
        if (!has) {
            mn = omn; mx = omx; imn = oimn; imx = oimx; has = true;
            return;
        }
        if (omn < mn || (omn == mn && oimn < imn)) { mn = omn; imn = oimn; }
        if (omx > mx || (omx == mx && oimx < imx)) { mx = omx; imx = oimx; }
    }
};

__global__ __launch_bounds__(K1_BLOCK) void k1_bbox_group(const double2 *__restrict__ xy,
                                                          const int32_t *__restrict__ pt_off, int64_t n_boxes,
                                                          double *__restrict__ out_box4,
                                                          int32_t *__restrict__ out_arg4) {
    constexpr int GROUPS = K1_BLOCK / K1_GROUP;
    const int gl = threadIdx.x & (K1_GROUP - 1);
    const int64_t stride = (int64_t)gridDim.x * GROUPS;
    // every lane of a group sees the same box, so the folds below only ever pair lanes that are in the same iteration
    for (int64_t b = (int64_t)blockIdx.x * GROUPS + threadIdx.x / K1_GROUP; b < n_boxes; b += stride) {
        const int32_t s = pt_off[b], e = pt_off[b + 1];
        GroupAcc ax, ay;
        ax.has = ay.has = false;
        ax.mn = ax.mx = ay.mn = ay.mx = __builtin_nan("");
        ax.imn = ax.imx = ay.imn = ay.imx = -1;
        for (int32_t p = s + gl; p < e; p += K1_GROUP) {
            const double2 v = xy[p];
            ax.take(v.x, p - s);
            ay.take(v.y, p - s);
        }
#pragma unroll
        for (int d = K1_GROUP / 2; d >= 1; d >>= 1) {
            ax.fold(d);
            ay.fold(d);
        }
        if (gl == 0) {   // an empty polygon leaves NaN / -1, like BoxAcc::empty()
            double2 *ob = reinterpret_cast<double2 *>(out_box4 + 4 * b);
            ob[0] = make_double2(ax.mn, ay.mn);
            ob[1] = make_double2(ax.mx, ay.mx);
            *reinterpret_cast<int4 *>(out_arg4 + 4 * b) = make_int4(ax.imn, ay.imn, ax.imx, ay.imx);
        }
    }
}

// dyd_set_option("k1_variant"): -1 = by the table's shape (default: the group kernel from 48 points per box on), 0 = the
// tile kernel, 1 = the direct kernel (no LDS), 2 = the group kernel
static int g_k1_variant = -1;
void set_k1_variant(int v) { g_k1_variant = v; }
bool k1_wants_groups(int64_t n_boxes, int64_t n_points) {   // also asked by the fused entry point
    if (g_k1_variant == 2) return true;
    return g_k1_variant < 0 && n_points > 48 * n_boxes;   // tools/points_sweep.py: 32 points per box 0.40 ms vs the tile kernels' 0.30, 64: 0.27 vs 0.36
}

int launch_k1(const double *xy, const int32_t *pt_off, int64_t n_boxes, int64_t n_points, double *out_box4,
              int32_t *out_arg4, hipStream_t st) {
    if (n_boxes == 0) return DYD_OK;
    if (k1_wants_groups(n_boxes, n_points)) {
        const int64_t want = ceil_div(n_boxes, (int64_t)(K1_BLOCK / K1_GROUP));
        const int64_t cap = (int64_t)ctx().num_cu * 16;   // grid-stride: a few workgroups per CU are enough
        hipLaunchKernelGGL(k1_bbox_group, dim3((unsigned)(want < cap ? want : cap)), dim3(K1_BLOCK), 0, st,
                           reinterpret_cast<const double2 *>(xy), pt_off, n_boxes, out_box4, out_arg4);
        DYD_HIP(hipGetLastError());
        return DYD_OK;
    }
    const int64_t tiles = ceil_div(n_boxes, K1_BLOCK);
    if (tiles > 0x7fffffffLL) {
        set_error("n_boxes=%lld exceeds one launch", (long long)n_boxes);
        return DYD_ERR_RANGE;
    }
    const double2 *pts = reinterpret_cast<const double2 *>(xy);
    if (g_k1_variant == 1)
        hipLaunchKernelGGL(k1_bbox_direct, dim3((unsigned)tiles), dim3(K1_BLOCK), 0, st, pts, pt_off,
                           n_boxes, out_box4, out_arg4);
    else
        hipLaunchKernelGGL(k1_bbox_lds, dim3((unsigned)tiles), dim3(K1_BLOCK), 0, st, pts, pt_off,
                           n_boxes, out_box4, out_arg4);
    DYD_HIP(hipGetLastError());
    return DYD_OK;
}

}  // namespace dyd

using namespace dyd;

extern "C" {

int dyd_bbox_minmax_dev(const double *xy, const int32_t *pt_off, int64_t n_boxes, int64_t n_points, double *out_box4,
                        int32_t *out_arg4, void *stream) {
    DYD_API_ENTER();
    DYD_REQUIRE(n_boxes >= 0, "n_boxes < 0");
    if (n_boxes == 0) return DYD_OK;
    DYD_REQUIRE(pt_off && out_box4 && out_arg4, "null pointer");
    DYD_REQUIRE((reinterpret_cast<uintptr_t>(xy) & 15) == 0, "xy must be 16-byte aligned");
    DYD_REQUIRE((reinterpret_cast<uintptr_t>(out_box4) & 15) == 0 &&
                    (reinterpret_cast<uintptr_t>(out_arg4) & 15) == 0,
                "outputs must be 16-byte aligned");
    return launch_k1(xy, pt_off, n_boxes, n_points, out_box4, out_arg4, pick_stream(stream));
}

int dyd_bbox_minmax(const double *xy, const int32_t *pt_off, int64_t n_boxes, double *out_box4,
                    int32_t *out_arg4) {
    DYD_API_ENTER();
    DYD_REQUIRE(n_boxes >= 0, "n_boxes < 0");
    if (n_boxes == 0) return DYD_OK;
    DYD_REQUIRE(pt_off && out_box4 && out_arg4, "null pointer");
    DYD_REQUIRE(pt_off[0] == 0, "pt_off[0] != 0");
    for (int64_t i = 0; i < n_boxes; ++i) DYD_REQUIRE(pt_off[i + 1] >= pt_off[i], "pt_off not monotone");
    const int64_t n_pts = pt_off[n_boxes];
    DYD_REQUIRE(n_pts == 0 || xy, "xy is null");
    DevBuf d_xy, d_off, d_box, d_arg;
    int rc;
    if ((rc = d_xy.alloc(16 * (size_t)n_pts)) || (rc = d_off.alloc(4 * (size_t)(n_boxes + 1))) ||
        (rc = d_box.alloc(32 * (size_t)n_boxes)) || (rc = d_arg.alloc(16 * (size_t)n_boxes)))
        return rc;
    hipStream_t st = ctx().stream;
    if (n_pts) DYD_HIP(hipMemcpyAsync(d_xy.p, xy, 16 * (size_t)n_pts, hipMemcpyHostToDevice, st));
    DYD_HIP(hipMemcpyAsync(d_off.p, pt_off, 4 * (size_t)(n_boxes + 1), hipMemcpyHostToDevice, st));
    KernelTimer t(st);
    rc = launch_k1(d_xy.as<double>(), d_off.as<int32_t>(), n_boxes, n_pts, d_box.as<double>(),
                   d_arg.as<int32_t>(), st);
    if (rc) return rc;
    t.finish();
    DYD_HIP(hipMemcpyAsync(out_box4, d_box.p, 32 * (size_t)n_boxes, hipMemcpyDeviceToHost, st));
    DYD_HIP(hipMemcpyAsync(out_arg4, d_arg.p, 16 * (size_t)n_boxes, hipMemcpyDeviceToHost, st));
    DYD_HIP(hipStreamSynchronize(st));
    return DYD_OK;
}

}  // extern "C"
