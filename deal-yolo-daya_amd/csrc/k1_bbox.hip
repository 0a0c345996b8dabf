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

static int g_k1_variant = 0;
void set_k1_variant(int v) { g_k1_variant = v; }

int launch_k1(const double *xy, const int32_t *pt_off, int64_t n_boxes, double *out_box4,
              int32_t *out_arg4, hipStream_t st) {
    if (n_boxes == 0) return DYD_OK;
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

int dyd_bbox_minmax_dev(const double *xy, const int32_t *pt_off, int64_t n_boxes, double *out_box4,
                        int32_t *out_arg4, void *stream) {
    DYD_API_ENTER();
    DYD_REQUIRE(n_boxes >= 0, "n_boxes < 0");
    if (n_boxes == 0) return DYD_OK;
    DYD_REQUIRE(pt_off && out_box4 && out_arg4, "null pointer");
    DYD_REQUIRE((reinterpret_cast<uintptr_t>(xy) & 15) == 0, "xy must be 16-byte aligned");
    DYD_REQUIRE((reinterpret_cast<uintptr_t>(out_box4) & 15) == 0 &&
                    (reinterpret_cast<uintptr_t>(out_arg4) & 15) == 0,
                "outputs must be 16-byte aligned");
    return launch_k1(xy, pt_off, n_boxes, out_box4, out_arg4, pick_stream(stream));
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
    rc = launch_k1(d_xy.as<double>(), d_off.as<int32_t>(), n_boxes, d_box.as<double>(),
                   d_arg.as<int32_t>(), st);
    if (rc) return rc;
    t.finish();
    DYD_HIP(hipMemcpyAsync(out_box4, d_box.p, 32 * (size_t)n_boxes, hipMemcpyDeviceToHost, st));
    DYD_HIP(hipMemcpyAsync(out_arg4, d_arg.p, 16 * (size_t)n_boxes, hipMemcpyDeviceToHost, st));
    DYD_HIP(hipStreamSynchronize(st));
    return DYD_OK;
}

}  // extern "C"
