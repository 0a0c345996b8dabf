// k1_tile.h — device code of K1 (polygon ptList -> bbox), shared by k1_bbox.hip and the fused
// K1+K2 kernel.  See k1_bbox.hip for the semantics (reference core/processor.py:252-260).
#pragma once

#include "dyd_common.h"

namespace dyd {

constexpr int K1_BLOCK = 256;
constexpr int K1_CHUNK = 2048;  // points per LDS chunk: 32 KiB -> 5 workgroups (20 waves) per CU

struct BoxAcc {
    double mnx, mny, mxx, mxy;
    int32_t imnx, imny, imxx, imxy;
    __device__ __forceinline__ void first(double x, double y) {
        mnx = mxx = x;
        mny = mxy = y;
        imnx = imny = imxx = imxy = 0;
    }
    // CPython's builtin min/max: the running best is replaced only by a STRICTLY better item
    __device__ __forceinline__ void next(double x, double y, int32_t k) {
        if (x < mnx) { mnx = x; imnx = k; }
        if (x > mxx) { mxx = x; imxx = k; }
        if (y < mny) { mny = y; imny = k; }
        if (y > mxy) { mxy = y; imxy = k; }
    }
    __device__ __forceinline__ void empty() {
        mnx = mny = mxx = mxy = __builtin_nan("");
        imnx = imny = imxx = imxy = -1;
    }
    // NT: non-temporal stores — for a kernel whose boxes are not read again from memory (the fused K1+K2 hands them to K2
    // through LDS): 2 % on back-to-back launches (0.624 -> 0.610 ms), the 0.8 GB of output no longer displaces the point stream
    template <bool NT = false>
    __device__ __forceinline__ void store(double *out_box4, int32_t *out_arg4, int64_t b) const {
        if (NT) {
            typedef double v2d __attribute__((ext_vector_type(2)));
            typedef int v4i __attribute__((ext_vector_type(4)));
            v2d *ob = reinterpret_cast<v2d *>(out_box4 + 4 * b);
            const v2d lo2 = {mnx, mny}, hi2 = {mxx, mxy};
            const v4i ai = {imnx, imny, imxx, imxy};
            __builtin_nontemporal_store(lo2, ob);
            __builtin_nontemporal_store(hi2, ob + 1);
            __builtin_nontemporal_store(ai, reinterpret_cast<v4i *>(out_arg4 + 4 * b));
            return;
        }
        double2 *ob = reinterpret_cast<double2 *>(out_box4 + 4 * b);
        ob[0] = make_double2(mnx, mny);
        ob[1] = make_double2(mxx, mxy);
        *reinterpret_cast<int4 *>(out_arg4 + 4 * b) = make_int4(imnx, imny, imxx, imxy);
    }
};

// One tile: boxes [b0, min(b0 + K1_BLOCK, b_end)), lane t owns box b0 + t.  The tile's contiguous
// point range is streamed HBM -> LDS (s_pts, CHUNK points) with coalesced 16-B-per-lane loads;
// each lane then walks its box's points in LDS in their original order.  Must be called by all
// K1_BLOCK threads of the workgroup (it contains workgroup barriers).
template <int CHUNK = K1_CHUNK>
__device__ __forceinline__ void k1_process_tile(const double2 *__restrict__ xy,
                                                const int32_t *__restrict__ pt_off, int64_t b0, int64_t b_end,
                                                double *out_box4, int32_t *out_arg4, double2 *s_pts) {
    const int tid = threadIdx.x;
    const int64_t b = b0 + tid;
    const bool active = b < b_end;
    const int64_t b1 = (b0 + K1_BLOCK < b_end) ? b0 + K1_BLOCK : b_end;
    const int32_t ts = pt_off[b0];  // tile's point range (workgroup-uniform)
    const int32_t te = pt_off[b1];
    int32_t s = 0, e = 0;
    if (active) {
        s = pt_off[b];
        e = pt_off[b + 1];
    }
    BoxAcc acc;
    acc.empty();
    for (int32_t cs = ts; cs < te; cs += CHUNK) {
        const int32_t ce = (te - cs > CHUNK) ? cs + CHUNK : te;
        __syncthreads();  // the previous chunk (or the previous tile / phase) is fully consumed
        for (int32_t p = cs + tid; p < ce; p += K1_BLOCK) s_pts[p - cs] = xy[p];
        __syncthreads();
        int32_t lo = s > cs ? s : cs;
        const int32_t hi = e < ce ? e : ce;
        if (lo < hi) {
            if (lo == s) {
                const double2 v = s_pts[lo - cs];
                acc.first(v.x, v.y);
                ++lo;
            }
            for (int32_t p = lo; p < hi; ++p) {
                const double2 v = s_pts[p - cs];
                acc.next(v.x, v.y, p - s);
            }
        }
    }
    if (active) acc.store(out_box4, out_arg4, b);
}

}  // namespace dyd
