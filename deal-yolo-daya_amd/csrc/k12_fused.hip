// k12_fused.hip — poly -> bbox -> IoU flag for rows whose boxes all come from K1
// (reference ui/pages/processing.py:580-598 runs process_csv_replace_ptlist and
// filter_by_box_count_and_iou back to back on the same rows), plus the tuning hook.
//
// K1's out_box4 is (min_x, min_y, max_x, max_y), which is exactly the two-point ptList the
// reference's IoU step reads back (processor.py:260 -> :354-362), so K2 consumes it directly.
#include <cstring>

#include "dyd_common.h"

namespace dyd {
int launch_k1(const double *xy, const int32_t *pt_off, int64_t n_boxes, double *out_box4, int32_t *out_arg4,
              hipStream_t st);
int launch_k2(const double *box4, const int32_t *row_off, int64_t n_rows, int32_t min_boxes, double thr,
              uint8_t *out_high, double *out_max, hipStream_t st);
void set_k1_variant(int v);
}  // namespace dyd

using namespace dyd;

extern "C" {

int dyd_bbox_iou_fused_dev(const double *xy, const int32_t *pt_off, const int32_t *box_off, int64_t n_rows,
                           int64_t n_boxes, int32_t min_boxes, double thr, double *out_box4, int32_t *out_arg4,
                           uint8_t *out_high, void *stream) {
    DYD_API_ENTER();
    DYD_REQUIRE(n_rows >= 0 && n_boxes >= 0, "negative size");
    if (n_rows == 0) return DYD_OK;
    DYD_REQUIRE(pt_off && box_off && out_box4 && out_arg4 && out_high, "null pointer");
    DYD_REQUIRE(((reinterpret_cast<uintptr_t>(xy) | reinterpret_cast<uintptr_t>(out_box4) |
                  reinterpret_cast<uintptr_t>(out_arg4)) & 15) == 0,
                "xy / out_box4 / out_arg4 must be 16-byte aligned");
    hipStream_t st = pick_stream(stream);
    int rc = launch_k1(xy, pt_off, n_boxes, out_box4, out_arg4, st);
    if (rc) return rc;
    return launch_k2(out_box4, box_off, n_rows, min_boxes, thr, out_high, nullptr, st);
}

// Tuning / A-B hook (not part of the reference-facing ABI): selects kernel variants.
int dyd_set_option(const char *key, int64_t value) {
    std::lock_guard<std::recursive_mutex> lock(api_mutex());
    if (!key) return DYD_ERR_INVALID;
    if (!strcmp(key, "k1_variant")) {
        set_k1_variant((int)value);
        return DYD_OK;
    }
    set_error("unknown option %s", key);
    return DYD_ERR_INVALID;
}

}  // extern "C"
