/*
 * dyd.h — C ABI of libdyd_gfx950.so, the MI355X (gfx950) device stage of the
 * annotation hot path of Cyclones-Y/Deal-Yolo-Daya.
 *
 * The reference (`src/deal_yolo_data/core/processor.py`) is pure Python and has no
 * FFI of its own; every entry point below replaces a Python loop or a pandas call of
 * that file and is what a ctypes binding added to the reference would bind
 * (INTEGRATION.md shows that binding).  Each declaration cites the reference lines
 * it replaces.
 *
 * Conventions
 *   - every function returns DYD_OK (0) or a negative DYD_ERR_* code; the message
 *     for the calling thread's last failure is dyd_last_error().  No C++ exception
 *     crosses this boundary.
 *   - all buffers are caller-owned and contiguous.  Functions without a suffix take
 *     HOST pointers and stage H2D / D2H themselves; `_dev` twins take DEVICE
 *     pointers (from dyd_malloc, or any hipMalloc'ed memory of the same device, e.g.
 *     a torch tensor's data_ptr) plus the hipStream_t to launch on, used as given
 *     (NULL = HIP's null stream).  `_dev` calls are asynchronous on that stream.
 *   - offsets arrays have n+1 entries, start at 0 and are non-decreasing.
 *   - the library keeps one lazily created context per process (device, stream,
 *     scratch); entry points are serialised by a process-wide mutex, so they may be
 *     called from any thread (Streamlit runs each session's script on its own
 *     thread: reference ui/pages/processing.py:200-213).
 */
#ifndef DYD_H
#define DYD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DYD_OK 0
#define DYD_ERR_INVALID (-1)   /* bad argument (null pointer, negative size, bad offsets) */
#define DYD_ERR_NO_DEVICE (-2) /* no gfx950 device visible / dyd_init failed */
#define DYD_ERR_HIP (-3)       /* a HIP runtime call or kernel launch failed */
#define DYD_ERR_OOM (-4)       /* device allocation failed */
#define DYD_ERR_RANGE (-5)     /* size exceeds what an int32 offset array can address */

/* keep modes of dyd_dedup == pandas drop_duplicates(keep=...) (processor.py:140-144) */
#define DYD_KEEP_FIRST 0
#define DYD_KEEP_LAST 1
#define DYD_KEEP_NONE 2

/* ---- context -------------------------------------------------------------------- */
int dyd_init(int device_or_minus1);
void dyd_shutdown(void);
const char *dyd_last_error(void);
int dyd_device_count(void);
const char *dyd_version(void);
/* name of the device the context is bound to ("" before dyd_init) */
const char *dyd_device_name(void);

/* ---- device memory + timing (for callers that keep data resident in HBM) -------- */
int dyd_malloc(void **dptr, size_t bytes);
int dyd_free(void *dptr);
int dyd_h2d(void *dst_dev, const void *src_host, size_t bytes);
int dyd_d2h(void *dst_host, const void *src_dev, size_t bytes);
int dyd_memset(void *dst_dev, int byte, size_t bytes);
int dyd_sync(void *stream);
/* Device-side failures (K4 / K5: hash table full, an inserted key not found again) are recorded in one status word on the device.
 * Host-pointer entry points check it themselves and return DYD_ERR_HIP; a caller of the asynchronous `_dev` twins asks here once its
 * launches are queued: synchronises `stream`, returns DYD_OK or DYD_ERR_HIP (message in dyd_last_error) and clears the word. */
int dyd_device_status(void *stream);
/* elapsed ms of the kernels launched by the calling thread's most recent non-_dev
 * entry point (hipEvent pair around the kernel launches only, staging excluded) */
double dyd_last_kernel_ms(void);

/* ---- K1: polygon ptList -> bbox -------------------------------------------------
 * Replaces get_bbox_points (processor.py:252-260), called per object at :273.
 * xy      : P points, interleaved (x,y) f64                       [2*P]
 * pt_off  : point offsets per box                                 [n_boxes+1]
 * out_box4: (min_x, min_y, max_x, max_y) per box                  [4*n_boxes]
 * out_arg4: index INSIDE the box of the point that supplies each of the four values,
 *           in the same order                                     [4*n_boxes]
 * Semantics are CPython's builtin min/max over the box's points in order: the FIRST
 * extremal element wins (strict < / > replaces the running best), so -0.0 vs 0.0 and
 * int-vs-float ties resolve to the lower index, and a NaN wins only from position 0.
 * A box with no points yields four NaNs and four -1 (the host emits JSON null,
 * processor.py:254-255). */
int dyd_bbox_minmax(const double *xy, const int32_t *pt_off, int64_t n_boxes,
                    double *out_box4, int32_t *out_arg4);
/* n_points = pt_off[n_boxes] (the table's shape picks the kernel: a lane per polygon for short polygons, sixteen lanes per
 * polygon from 48 points per polygon on); a negative value means "not known" and selects the short-polygon kernels. */
int dyd_bbox_minmax_dev(const double *xy, const int32_t *pt_off, int64_t n_boxes, int64_t n_points,
                        double *out_box4, int32_t *out_arg4, void *stream);

/* ---- K2: per-image box-count + all-pairs IoU filter ------------------------------
 * Replaces meet_conditions (processor.py:368-376) + calculate_iou (:328-339) and the
 * corner normalisation of extract_boxes (:359-362).
 * box4    : per box the two ptList points as stored (p1x, p1y, p2x, p2y); corners are
 *           re-normalised in the kernel with first-wins min/max       [4*B]
 * row_off : box offsets per image row                                  [n_rows+1]
 * out_high: 1 iff n_i >= min_boxes and some pair i<j has IoU >= thr    [n_rows]
 * out_max_iou_or_null: optional diagnostic, max pair IoU of the row (0.0 when the row
 *           has fewer than two boxes); not a reference output          [n_rows]
 * f64 arithmetic in the reference's operation order, no contraction, IEEE division. */
int dyd_iou_any_ge(const double *box4, const int32_t *row_off, int64_t n_rows,
                   int32_t min_boxes, double thr, uint8_t *out_high,
                   double *out_max_iou_or_null);
/* n_boxes = row_off[n_rows] (picks the kernel by the table's shape; negative: not known) */
int dyd_iou_any_ge_dev(const double *box4, const int32_t *row_off, int64_t n_rows, int64_t n_boxes,
                       int32_t min_boxes, double thr, uint8_t *out_high,
                       double *out_max_iou_or_null, void *stream);

/* ---- K1+K2 fused: poly -> bbox -> IoU flag in one pass ---------------------------
 * One launch that produces K1's outputs and K2's flag for rows whose boxes all come
 * from K1 (processing.py:580-598 runs the two steps back to back on the same rows).
 * box_off : box offsets per image row [n_rows+1], n_boxes = box_off[n_rows], n_points = pt_off[n_boxes]
 *           (or negative: not known); other arguments as K1 / K2.
 * The flag is the one the reference's two steps produce in sequence, not merely "K2 on K1's boxes": a polygon
 * without a valid point is written as a ptList of null coordinates by the replace step (processor.py:254-255), and
 * extract_boxes of the IoU step raises on it inside its blanket try (:359 -> :364-365), so the row's box list is the
 * PREFIX before that object — out_high[r] is computed over that prefix (out_box4 / out_arg4 still cover every box).
 * dyd_bbox_iou_fused is the host-pointer twin (stages the inputs, copies out_arg4 / out_high back, out_box4 only when
 * it is not NULL — the JSON emitter needs the arg indices, not the values). */
int dyd_bbox_iou_fused(const double *xy, const int32_t *pt_off, const int32_t *box_off, int64_t n_rows,
                       int32_t min_boxes, double thr, double *out_box4_or_null, int32_t *out_arg4,
                       uint8_t *out_high);
/* The same pass through a STAGING SLOT the caller holds across many calls: a stream, two timing events, a device arena and a
 * pinned host arena, all owned by the context and kept between calls (nothing is created, allocated or freed per pass once the
 * arenas have grown to the working size).  dyd_stage_acquire hands out a free slot — a new one while fewer than 64 exist, else it
 * waits — with its pinned arena grown to pinned_bytes if the budget (DYD_PINNED_POOL_MB, default 1024 for all slots together)
 * allows: *pinned / *pinned_cap tell what the caller got (possibly less, possibly nothing).  Host arrays of the _staged call may
 * lie inside that arena (then every copy is an asynchronous DMA) or anywhere else.  dyd_bbox_iou_fused is acquire(0) + staged +
 * release.  The native replace -> IoU pass (dyd_json_replace_iou) scans straight into the pinned arena. */
typedef struct dyd_stage dyd_stage;
int dyd_stage_acquire(size_t pinned_bytes, dyd_stage **out, void **pinned, size_t *pinned_cap);
void dyd_stage_release(dyd_stage *stage);
int dyd_bbox_iou_fused_staged(dyd_stage *stage, const double *xy, const int32_t *pt_off, const int32_t *box_off, int64_t n_rows,
                              int32_t min_boxes, double thr, double *out_box4_or_null, int32_t *out_arg4, uint8_t *out_high);
int dyd_bbox_iou_fused_dev(const double *xy, const int32_t *pt_off, const int32_t *box_off,
                           int64_t n_rows, int64_t n_boxes, int64_t n_points, int32_t min_boxes, double thr,
                           double *out_box4, int32_t *out_arg4, uint8_t *out_high, void *stream);

/* ---- K3: 128-bit hash of a string column -----------------------------------------
 * The equality test inside DataFrame.drop_duplicates (processor.py:140) and
 * Series.isin (:198) is replaced by equality of 128-bit hashes of the host's
 * canonical byte form of each cell (MurmurHash3 x64_128, seed 0).
 * bytes: concatenated cells; off: byte offsets [n+1]; out_hi_lo: (h1, h2) per row [2*n] */
int dyd_hash128(const uint8_t *bytes, const int64_t *off, int64_t n, uint64_t *out_hi_lo);
int dyd_hash128_dev(const uint8_t *bytes, const int64_t *off, int64_t n, uint64_t *out_hi_lo,
                    void *stream);

/* ---- K4: first / last / none-occurrence mask over hash keys ----------------------
 * Replaces drop_duplicates(subset=["source"], keep=keep) (processor.py:140-144).
 * h: (h1,h2) per row [2*n]; out_keep[i] = 1 iff row i survives. */
int dyd_dedup(const uint64_t *h, int64_t n, int keep_mode, uint8_t *out_keep);
int dyd_dedup_dev(const uint64_t *h, int64_t n, int keep_mode, uint8_t *out_keep, void *stream);

/* ---- K5: membership of main keys in a reference key set --------------------------
 * Replaces Series.isin(ref_values) (processor.py:194-199). out_mask[i] = 1 iff h[i] is
 * one of the r reference keys. */
int dyd_isin(const uint64_t *h, int64_t n, const uint64_t *ref_h, int64_t r, uint8_t *out_mask);
/* Verification of hash equality.  K4 / K5 call two cells equal when their 128-bit hashes are; pandas compares values
 * (processor.py:140-144, :198).  out_partner[i] = the first row whose hash equals row i's (i itself for a first occurrence) /
 * the reference row that main row i hit (-1: no hit); the host then compares the BYTES of every such pair
 * (dyd_host_cells_differ, multithreaded host code: cells given as flat text + offsets, optionally through index arrays; pairs with
 * a negative index are skipped; returns the number of pairs that differ, out_differs marks them) — zero means every match was a
 * match of values.  One extra gather per row on the device; the step functions do this by default (verify=True). */
int dyd_dedup_partner(const uint64_t *h, int64_t n, int64_t *out_partner);
int dyd_isin_partner(const uint64_t *h, int64_t n, const uint64_t *ref_h, int64_t r, int64_t *out_partner);
int64_t dyd_host_cells_differ(const uint8_t *text_a, const int64_t *off_a, const int64_t *idx_a, const uint8_t *text_b, const int64_t *off_b,
                              const int64_t *idx_b, int64_t n, int n_threads, uint8_t *out_differs_or_null);
int dyd_isin_dev(const uint64_t *h, int64_t n, const uint64_t *ref_h, int64_t r,
                 uint8_t *out_mask, void *stream);

/* ---- multi-GPU -------------------------------------------------------------------------------------------------
 * One process per GPU; rows are sharded contiguously; K1 / K2 / K7 need no communication.  The path's one real exchange — the
 * all-gather of each shard's locally unique 16-byte keys (and of the reference keys) — is issued by the host layer
 * (deal-yolo-daya_amd/distributed.py) through the process group the caller already has: torch.distributed with backend "nccl",
 * which IS RCCL over xGMI on ROCm, on the stream the `_dev` kernels run on.  SURVEY §8b sketched a library-owned communicator
 * (dyd_comm_init(nranks, rank, rccl_unique_id) + dyd_comm* variants); it is deliberately NOT part of this ABI: a second
 * bootstrap (unique-id exchange) and a second RCCL communicator beside the caller's, for a single all-gather, would duplicate
 * state the caller must own anyway (device binding, stream, process-group lifetime).  The library's side of a sharded call is
 * therefore the plain `_dev` entry points on the shard's arrays: dyd_hash128_dev, dyd_dedup_dev (shard-local pre-dedup),
 * dyd_isin_dev (probe of the local survivors against the other ranks' keys), dyd_split_ids_seeded_dev (with cat_rank_base),
 * and dyd_device_status.  dyd_dedup_global_dev below is the older form (every rank inserts ALL gathered keys). */

/* ---- multi-GPU dedup: keys of ALL ranks after the allgather ----------------------
 * all_h : gathered keys of every rank in global row order          [2*n_all]
 * first_global / n_local: this rank owns global rows [first_global, first_global+n_local)
 * out_keep: mask for the rank's own rows                             [n_local] */
int dyd_dedup_global_dev(const uint64_t *all_h, int64_t n_all, int64_t first_global,
                         int64_t n_local, int keep_mode, uint8_t *out_keep, void *stream);

/* ---- K6: train/val/test split ids -------------------------------------------------
 * Replaces DataFrame.sample(frac=1, random_state=seed) + the int(n*ratio) cuts per
 * category (processor.py:796-806).
 * dyd_mt19937_permutation is host code: numpy's legacy RandomState(seed).permutation(n)
 * (init_genrand + reversed Fisher-Yates with masked-rejection 32-bit draws).
 * cat            : category id per expanded row, -1 = unclassified     [n]
 * cat_perm_concat: per category its permutation, concatenated           [sum n_c]
 * cat_off        : start of each category inside cat_perm_concat        [n_cat+1]
 * n_train/n_val  : cut sizes per category                               [n_cat]
 * out_split      : 0 train / 1 val / 2 test / 255 unclassified          [n]
 * out_pos        : position of the row inside its shuffled category     [n] (-1 unclassified) */
int dyd_mt19937_permutation(uint32_t seed, int64_t n, int64_t *out);
int dyd_split_ids(const int32_t *cat, int64_t n, const int64_t *cat_perm_concat,
                  const int64_t *cat_off, const int64_t *n_train, const int64_t *n_val,
                  int32_t n_cat, uint8_t *out_split, int64_t *out_pos);
int dyd_split_ids_dev(const int32_t *cat, int64_t n, const int64_t *cat_perm_concat,
                      const int64_t *cat_off, const int64_t *n_train, const int64_t *n_val,
                      int32_t n_cat, uint8_t *out_split, int64_t *out_pos, void *stream);

/* ---- K8: the permutation itself on the device ------------------------------------------------------------------
 * numpy's legacy RandomState(seed).permutation(n) (what DataFrame.sample(frac=1, random_state=seed) shuffles with, :800) computed
 * in parallel on the GPU, identical to dyd_mt19937_permutation element for element: MT19937 stream by one workgroup, the masked
 * rejection resolved by iterated device-wide scans, the Fisher-Yates swap chain replaced by a closed form over one radix sort
 * (csrc/k8_perm.hip).  out_perm[k] = the value at shuffled position k, out_inverse[v] = the shuffled position of value v; either
 * may be NULL.  n <= 2^30.  Synchronous (it reads back a few words between rounds). */
int dyd_mt19937_permutation_dev(uint32_t seed, int64_t n, int64_t *out_perm_or_null, int64_t *out_inverse_or_null, void *stream);
/* K6 with the permutations made by K8 from `seed` (every category is shuffled with the same random_state, :800): no permutation
 * array crosses the boundary and no inversion pass is needed (K8 yields the inverse K6 looks positions up in).  cat_sizes /
 * n_train / n_val (and cat_rank_base for a shard, see dyd_split_ids_sharded_dev) are HOST arrays [n_cat]; cat / out_* as in
 * dyd_split_ids (device pointers for _dev, host pointers otherwise).  Synchronous. */
int dyd_split_ids_seeded(const int32_t *cat, int64_t n, uint32_t seed, const int64_t *cat_sizes, const int64_t *n_train,
                         const int64_t *n_val, int32_t n_cat, uint8_t *out_split, int64_t *out_pos);
int dyd_split_ids_seeded_dev(const int32_t *cat, int64_t n, uint32_t seed, const int64_t *cat_sizes_host, const int64_t *n_train_host,
                             const int64_t *n_val_host, int32_t n_cat, const int64_t *cat_rank_base_host_or_null, uint8_t *out_split,
                             int64_t *out_pos, void *stream);

/* multi-GPU K6: the rank holds a contiguous shard of the expanded rows; cat_rank_base[c] = number
 * of rows of category c held by lower ranks (from one allgather of per-rank category counts),
 * cat_off / perm / n_train / n_val describe the GLOBAL categories. */
int dyd_split_ids_sharded_dev(const int32_t *cat, int64_t n, const int64_t *cat_perm_concat,
                              const int64_t *cat_off, const int64_t *n_train, const int64_t *n_val,
                              int32_t n_cat, const int64_t *cat_rank_base, uint8_t *out_split,
                              int64_t *out_pos, void *stream);

/* ---- K7: YOLO label lines ----------------------------------------------
 * Replaces the per-box arithmetic and "%.6f" formatting of generate_yolo_datasets_from_excels
 * (processor.py:1046-1052) and the "\n".join of a row's lines (:1054):
 *   x1,x2 = min,max; y1,y2 = min,max; bw = max(x2-x1, 0.0); bh = max(y2-y1, 0.0); skip if bw<=0 or bh<=0;
 *   "{cid} {(x1+x2)/2/width:.6f} {(y1+y2)/2/height:.6f} {bw/width:.6f} {bh/height:.6f}"
 * box4         : boxes as (x1, y1, x2, y2) f64, any corner order              [4*n_boxes]
 * row_off      : boxes of row i are [row_off[i], row_off[i+1])               [n_rows+1]
 * sel_or_null  : 1 = the box carries the row's label (b[0] == label_value, :1006), NULL = all  [n_boxes]
 * width/height : the row's image size (:1013-1014)                            [n_rows]
 * class_id     : class_to_id[label_value] (:1049)                             [n_rows]
 * out_text_off : byte range of row i in the text = [off[i], off[i+1])         [n_rows+1]
 * out_flag     : 0 text written, 1 no line (the reference skips the row: 标注框无效 / 无匹配标签框),
 *                2 left to the host: zero width/height (the reference tests `not width` first, :1016),
 *                negative class id, or a value >= 2^43 whose "%.6f" has up to 316 characters  [n_rows]
 * dyd_yolo_lines     : host pointers; *out_text is allocated by the library (release with dyd_host_free).
 * dyd_yolo_lines_dev : device pointers; out_text_or_null == NULL only measures (offsets, flags, total);
 *                      otherwise text_cap bytes are available and DYD_ERR_RANGE is returned, with the needed
 *                      size in *out_total, when that is too little.  *out_total is a HOST int64.  n_boxes = row_off[n_rows]
 *                      (the table's shape picks the kernel; a negative value makes the entry read it back from the device). */
int dyd_yolo_lines(const double *box4, const int32_t *row_off, const uint8_t *sel_or_null,
                   const double *width, const double *height, const int32_t *class_id, int64_t n_rows,
                   int64_t *out_text_off, uint8_t *out_flag, uint8_t **out_text, int64_t *out_text_len);
int dyd_yolo_lines_dev(const double *box4, const int32_t *row_off, const uint8_t *sel_or_null,
                       const double *width, const double *height, const int32_t *class_id, int64_t n_rows,
                       int64_t n_boxes, int64_t *out_text_off, uint8_t *out_flag, uint8_t *out_text_or_null, int64_t text_cap,
                       int64_t *out_total, void *stream);

/* ---- native flatten / emit (HOST code, multithreaded) ------------------------------
 * Schema-specialised JSON scanner + canonical re-emitter that replaces json.loads / json.dumps inside
 * parse_and_replace_ptlist (processor.py:262-281), extract_width_height (:285-292) and extract_boxes
 * (:341-366).  Cells are passed as concatenated UTF-8 text + offsets; `missing[i]` marks NaN cells.
 * status per cell: 0 regular, 1 undecodable JSON (the reference yields None / no boxes), 2 irregular
 * (the caller must process the cell with the Python flatten of flatten.py), 3 missing.
 * The handle owns every array the accessors return; free it with dyd_scan_free. */
typedef struct dyd_scan dyd_scan;
int dyd_json_scan_polygons(const uint8_t *text, const int64_t *cell_off, const uint8_t *missing,
                           int64_t n_cells, int n_threads, dyd_scan **out);
int dyd_json_emit_polygons(dyd_scan *scan, const uint8_t *text, const int64_t *cell_off,
                           const int32_t *arg4, int n_threads, const uint8_t **out_text,
                           const int64_t **out_off);
int dyd_json_scan_boxes(const uint8_t *text, const int64_t *cell_off, const uint8_t *missing,
                        int64_t n_cells, int n_threads, dyd_scan **out);
int64_t dyd_scan_n_boxes(const dyd_scan *scan);
int64_t dyd_scan_n_points(const dyd_scan *scan);
const double *dyd_scan_xy(const dyd_scan *scan);              /* points [2*P] (polygons) or box4 [4*B] (boxes) */
const int32_t *dyd_scan_pt_off(const dyd_scan *scan);         /* [n_boxes+1] */
const int32_t *dyd_scan_cell_box_off(const dyd_scan *scan);   /* [n_cells+1] */
const uint8_t *dyd_scan_status(const dyd_scan *scan);         /* [n_cells] */
const uint8_t *dyd_scan_wh_kind(const dyd_scan *scan, int which);   /* which: 0 width, 1 height; 0 none 1 int 2 float 3 other */
const double *dyd_scan_wh_value(const dyd_scan *scan, int which);
/* polygon scan only, [n_cells]: 1 = some coordinate of the cell is an int beyond 2^25.  calculate_iou (processor.py:328-339)
 * multiplies coordinate differences in CPython's exact int arithmetic; f64 follows it only while every product stays below
 * 2^53, so the fused K1+K2 flag of such a cell is not used: the host decides it from the emitted boxes (flatten.py). */
const uint8_t *dyd_scan_iou_host(const dyd_scan *scan);
/* polygon scan only: how many cells the single-parse lane (csrc/host_json_fast.h) took; the others went through the exact walker */
int64_t dyd_scan_fast_cells(const dyd_scan *scan);
/* dyd_json_scan_polygons over one (pointer, length) pair per cell instead of a flat buffer — e.g. the UTF-8 views of a DataFrame
 * column's str objects, so that no cell is copied.  The pointers must stay valid until dyd_scan_free; dyd_json_emit_polygons may
 * then be called with text == cell_off == NULL. */
int dyd_json_scan_polygons_v(const uint8_t *const *cell_ptr, const int64_t *cell_len, const uint8_t *missing, int64_t n_cells,
                             int n_threads, dyd_scan **out);
/* YOLO step (utils.py:681-710, processor.py:1006): per cell the (min x, min y, max x, max y) of every named object
 * with a non-empty ptList, in dyd_scan_xy as box4, and dyd_scan_sel[b] = 1 when the object's name equals the row's
 * label value (label_text / label_off: one label per cell).  Undecodable cells give no boxes, like the reference's
 * blanket except; status 2 cells are left to the Python path. */
int dyd_json_scan_labelled(const uint8_t *text, const int64_t *cell_off, const uint8_t *missing, int64_t n_cells,
                           const uint8_t *label_text, const int64_t *label_off, int n_threads, dyd_scan **out);
const uint8_t *dyd_scan_sel(const dyd_scan *scan);             /* [n_boxes] (labelled scan only) */
/* The replace step and the IoU step in ONE native pass (processor.py:262-281 then :341-376; ui/pages/processing.py:580-598 runs them
 * back to back): cells as flat text + offsets, or as one (pointer, length) pair per cell (text == cell_off == NULL).  Every worker
 * thread holds one staging slot (dyd_stage_acquire) and takes its share of the cells through scan -> dyd_bbox_iou_fused_staged -> emit
 * chunk by chunk (DYD_PIPE_CHUNK_KB of cell text, default 8192), the points scanned straight into the slot's pinned arena.  The handle then
 * holds per cell: dyd_scan_status, dyd_scan_high (the IoU step's flag, meaningful for status 0 cells; the caller decides cells with
 * dyd_scan_iou_host != 0 and status 2 cells itself), dyd_scan_wh_*; and the emitted text per part (dyd_scan_part) or gathered
 * (dyd_scan_text).  Needs the device (no CPU fallback). */
int dyd_json_replace_iou(const uint8_t *text, const int64_t *cell_off, const uint8_t *const *cell_ptr, const int64_t *cell_len,
                         const uint8_t *missing, int64_t n_cells, int32_t min_boxes, double thr, int n_threads, dyd_scan **out);
/* Emitted text outlives the pass that wrote it; a freed handle parks those blocks (up to DYD_HOST_POOL_MB, default 3072) so that the
 * next pass writes into warm memory instead of tearing 2 GB of page tables down and faulting them in again.  This gives them back. */
void dyd_host_pool_trim(void);
const uint8_t *dyd_scan_high(const dyd_scan *scan);            /* [n_cells] */
int32_t dyd_scan_parts(const dyd_scan *scan);
int dyd_scan_part(const dyd_scan *scan, int32_t k, int64_t *lo, int64_t *hi, const uint8_t **text, const int64_t **off);
int dyd_scan_text(dyd_scan *scan, const uint8_t **text, const int64_t **off);
void dyd_scan_totals(const dyd_scan *scan, int64_t *counts3, double *seconds3);   /* boxes, points, fast-lane cells | scan, device, emit s */
void dyd_scan_free(dyd_scan *scan);
/* measurement / test aid (host, multithreaded): the annotation cells of a synthetic table as json.dumps would write them
 * (deal-yolo-daya_amd/synth.py: row_json) — flat utf-8 in *out_text (release with dyd_host_free) and offsets [n_rows+1]. */
int dyd_synth_json(const double *xy, const int32_t *pt_off, const int32_t *box_off, const int32_t *label, const uint8_t *int_row,
                   int64_t n_rows, int64_t width, int64_t height, int n_threads, uint8_t **out_text, int64_t *out_off);

/* ---- native expansion of the split step (HOST code, multithreaded) -----------------------------------
 * Replaces the per-row Python of split_dataset_by_rules (processor.py:712-792; utils.py:645-662): every row
 * becomes one record per (object, label of the object's name found in the rules), the record's JSON being the
 * row's document with "objects" reduced to that object and its "name" set to the label (:760-767).
 * text/cell_off/missing: the rows' JSON cells (missing = no usable cell: "空数据"); label_text/label_off: the
 * keys of label_to_category.  Per cell: status (0 expanded, 1 空数据, 2 JSON解析失败, 3 objects不是列表,
 * 4 标注字段objects为空, 5 irregular: the Python path decides), number of records, the label combination
 * ("，".join(sorted(labels)), :736) and the joined reasons ("；".join(sorted(标签…未在规则中定义)), :779).
 * Records in row order: cell index, label index, JSON text.  Events in the order the reference appends to its
 * unclassified list: cell index, kind (1 标注框缺少name字段, 2 label not in the rules, 3 nothing classified),
 * the label for kind 2. */
typedef struct dyd_split dyd_split;
int dyd_json_split_expand(const uint8_t *text, const int64_t *cell_off, const uint8_t *missing, int64_t n_cells,
                          const uint8_t *label_text, const int64_t *label_off, int32_t n_labels, int n_threads,
                          dyd_split **out);
const uint8_t *dyd_split_status(const dyd_split *h);          /* [n_cells] */
const int32_t *dyd_split_n_expanded(const dyd_split *h);      /* [n_cells] */
int64_t dyd_split_rows(const dyd_split *h);
const int64_t *dyd_split_row_cell(const dyd_split *h);        /* [rows] */
const int32_t *dyd_split_row_label(const dyd_split *h);       /* [rows] */
int64_t dyd_split_events(const dyd_split *h);
const int64_t *dyd_split_event_cell(const dyd_split *h);      /* [events] */
const uint8_t *dyd_split_event_kind(const dyd_split *h);      /* [events] */
/* which: 0 record JSON [rows] (a flat copy, made on the first request), 1 label combination [n_cells], 2 joined reasons
 * [n_cells], 3 event label [events] (made on the first request), 4 the distinct undefined labels [dyd_split_undefined] */
int dyd_split_strings(dyd_split *h, int which, const uint8_t **data, const int64_t **off);
/* the same expansion over one (pointer, length) view per cell — the str objects of a DataFrame column, nothing copied */
int dyd_json_split_expand_v(const uint8_t *const *cell_ptr, const int64_t *cell_len, const uint8_t *missing, int64_t n_cells,
                            const uint8_t *label_text, const int64_t *label_off, int32_t n_labels, int n_threads,
                            dyd_split **out);
/* table-scale accessors: the record texts stay in the worker threads' buffers, one (address, length) view per record in row
 * order (valid until dyd_split_free); an event of kind 2 carries the index of its label in the table of distinct undefined
 * labels (-1 otherwise); per label of the rules the first record carrying it (-1: none) and its number of records — the
 * first-appearance order of the categories (processor.py:773) without a pass over the records */
int dyd_split_rec_views(const dyd_split *h, const uint64_t **ptr, const int64_t **len);
const int32_t *dyd_split_event_code(const dyd_split *h);      /* [events] */
int64_t dyd_split_undefined(const dyd_split *h);
const int64_t *dyd_split_label_first(const dyd_split *h);     /* [n_labels] */
const int64_t *dyd_split_label_count(const dyd_split *h);     /* [n_labels] */
int64_t dyd_split_fast_cells(const dyd_split *h);             /* cells the single-parse lane took */
int dyd_split_all_ascii(const dyd_split *h);                  /* 1: every record text is pure ASCII */
/* The reasons of a table are few distinct texts (they name a row's undefined labels): per cell the index of its text among the
 * distinct ones (-1: no reasons), or NULL when there were more than 4096 of them; their number; per distinct text the first cell
 * carrying it (its bytes are dyd_split_strings(h, 2) at that cell). */
const int32_t *dyd_split_reason_code(const dyd_split *h);
int64_t dyd_split_reason_distinct(const dyd_split *h);
const int64_t *dyd_split_reason_first(const dyd_split *h);
void dyd_split_seconds(const dyd_split *h, double *parse_gather2);
void dyd_split_free(dyd_split *h);

/* ---- native relabelling of the label_replace step (HOST code, multithreaded) ------------------------------
 * Replaces the per-row Python of replace_labels_by_mapping (processor.py:565-609; utils.py:659-679): in every
 * document whose "objects" is a list, the name of each dict element is split into labels, the labels found among
 * the keys are replaced, the result de-duplicated, sorted and joined with ","; the whole document is re-serialised
 * as json.dumps(..., ensure_ascii=False) does.  key_text/key_off, val_text/val_off: the old -> new label pairs.
 * Per cell: status (0 rewritten, 1 skipped: no usable cell, 2 JSON decode error, 3 left as it is: "objects" absent
 * or not a list, 5 irregular: the Python path decides), five counters (objects, names missing, labels seen, labels
 * replaced, objects renamed), whether some name would change, and for those the old / new names joined with "；"
 * (:605-609).  The labels that are not keys of the mapping, in order of appearance, with their cell. */
typedef struct dyd_relabel dyd_relabel;
int dyd_json_relabel(const uint8_t *text, const int64_t *cell_off, const uint8_t *missing, int64_t n_cells,
                     const uint8_t *key_text, const int64_t *key_off, const uint8_t *val_text, const int64_t *val_off,
                     int32_t n_pairs, int n_threads, dyd_relabel **out);
const uint8_t *dyd_relabel_status(const dyd_relabel *h);        /* [n_cells] */
const uint8_t *dyd_relabel_has_diff(const dyd_relabel *h);      /* [n_cells] */
const int32_t *dyd_relabel_counts(const dyd_relabel *h);        /* [n_cells][5] */
int64_t dyd_relabel_tokens(const dyd_relabel *h);
const int64_t *dyd_relabel_token_cell(const dyd_relabel *h);    /* [tokens] */
/* which: 0 text after the step [n_cells] (the cell itself unless status 0; empty for status 1), 1 joined old names [n_cells], 2 joined new names [n_cells], 3 unmatched label [tokens] */
int dyd_relabel_strings(const dyd_relabel *h, int which, const uint8_t **data, const int64_t **off);
void dyd_relabel_free(dyd_relabel *h);

/* ---- native CSV hand-off (HOST code) ------------------------------------------------
 * Replaces pandas read_csv / to_csv around the two heavy JSON columns (processor.py:235, :309, :379,
 * :404, :407).  dyd_csv_index tokenises a utf-8 buffer with pandas' C-parser conventions and FAILS on
 * anything it does not reproduce exactly (the caller then uses pandas); dyd_csv_extract returns one
 * column as flat utf-8 + offsets + a per-cell class (0 text, 1 missing = one of pandas' default NA strings,
 * 2 text that dtype inference could read as a number / boolean); dyd_csv_project returns CSV text
 * of the same width in which only the `keep` columns carry their cells, for pandas itself to parse with
 * usecols (same width => same low-memory piece boundaries => same per-piece dtype inference as on the
 * original file); dyd_csv_write writes typed column buffers
 * (kind 0 utf-8, 1 int64, 2 float64, 3 bool) like DataFrame.to_csv(index=False). */
typedef struct dyd_csv dyd_csv;
typedef struct dyd_csv_col {
    int32_t kind;
    const void *data;
    const int64_t *off;
    const uint8_t *na;
} dyd_csv_col;
int dyd_csv_index(const uint8_t *text, int64_t len, dyd_csv **out);
int64_t dyd_csv_rows(const dyd_csv *csv);
int32_t dyd_csv_cols(const dyd_csv *csv);
int64_t dyd_csv_header(const dyd_csv *csv, int32_t col, uint8_t *buf, int64_t cap);
int dyd_csv_extract(dyd_csv *csv, int32_t col, const uint8_t **bytes, const int64_t **off, const uint8_t **na);
int dyd_csv_project(dyd_csv *csv, const int32_t *keep, int32_t n_keep, const uint8_t **text, int64_t *len);
int64_t dyd_csv_col_bytes(const dyd_csv *csv, int32_t col);   /* total cell bytes of a column */
int64_t dyd_csv_row_end(const dyd_csv *csv, int64_t row);     /* byte offset behind data row `row` (-1: the header) */
int dyd_csv_has_cr(const dyd_csv *csv);                       /* 1: some lines end with "\r\n" */
void dyd_csv_free(dyd_csv *csv);
/* mode 0: write `path` anew, 1: to memory (*mem_out, dyd_host_free), 2: append to `path` (merge step, :73-76) */
int dyd_csv_write(const char *path, const uint8_t *header, int64_t header_len, const dyd_csv_col *cols,
                  int32_t n_cols, int64_t n_rows, const int64_t *rows, int64_t n_sel, int quote_cr,
                  int n_threads, int mode, uint8_t **mem_out, int64_t *mem_len);
void dyd_host_free(void *p);

/* ---- tuning hook (not reference-facing): selects kernel variants for A/B measurement,
 * e.g. dyd_set_option("k1_variant", 1) = K1 without LDS staging; "k7_variant": -1 by the table's shape (default),
 * 2 / 22 row tiles (one / two per ticket), 30 box tiles (rows of many boxes); "fused_variant": -1 by the table's shape (default:
 * 4 up to 32 boxes per image on average, 10 beyond, 6 / 9 for polygons of 20..48 points), 4 = wave kernel, 10 = its dense
 * instantiation (rows of 40..256 boxes sorted and swept), 6 / 9 = workgroup tilings, 1 = two launches;
 * "k2_variant": -1 by shape, 4 = the wave kernel's pair stage, 0..3 / 5 = tile kernels (2 / 3 / 5 with the f32 filter and the sweep). */
int dyd_set_option(const char *key, int64_t value);
/* measurement aid: plain streaming kernel (mode 0 copy, 1 read-only, 2 write-only, 3-5 the same non-temporal, 16 B per
 * lane) used to record the box's HBM ceiling next to the kernels' achieved GB/s; modes 6 / 7 / 8: one 8-byte word per lane
 * at a pseudo-random place of `dst` (scatter / gather / atomicMin, every word once) — the ceiling K4/K5/K6 are quoted against;
 * modes 10 / 11 / 12: the fused kernel's 72 % read / 28 % write mix (five 16-byte loads in flight per lane, two 16-byte stores;
 * plain, non-temporal stores, non-temporal both): `bytes` of src are read, 2/5 of that written to dst. */
int dyd_membench_dev(int mode, const void *src, void *dst, int64_t bytes, int blocks, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* DYD_H */
