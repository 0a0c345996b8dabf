/*
 * dyd_oracle.c — TEST INFRASTRUCTURE, not product code.
 *
 * Plain-C, single-threaded CPU restatement of the numeric cores of the reference's
 * annotation hot path (Cyclones-Y/Deal-Yolo-Daya, src/deal_yolo_data/core/processor.py).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it; the
 * product (deal-yolo-daya_amd/) never does.
 *
 * Parity pinning: the reference has no tests or golden vectors of its own
 * (tests/__init__.py is empty), so this file is pinned by the fixtures under tests/golden/, which
 * tests/golden/make_golden.py generated in the build container by importing and
 * running the reference itself; tests/test_oracle_golden.py replays them.
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off, no fast-math: every double
 * operation must round exactly like CPython's float arithmetic).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ---- K1 ---- get_bbox_points, processor.py:252-260 --------------------------------
 * builtin min(): best = first; for each later item: if item < best: best = item.
 * builtin max(): same with >.  A NaN never compares true, so it survives only as the
 * first element. */
void orc_bbox_minmax(const double *xy, const int32_t *pt_off, int64_t n_boxes, double *out_box4,
                     int32_t *out_arg4) {
    for (int64_t b = 0; b < n_boxes; ++b) {
        int32_t s = pt_off[b], e = pt_off[b + 1];
        double *o = out_box4 + 4 * b;
        int32_t *a = out_arg4 + 4 * b;
        if (e <= s) { /* processor.py:254-255 -> None, None */
            o[0] = o[1] = o[2] = o[3] = NAN;
            a[0] = a[1] = a[2] = a[3] = -1;
            continue;
        }
        double mnx = xy[2 * (int64_t)s], mny = xy[2 * (int64_t)s + 1];
        double mxx = mnx, mxy = mny;
        int32_t imnx = 0, imny = 0, imxx = 0, imxy = 0;
        for (int32_t p = s + 1; p < e; ++p) {
            double x = xy[2 * (int64_t)p], y = xy[2 * (int64_t)p + 1];
            if (x < mnx) { mnx = x; imnx = p - s; }
            if (x > mxx) { mxx = x; imxx = p - s; }
            if (y < mny) { mny = y; imny = p - s; }
            if (y > mxy) { mxy = y; imxy = p - s; }
        }
        o[0] = mnx; o[1] = mny; o[2] = mxx; o[3] = mxy;
        a[0] = imnx; a[1] = imny; a[2] = imxx; a[3] = imxy;
    }
}

/* two-argument builtin min(a, b) / max(a, b): a unless b is strictly better */
static inline double py_min2(double a, double b) { return (b < a) ? b : a; }
static inline double py_max2(double a, double b) { return (b > a) ? b : a; }

/* calculate_iou, processor.py:328-339 (boxes already corner-normalised) */
static double orc_iou(const double *p, const double *q) {
    double x1 = py_max2(p[0], q[0]);
    double y1 = py_max2(p[1], q[1]);
    double x2 = py_min2(p[2], q[2]);
    double y2 = py_min2(p[3], q[3]);
    double w = x2 - x1, h = y2 - y1;
    double inter = py_max2(0.0, w) * py_max2(0.0, h);
    if (inter == 0) return 0.0;
    double a1 = (p[2] - p[0]) * (p[3] - p[1]);
    double a2 = (q[2] - q[0]) * (q[3] - q[1]);
    double uni = a1 + a2 - inter;
    return (uni != 0) ? inter / uni : 0.0;
}

/* ---- K2 ---- meet_conditions :368-376 with the corner normalisation of
 * extract_boxes :359-362 applied to the stored two points first. */
void orc_iou_any_ge(const double *box4, const int32_t *row_off, int64_t n_rows, int32_t min_boxes,
                    double thr, uint8_t *out_high, double *out_max_iou) {
    int32_t cap = 0;
    for (int64_t r = 0; r < n_rows; ++r) {
        int32_t n = row_off[r + 1] - row_off[r];
        if (n > cap) cap = n;
    }
    double *nb = (double *)malloc(sizeof(double) * 4 * (size_t)(cap > 0 ? cap : 1));
    for (int64_t r = 0; r < n_rows; ++r) {
        int32_t s = row_off[r], n = row_off[r + 1] - s;
        for (int32_t i = 0; i < n; ++i) {
            const double *b = box4 + 4 * (int64_t)(s + i);
            nb[4 * i + 0] = py_min2(b[0], b[2]);
            nb[4 * i + 1] = py_min2(b[1], b[3]);
            nb[4 * i + 2] = py_max2(b[0], b[2]);
            nb[4 * i + 3] = py_max2(b[1], b[3]);
        }
        uint8_t high = 0;
        double mx = 0.0;
        if (out_max_iou) { /* diagnostic: no early exit */
            for (int32_t i = 0; i < n; ++i)
                for (int32_t j = i + 1; j < n; ++j) {
                    double v = orc_iou(nb + 4 * i, nb + 4 * j);
                    if (v > mx) mx = v;
                    if (n >= min_boxes && v >= thr) high = 1;
                }
            out_max_iou[r] = mx;
        } else if (n >= min_boxes) {
            for (int32_t i = 0; i < n && !high; ++i)
                for (int32_t j = i + 1; j < n; ++j)
                    if (orc_iou(nb + 4 * i, nb + 4 * j) >= thr) { high = 1; break; }
        }
        out_high[r] = high;
    }
    free(nb);
}

/* ---- K3 ---- MurmurHash3 x64_128 (A. Appleby, public domain algorithm), seed 0 ----- */
static inline uint64_t rotl64(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }
static inline uint64_t fmix64(uint64_t k) {
    k ^= k >> 33; k *= 0xff51afd7ed558ccdULL;
    k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ULL;
    k ^= k >> 33;
    return k;
}
static void murmur3_x64_128(const uint8_t *data, int64_t len, uint64_t *out) {
    const uint64_t c1 = 0x87c37b91114253d5ULL, c2 = 0x4cf5ad432745937fULL;
    uint64_t h1 = 0, h2 = 0;
    int64_t nblocks = len / 16;
    for (int64_t i = 0; i < nblocks; ++i) {
        uint64_t k1, k2;
        memcpy(&k1, data + 16 * i, 8);
        memcpy(&k2, data + 16 * i + 8, 8);
        k1 *= c1; k1 = rotl64(k1, 31); k1 *= c2; h1 ^= k1;
        h1 = rotl64(h1, 27); h1 += h2; h1 = h1 * 5 + 0x52dce729;
        k2 *= c2; k2 = rotl64(k2, 33); k2 *= c1; h2 ^= k2;
        h2 = rotl64(h2, 31); h2 += h1; h2 = h2 * 5 + 0x38495ab5;
    }
    const uint8_t *tail = data + 16 * nblocks;
    uint64_t k1 = 0, k2 = 0;
    int rem = (int)(len & 15);
    for (int i = rem - 1; i >= 8; --i) k2 ^= (uint64_t)tail[i] << (8 * (i - 8));
    if (rem > 8) { k2 *= c2; k2 = rotl64(k2, 33); k2 *= c1; h2 ^= k2; }
    for (int i = (rem > 8 ? 8 : rem) - 1; i >= 0; --i) k1 ^= (uint64_t)tail[i] << (8 * i);
    if (rem > 0) { k1 *= c1; k1 = rotl64(k1, 31); k1 *= c2; h1 ^= k1; }
    h1 ^= (uint64_t)len; h2 ^= (uint64_t)len;
    h1 += h2; h2 += h1;
    h1 = fmix64(h1); h2 = fmix64(h2);
    h1 += h2; h2 += h1;
    out[0] = h1; out[1] = h2;
}
void orc_hash128(const uint8_t *bytes, const int64_t *off, int64_t n, uint64_t *out) {
    for (int64_t i = 0; i < n; ++i) murmur3_x64_128(bytes + off[i], off[i + 1] - off[i], out + 2 * i);
}

/* ---- K4 ---- drop_duplicates(keep=first|last|False), processor.py:140-144 ---------- */
typedef struct { uint64_t a, b; int64_t i; } key_t3;
static int cmp_key(const void *x, const void *y) {
    const key_t3 *p = (const key_t3 *)x, *q = (const key_t3 *)y;
    if (p->a != q->a) return p->a < q->a ? -1 : 1;
    if (p->b != q->b) return p->b < q->b ? -1 : 1;
    return p->i < q->i ? -1 : (p->i > q->i);
}
void orc_dedup(const uint64_t *h, int64_t n, int keep_mode, uint8_t *out_keep) {
    key_t3 *k = (key_t3 *)malloc(sizeof(key_t3) * (size_t)(n > 0 ? n : 1));
    for (int64_t i = 0; i < n; ++i) { k[i].a = h[2 * i]; k[i].b = h[2 * i + 1]; k[i].i = i; }
    qsort(k, (size_t)n, sizeof(key_t3), cmp_key);
    for (int64_t s = 0; s < n;) {
        int64_t e = s + 1;
        while (e < n && k[e].a == k[s].a && k[e].b == k[s].b) ++e;
        for (int64_t t = s; t < e; ++t) out_keep[k[t].i] = 0;
        if (keep_mode == 0) out_keep[k[s].i] = 1;
        else if (keep_mode == 1) out_keep[k[e - 1].i] = 1;
        else if (e - s == 1) out_keep[k[s].i] = 1;
        s = e;
    }
    free(k);
}

/* ---- K5 ---- Series.isin(ref_values), processor.py:194-199 ------------------------- */
static int cmp_pair(const void *x, const void *y) {
    const uint64_t *p = (const uint64_t *)x, *q = (const uint64_t *)y;
    if (p[0] != q[0]) return p[0] < q[0] ? -1 : 1;
    if (p[1] != q[1]) return p[1] < q[1] ? -1 : 1;
    return 0;
}
void orc_isin(const uint64_t *h, int64_t n, const uint64_t *ref_h, int64_t r, uint8_t *out_mask) {
    uint64_t *s = (uint64_t *)malloc(16 * (size_t)(r > 0 ? r : 1));
    memcpy(s, ref_h, 16 * (size_t)r);
    qsort(s, (size_t)r, 16, cmp_pair);
    for (int64_t i = 0; i < n; ++i)
        out_mask[i] = (r > 0 && bsearch(h + 2 * i, s, (size_t)r, 16, cmp_pair)) ? 1 : 0;
    free(s);
}

/* ---- K6 ---- DataFrame.sample(frac=1, random_state=seed), processor.py:800 ---------
 * pandas -> RandomState(seed).choice(n, n, replace=False) -> permutation(n)[:n]
 * (numpy legacy mtrand): init_genrand(seed), then for i = n-1 .. 1:
 * j = random_interval(i) (32-bit draws masked to the smallest 2^k-1 >= i, rejected
 * while > i); swap(a[i], a[j]). */
typedef struct { uint32_t mt[624]; int pos; } mt_t;
static void mt_seed(mt_t *s, uint32_t seed) {
    s->mt[0] = seed;
    for (int i = 1; i < 624; ++i) s->mt[i] = 1812433253U * (s->mt[i - 1] ^ (s->mt[i - 1] >> 30)) + (uint32_t)i;
    s->pos = 624;
}
static uint32_t mt_next(mt_t *s) {
    if (s->pos >= 624) {
        uint32_t *mt = s->mt;
        for (int k = 0; k < 624; ++k) {
            uint32_t y = (mt[k] & 0x80000000U) | (mt[(k + 1) % 624] & 0x7fffffffU);
            mt[k] = mt[(k + 397) % 624] ^ (y >> 1) ^ ((y & 1U) ? 0x9908b0dfU : 0U);
        }
        s->pos = 0;
    }
    uint32_t y = s->mt[s->pos++];
    y ^= y >> 11; y ^= (y << 7) & 0x9d2c5680U; y ^= (y << 15) & 0xefc60000U; y ^= y >> 18;
    return y;
}
void orc_mt19937_permutation(uint32_t seed, int64_t n, int64_t *out) {
    mt_t s;
    mt_seed(&s, seed);
    for (int64_t i = 0; i < n; ++i) out[i] = i;
    for (int64_t i = n - 1; i >= 1; --i) {
        uint64_t mask = (uint64_t)i;
        mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4;
        mask |= mask >> 8; mask |= mask >> 16; mask |= mask >> 32;
        uint64_t v;
        if ((uint64_t)i <= 0xffffffffULL) {
            do { v = mt_next(&s) & mask; } while (v > (uint64_t)i);
        } else {
            do { uint64_t hi = mt_next(&s); uint64_t lo = mt_next(&s); v = ((hi << 32) | lo) & mask; } while (v > (uint64_t)i);
        }
        int64_t t = out[i]; out[i] = out[v]; out[v] = t;
    }
}

/* split ids, processor.py:800-806: the row with in-category rank r sits at shuffled
 * position k where perm[k] == r; k < n_train -> train, < n_train+n_val -> val, else test */
void orc_split_ids(const int32_t *cat, int64_t n, const int64_t *perm, const int64_t *cat_off,
                   const int64_t *n_train, const int64_t *n_val, int32_t n_cat, uint8_t *out_split,
                   int64_t *out_pos) {
    int64_t total = cat_off[n_cat];
    int64_t *inv = (int64_t *)malloc(sizeof(int64_t) * (size_t)(total > 0 ? total : 1));
    int64_t *seen = (int64_t *)calloc((size_t)(n_cat > 0 ? n_cat : 1), sizeof(int64_t));
    for (int32_t c = 0; c < n_cat; ++c)
        for (int64_t k = cat_off[c]; k < cat_off[c + 1]; ++k) inv[cat_off[c] + perm[k]] = k - cat_off[c];
    for (int64_t i = 0; i < n; ++i) {
        int32_t c = cat[i];
        if (c < 0 || c >= n_cat) { out_split[i] = 255; out_pos[i] = -1; continue; }
        int64_t pos = inv[cat_off[c] + seen[c]++];
        out_pos[i] = pos;
        out_split[i] = pos < n_train[c] ? 0 : (pos < n_train[c] + n_val[c] ? 1 : 2);
    }
    free(inv);
    free(seen);
}

/* ---- K7: YOLO label lines -- restates processor.py:1046-1052 (per-box arithmetic, "%.6f") and :1054
 * ("\n".join).  glibc's printf rounds the exact binary value half-to-even like CPython's float format.
 * Two-call protocol: out_text == NULL measures; flags: 0 text, 1 no line, 2 zero width / height or a
 * negative class id (the reference decides those before the arithmetic, :1016). */
#include <stdio.h>
int64_t orc_yolo_lines(const double *box4, const int32_t *row_off, const uint8_t *sel, const double *width,
                       const double *height, const int32_t *class_id, int64_t n_rows, int64_t *out_off,
                       uint8_t *out_flag, uint8_t *out_text) {
    int64_t pos = 0;
    char line[4 * 330 + 32];
    for (int64_t r = 0; r < n_rows; ++r) {
        out_off[r] = pos;
        const double w = width[r], h = height[r];
        if (w == 0.0 || h == 0.0 || class_id[r] < 0) {
            out_flag[r] = 2;
            continue;
        }
        int lines = 0;
        for (int32_t b = row_off[r]; b < row_off[r + 1]; ++b) {
            if (sel && !sel[b]) continue;
            const double *p = box4 + 4 * (int64_t)b;
            const double x1 = py_min2(p[0], p[2]), x2 = py_max2(p[0], p[2]);
            const double y1 = py_min2(p[1], p[3]), y2 = py_max2(p[1], p[3]);
            const double bw = py_max2(x2 - x1, 0.0), bh = py_max2(y2 - y1, 0.0);
            if (bw <= 0.0 || bh <= 0.0) continue;
            double v[4] = {(x1 + x2) / 2.0 / w, (y1 + y2) / 2.0 / h, bw / w, bh / h};
            int n = snprintf(line, sizeof line, "%s%d", lines ? "\n" : "", class_id[r]);
            for (int k = 0; k < 4; ++k) {
                if (v[k] != v[k]) n += snprintf(line + n, sizeof line - (size_t)n, " nan");   /* never "-nan" */
                else n += snprintf(line + n, sizeof line - (size_t)n, " %.6f", v[k]);
            }
            if (out_text) memcpy(out_text + pos, line, (size_t)n);
            pos += n;
            ++lines;
        }
        out_flag[r] = lines ? 0 : 1;
    }
    out_off[n_rows] = pos;
    return pos;
}
