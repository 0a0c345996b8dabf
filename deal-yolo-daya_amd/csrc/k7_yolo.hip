// k7_yolo.hip — K7: YOLO label lines.
//
// Replaces the per-box arithmetic and string formatting of generate_yolo_datasets_from_excels
// (reference core/processor.py:1046-1052):
//
//     x1, x2 = min(x1, x2), max(x1, x2);  y1, y2 = min(y1, y2), max(y1, y2)
//     bw = max(x2 - x1, 0.0);  bh = max(y2 - y1, 0.0);  skip the box if bw <= 0 or bh <= 0
//     f"{cid} {(x1 + x2) / 2 / width:.6f} {(y1 + y2) / 2 / height:.6f} {bw / width:.6f} {bh / height:.6f}"
//
// and "\n".join(lines) per row (:1054).  The f64 operations are done in the reference's order
// (IEEE add / sub / div, -ffp-contract=off) and "%.6f" is printed EXACTLY: |v| * 10^6 is formed as a
// 128-bit integer times a power of two and rounded half-to-even on the exact binary value, which is what
// CPython's correctly rounded float formatting does.  Rows whose numbers the device does not print (a
// value of 2^43 or more, a zero width / height, a negative class id) are flagged for the host.
//
// Layout in HBM: box4 = B x (x1, y1, x2, y2) f64 (corner order as found), row_off = N+1 int32, optional
// sel = B u8 (box belongs to the row's label), width / height = N f64, class_id = N int32.  Outputs:
// text_off = N+1 int64, flag = N u8, text = T bytes (rows back to back, no terminator).
// Algorithmic bytes per launch: 32*B [+ B] + 4*(N+1) + 16*N + 4*N in, 8*(N+1) + N + T out.  Bound: HBM.
//
// One pass: a workgroup takes the next tile of 256 rows from a ticket counter, measures its rows (one row
// per lane), scans the lengths in the workgroup, obtains the byte offset of the tile with a decoupled
// look-back over the tiles before it (one 8-byte {flag, bytes} word per tile, written and polled with
// agent-scope relaxed atomics; a tile publishes its own byte count before it looks back, so no tile waits
// for more than the measuring of its predecessors), prints the rows into LDS at the same 16-byte phase as
// their place in the output, and streams the LDS image out with 16-byte stores.
#include "dyd_common.h"

namespace dyd {

constexpr int K7_BLOCK = 256;
constexpr int K7_WAVES = K7_BLOCK / kWave;
constexpr int K7_LDS_TEXT = 24 * 1024;          // bytes of text staged per tile (typical tile: 256 x 38 B)
constexpr uint64_t K7_FLAG_AGG = 1ull << 62;    // the word holds the tile's own byte count
constexpr uint64_t K7_FLAG_PFX = 2ull << 62;    // the word holds the inclusive prefix up to this tile
constexpr uint64_t K7_VALUE = (1ull << 62) - 1;
constexpr int K7_SPIN_LIMIT = 1 << 22;          // polls before a tile gives up (sets the error word)

enum : uint32_t { K7_FINITE = 0, K7_NAN = 1, K7_INF = 2, K7_EXOTIC = 3 };

struct Num6 {
    uint64_t q;       // round-half-even(|v| * 10^6) for finite values below 2^43
    uint32_t kind;
    uint32_t neg;     // sign bit of v (printed also for -0.0 and for values that round to zero)
};

// exact |v| * 10^6 rounded half-to-even
__device__ __forceinline__ Num6 classify(double v) {
    const uint64_t bits = (uint64_t)__double_as_longlong(v);
    Num6 r;
    r.neg = (uint32_t)(bits >> 63);
    r.q = 0;
    const uint32_t ex = (uint32_t)(bits >> 52) & 0x7ffu;
    const uint64_t frac = bits & ((1ull << 52) - 1);
    if (ex == 0x7ffu) {
        r.kind = frac ? K7_NAN : K7_INF;
        return r;
    }
    if (ex >= 1023u + 43u) {  // |v| >= 2^43: up to 309 integer digits, printed by the host
        r.kind = K7_EXOTIC;
        return r;
    }
    r.kind = K7_FINITE;
    const uint64_t m = ex ? (frac | (1ull << 52)) : frac;
    const uint32_t s = 1075u - (ex ? ex : 1u);  // |v| = m * 2^-s, 10 <= s <= 1074
    // P = m * 10^6 < 2^73 as (hi, lo)
    const uint64_t lo = m * 1000000ull;
    const uint64_t hi = __umul64hi(m, 1000000ull);
    uint64_t q;
    bool up;
    if (s >= 128u) {          // P < 2^73 < half of 2^s: rounds to zero
        q = 0;
        up = false;
    } else if (s > 64u) {     // 65..127: quotient and half bit live in hi
        const uint32_t t = s - 64u;
        q = hi >> t;
        const uint64_t rem_hi = hi & ((1ull << t) - 1);
        const uint64_t half_hi = 1ull << (t - 1);
        up = (rem_hi > half_hi) || (rem_hi == half_hi && (lo != 0 || (q & 1)));
    } else if (s == 64u) {
        q = hi;
        const uint64_t half = 1ull << 63;
        up = (lo > half) || (lo == half && (q & 1));
    } else {                  // 10..63
        q = (hi << (64u - s)) | (lo >> s);
        const uint64_t rem = lo & ((1ull << s) - 1);
        const uint64_t half = 1ull << (s - 1);
        up = (rem > half) || (rem == half && (q & 1));
    }
    r.q = q + (up ? 1 : 0);
    return r;
}

__device__ __forceinline__ int digits_u64(uint64_t v) {  // v < 10^14
    int n = 1;
    if (v >= 10000000ull) { v /= 10000000ull; n += 7; }  // now v < 10^7
    uint32_t w = (uint32_t)v;
    if (w >= 10000u) { w /= 10000u; n += 4; }
    if (w >= 100u) { w /= 100u; n += 2; }
    if (w >= 10u) n += 1;
    return n;
}

__device__ __forceinline__ int num_len(const Num6 &n) {
    if (n.kind == K7_NAN) return 3;
    if (n.kind == K7_INF) return 3 + (int)n.neg;
    return (int)n.neg + digits_u64(n.q / 1000000ull) + 7;
}

template <class Put>
__device__ __forceinline__ int num_put(const Num6 &n, int pos, Put put) {
    if (n.kind == K7_NAN) {
        put(pos, 'n'); put(pos + 1, 'a'); put(pos + 2, 'n');
        return pos + 3;
    }
    if (n.neg) put(pos++, '-');
    if (n.kind == K7_INF) {
        put(pos, 'i'); put(pos + 1, 'n'); put(pos + 2, 'f');
        return pos + 3;
    }
    uint64_t ip = n.q / 1000000ull;
    uint32_t fp = (uint32_t)(n.q - ip * 1000000ull);
    const int nd = digits_u64(ip);
    for (int k = nd - 1; k >= 0; --k) {
        const uint64_t t = ip / 10;
        put(pos + k, (char)('0' + (int)(ip - t * 10)));
        ip = t;
    }
    pos += nd;
    put(pos, '.');
#pragma unroll
    for (int k = 6; k >= 1; --k) {
        const uint32_t t = fp / 10u;
        put(pos + k, (char)('0' + (int)(fp - t * 10u)));
        fp = t;
    }
    return pos + 7;
}

struct Line {
    Num6 v[4];
    bool valid;   // the box gives a line
    bool exotic;
};

// the reference's arithmetic for one box (processor.py:1046-1052), first-wins min / max as in CPython
__device__ __forceinline__ Line box_line(const double *__restrict__ b, double w, double h) {
    const double ax = b[0], ay = b[1], bx = b[2], by = b[3];
    const double x1 = (bx < ax) ? bx : ax, x2 = (bx > ax) ? bx : ax;
    const double y1 = (by < ay) ? by : ay, y2 = (by > ay) ? by : ay;
    const double dx = x2 - x1, dy = y2 - y1;
    const double bw = (0.0 > dx) ? 0.0 : dx;   // max(dx, 0.0): NaN stays NaN
    const double bh = (0.0 > dy) ? 0.0 : dy;
    Line l;
    l.exotic = false;
    l.valid = !(bw <= 0.0 || bh <= 0.0);
    if (!l.valid) return l;
    l.v[0] = classify((x1 + x2) / 2.0 / w);
    l.v[1] = classify((y1 + y2) / 2.0 / h);
    l.v[2] = classify(bw / w);
    l.v[3] = classify(bh / h);
    l.exotic = l.v[0].kind == K7_EXOTIC || l.v[1].kind == K7_EXOTIC || l.v[2].kind == K7_EXOTIC || l.v[3].kind == K7_EXOTIC;
    return l;
}

__device__ __forceinline__ int cid_digits(uint32_t c) {
    int n = 1;
    if (c >= 100000u) { c /= 100000u; n += 5; }
    if (c >= 1000u) { c /= 1000u; n += 3; }   // c < 10^5 here
    if (c >= 100u) n += 2;
    else if (c >= 10u) n += 1;
    return n;
}

struct RowIn {
    int32_t b0, b1;
    double w, h;
    int32_t cid;
    bool host;   // zero width / height or negative class id: the host decides
};

// bytes of the row's text; flag: 0 text, 1 no line, 2 host
__device__ __forceinline__ uint32_t row_measure(const RowIn &r, const double *__restrict__ box4,
                                                const uint8_t *__restrict__ sel, uint32_t &flag) {
    if (r.host) {
        flag = 2;
        return 0;
    }
    uint32_t len = 0, lines = 0;
    const int cd = cid_digits((uint32_t)r.cid);
    for (int32_t b = r.b0; b < r.b1; ++b) {
        if (sel && !sel[b]) continue;
        const Line l = box_line(box4 + 4 * (int64_t)b, r.w, r.h);
        if (!l.valid) continue;
        if (l.exotic) {
            flag = 2;
            return 0;
        }
        len += (uint32_t)(cd + 4 + num_len(l.v[0]) + num_len(l.v[1]) + num_len(l.v[2]) + num_len(l.v[3]));
        ++lines;
    }
    flag = lines ? 0 : 1;
    return lines ? len + lines - 1 : 0;
}

template <class Put>
__device__ __forceinline__ void row_print(const RowIn &r, const double *__restrict__ box4,
                                          const uint8_t *__restrict__ sel, Put put) {
    int pos = 0;
    const int cd = cid_digits((uint32_t)r.cid);
    for (int32_t b = r.b0; b < r.b1; ++b) {
        if (sel && !sel[b]) continue;
        const Line l = box_line(box4 + 4 * (int64_t)b, r.w, r.h);
        if (!l.valid) continue;
        if (pos) put(pos++, '\n');
        uint32_t c = (uint32_t)r.cid;
        for (int k = cd - 1; k >= 0; --k) {
            const uint32_t t = c / 10u;
            put(pos + k, (char)('0' + (int)(c - t * 10u)));
            c = t;
        }
        pos += cd;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            put(pos++, ' ');
            pos = num_put(l.v[k], pos, put);
        }
    }
}

// state[0] = ticket counter, state[1] = error word, state[2 + t] = look-back word of tile t
__global__ __launch_bounds__(K7_BLOCK) void k7_yolo_kernel(const double *__restrict__ box4,
                                                           const int32_t *__restrict__ row_off,
                                                           const uint8_t *__restrict__ sel,
                                                           const double *__restrict__ width,
                                                           const double *__restrict__ height,
                                                           const int32_t *__restrict__ class_id, int64_t n_rows,
                                                           int64_t *__restrict__ text_off,
                                                           uint8_t *__restrict__ flag_out, uint8_t *text,
                                                           int64_t text_cap, unsigned long long *state) {
    __shared__ __attribute__((aligned(16))) unsigned char s_text[K7_LDS_TEXT + 16];
    __shared__ uint32_t s_wave[K7_WAVES];
    __shared__ unsigned long long s_bcast[2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) s_bcast[0] = atomicAdd(&state[0], 1ull);
    __syncthreads();
    const int64_t tile = (int64_t)s_bcast[0];
    unsigned long long *words = state + 2;
    const int64_t row = tile * K7_BLOCK + tid;

    // ---- measure --------------------------------------------------------------------------------
    RowIn r;
    uint32_t len = 0, flag = 0;
    const bool live = row < n_rows;
    if (live) {
        r.b0 = row_off[row];
        r.b1 = row_off[row + 1];
        r.w = width[row];
        r.h = height[row];
        r.cid = class_id[row];
        r.host = (r.w == 0.0) || (r.h == 0.0) || (r.cid < 0);
        len = row_measure(r, box4, sel, flag);
    }
    // ---- exclusive scan of len in the workgroup ---------------------------------------------------
    uint32_t incl = len;
#pragma unroll
    for (int d = 1; d < kWave; d <<= 1) {
        const uint32_t up = __shfl_up(incl, d);
        if (lane >= d) incl += up;
    }
    if (lane == kWave - 1) s_wave[wave] = incl;
    __syncthreads();
    uint32_t wave_base = 0, tile_bytes = 0;
#pragma unroll
    for (int w = 0; w < K7_WAVES; ++w) {
        if (w < wave) wave_base += s_wave[w];
        tile_bytes += s_wave[w];
    }
    const uint32_t toff = wave_base + incl - len;

    // ---- decoupled look-back (wave 0) -------------------------------------------------------------
    if (wave == 0) {
        if (lane == 0) {
            const unsigned long long mine = (tile == 0 ? K7_FLAG_PFX : K7_FLAG_AGG) | (unsigned long long)tile_bytes;
            __hip_atomic_store(&words[tile], mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        unsigned long long base = 0;
        int64_t look = tile - 1;      // nearest tile not yet accounted for
        bool failed = false;
        while (look >= 0) {
            const int64_t t = look - lane;
            unsigned long long wv = K7_FLAG_PFX;   // lanes before tile 0 read as an empty prefix
            if (t >= 0) {
                int spins = 0;
                do {
                    wv = __hip_atomic_load(&words[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if ((wv >> 62) == 0 && ++spins > K7_SPIN_LIMIT) {
                        failed = true;
                        break;
                    }
                    if ((wv >> 62) == 0) __builtin_amdgcn_s_sleep(1);
                } while ((wv >> 62) == 0);
            }
            if (__any(failed)) {
                failed = true;
                break;
            }
            const unsigned long long has_pfx = __ballot((wv >> 62) == 2);
            const int first = has_pfx ? __ffsll((long long)has_pfx) - 1 : kWave;   // nearest lane holding a prefix
            unsigned long long part = (lane <= first) ? (wv & K7_VALUE) : 0ull;
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) part += __shfl_xor(part, d);
            base += part;
            if (has_pfx) break;
            look -= kWave;
        }
        if (lane == 0) {
            if (failed) {
                atomicExch(&state[1], 1ull);
                base = 0;
            }
            if (tile != 0)
                __hip_atomic_store(&words[tile], K7_FLAG_PFX | ((base + tile_bytes) & K7_VALUE), __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
            s_bcast[1] = base;
        }
    }
    __syncthreads();
    const int64_t base = (int64_t)s_bcast[1];

    if (live) {
        text_off[row] = base + toff;
        flag_out[row] = (uint8_t)flag;
        if (row == n_rows - 1) text_off[n_rows] = base + toff + len;
    }
    if (!text || tile_bytes == 0) return;
    if (base + (int64_t)tile_bytes > text_cap) {   // the host sees the total in text_off[n_rows] and reports it
        if (tid == 0) atomicExch(&state[1], 2ull);
        return;
    }
    unsigned char *dst = text + base;
    if (tile_bytes <= (uint32_t)K7_LDS_TEXT) {
        const uint32_t phase = (uint32_t)(reinterpret_cast<uintptr_t>(dst) & 15u);
        if (live && len) {
            unsigned char *mine = s_text + phase + toff;
            row_print(r, box4, sel, [&](int p, char c) { mine[p] = (unsigned char)c; });
        }
        __syncthreads();
        // s_text[phase + i] -> dst[i]; 16-byte chunk k of the LDS image maps to the aligned address dst - phase + 16k
        const uint32_t end = phase + tile_bytes;
        const uint32_t n_chunks = (end + 15u) >> 4;
        unsigned char *aligned = dst - phase;
        for (uint32_t k = tid; k < n_chunks; k += K7_BLOCK) {
            const uint32_t lo = k << 4, hi = lo + 16u;
            if (lo >= phase && hi <= end) {
                *reinterpret_cast<uint4 *>(aligned + lo) = *reinterpret_cast<const uint4 *>(s_text + lo);
            } else {
                const uint32_t a = lo < phase ? phase : lo, b = hi > end ? end : hi;
                for (uint32_t i = a; i < b; ++i) aligned[i] = s_text[i];
            }
        }
    } else if (live && len) {   // a tile of very long rows: print straight to memory
        unsigned char *mine = dst + toff;
        row_print(r, box4, sel, [&](int p, char c) { mine[p] = (unsigned char)c; });
    }
}

static int yolo_launch(const double *box4, const int32_t *row_off, const uint8_t *sel, const double *width,
                       const double *height, const int32_t *class_id, int64_t n_rows, int64_t *text_off,
                       uint8_t *flag, uint8_t *text, int64_t text_cap, int64_t *total_out, hipStream_t st) {
    const int64_t n_tiles = ceil_div(n_rows, K7_BLOCK);
    void *scr = nullptr;
    const size_t state_bytes = (size_t)(n_tiles + 2) * 8;
    int rc = get_scratch(state_bytes, &scr, st);
    if (rc) return rc;
    DYD_HIP(hipMemsetAsync(scr, 0, state_bytes, st));
    hipLaunchKernelGGL(k7_yolo_kernel, dim3((unsigned)n_tiles), dim3(K7_BLOCK), 0, st, box4, row_off, sel, width,
                       height, class_id, n_rows, text_off, flag, text, text_cap,
                       static_cast<unsigned long long *>(scr));
    DYD_HIP(hipGetLastError());
    unsigned long long err = 0;
    int64_t total = 0;
    DYD_HIP(hipMemcpyAsync(&err, static_cast<unsigned long long *>(scr) + 1, 8, hipMemcpyDeviceToHost, st));
    DYD_HIP(hipMemcpyAsync(&total, text_off + n_rows, 8, hipMemcpyDeviceToHost, st));
    DYD_HIP(hipStreamSynchronize(st));
    release_scratch(st);
    if (total_out) *total_out = total;
    if (err == 1) {
        set_error("K7: a tile waited too long for the tiles before it");
        return DYD_ERR_HIP;
    }
    if (err == 2) {
        set_error("K7: text buffer too small (%lld bytes needed, %lld given)", (long long)total, (long long)text_cap);
        return DYD_ERR_RANGE;
    }
    return DYD_OK;
}

}  // namespace dyd

using namespace dyd;

extern "C" {

int dyd_yolo_lines_dev(const double *box4, const int32_t *row_off, const uint8_t *sel_or_null, const double *width,
                       const double *height, const int32_t *class_id, int64_t n_rows, int64_t *out_text_off,
                       uint8_t *out_flag, uint8_t *out_text_or_null, int64_t text_cap, int64_t *out_total,
                       void *stream) {
    DYD_API_ENTER();
    DYD_REQUIRE(n_rows >= 0 && text_cap >= 0, "negative size");
    DYD_REQUIRE(out_text_off, "null pointer");
    hipStream_t st = pick_stream(stream);
    if (n_rows == 0) {
        DYD_HIP(hipMemsetAsync(out_text_off, 0, 8, st));
        if (out_total) *out_total = 0;
        return DYD_OK;
    }
    DYD_REQUIRE(row_off && width && height && class_id && out_flag, "null pointer");
    DYD_REQUIRE(n_rows < (1LL << 40), "n_rows too large");
    return yolo_launch(box4, row_off, sel_or_null, width, height, class_id, n_rows, out_text_off, out_flag,
                       out_text_or_null, text_cap, out_total, st);
}

int dyd_yolo_lines(const double *box4, const int32_t *row_off, const uint8_t *sel_or_null, const double *width,
                   const double *height, const int32_t *class_id, int64_t n_rows, int64_t *out_text_off,
                   uint8_t *out_flag, uint8_t **out_text, int64_t *out_text_len) {
    DYD_API_ENTER();
    DYD_REQUIRE(n_rows >= 0, "negative size");
    DYD_REQUIRE(out_text_off && out_text && out_text_len, "null pointer");
    *out_text = nullptr;
    *out_text_len = 0;
    out_text_off[0] = 0;
    if (n_rows == 0) return DYD_OK;
    DYD_REQUIRE(row_off && width && height && class_id && out_flag, "null pointer");
    DYD_REQUIRE(row_off[0] == 0, "row_off[0] != 0");
    for (int64_t i = 0; i < n_rows; ++i) DYD_REQUIRE(row_off[i + 1] >= row_off[i], "row_off not monotone");
    const int64_t n_boxes = row_off[n_rows];
    DYD_REQUIRE(n_boxes == 0 || box4, "null pointer");
    DevBuf d_box, d_off, d_sel, d_w, d_h, d_cid, d_toff, d_flag, d_text;
    int rc;
    if ((rc = d_box.alloc(32 * (size_t)n_boxes)) || (rc = d_off.alloc(4 * (size_t)(n_rows + 1))) ||
        (rc = d_sel.alloc((size_t)n_boxes)) || (rc = d_w.alloc(8 * (size_t)n_rows)) ||
        (rc = d_h.alloc(8 * (size_t)n_rows)) || (rc = d_cid.alloc(4 * (size_t)n_rows)) ||
        (rc = d_toff.alloc(8 * (size_t)(n_rows + 1))) || (rc = d_flag.alloc((size_t)n_rows)))
        return rc;
    hipStream_t st = ctx().stream;
    if (n_boxes) DYD_HIP(hipMemcpyAsync(d_box.p, box4, 32 * (size_t)n_boxes, hipMemcpyHostToDevice, st));
    if (n_boxes && sel_or_null) DYD_HIP(hipMemcpyAsync(d_sel.p, sel_or_null, (size_t)n_boxes, hipMemcpyHostToDevice, st));
    DYD_HIP(hipMemcpyAsync(d_off.p, row_off, 4 * (size_t)(n_rows + 1), hipMemcpyHostToDevice, st));
    DYD_HIP(hipMemcpyAsync(d_w.p, width, 8 * (size_t)n_rows, hipMemcpyHostToDevice, st));
    DYD_HIP(hipMemcpyAsync(d_h.p, height, 8 * (size_t)n_rows, hipMemcpyHostToDevice, st));
    DYD_HIP(hipMemcpyAsync(d_cid.p, class_id, 4 * (size_t)n_rows, hipMemcpyHostToDevice, st));
    const uint8_t *sel = sel_or_null ? d_sel.as<uint8_t>() : nullptr;
    // first launch measures (no text buffer), second prints into a buffer of exactly that size
    int64_t total = 0;
    rc = yolo_launch(d_box.as<double>(), d_off.as<int32_t>(), sel, d_w.as<double>(), d_h.as<double>(),
                     d_cid.as<int32_t>(), n_rows, d_toff.as<int64_t>(), d_flag.as<uint8_t>(), nullptr, 0, &total, st);
    if (rc) return rc;
    uint8_t *host_text = static_cast<uint8_t *>(malloc((size_t)(total > 0 ? total : 1)));
    if (!host_text) {
        set_error("malloc(%lld) failed", (long long)total);
        return DYD_ERR_OOM;
    }
    if (total > 0) {
        if ((rc = d_text.alloc((size_t)total))) {
            free(host_text);
            return rc;
        }
        KernelTimer t(st);
        rc = yolo_launch(d_box.as<double>(), d_off.as<int32_t>(), sel, d_w.as<double>(), d_h.as<double>(),
                         d_cid.as<int32_t>(), n_rows, d_toff.as<int64_t>(), d_flag.as<uint8_t>(),
                         d_text.as<uint8_t>(), total, &total, st);
        if (rc) {
            free(host_text);
            return rc;
        }
        t.finish();
        hipError_t e = hipMemcpyAsync(host_text, d_text.p, (size_t)total, hipMemcpyDeviceToHost, st);
        if (e != hipSuccess) {
            free(host_text);
            set_error("hipMemcpyAsync failed: %s", hipGetErrorString(e));
            return DYD_ERR_HIP;
        }
    }
    hipError_t e = hipMemcpyAsync(out_text_off, d_toff.p, 8 * (size_t)(n_rows + 1), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipMemcpyAsync(out_flag, d_flag.p, (size_t)n_rows, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) {
        free(host_text);
        set_error("copy back failed: %s", hipGetErrorString(e));
        return DYD_ERR_HIP;
    }
    *out_text = host_text;
    *out_text_len = total;
    return DYD_OK;
}

}  // extern "C"
