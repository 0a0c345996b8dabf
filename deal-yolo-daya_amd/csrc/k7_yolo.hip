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
// One pass: a workgroup takes the next tile of 256 x RPT rows from a ticket counter, measures its rows (RPT rows
// per lane, the converted numbers of each row's first line stay in registers), scans the lengths in the workgroup, obtains the byte offset of the tile with a decoupled
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
    uint64_t ip;      // integer digits of round-half-even(|v| * 10^6) / 10^6, finite values below 2^43
    uint32_t fp;      // the six decimals, 0..999999
    uint32_t meta;    // kind | neg << 2 (sign bit of v: printed also for -0.0 and values that round to zero)
    __device__ __forceinline__ uint32_t kind() const { return meta & 3u; }
    __device__ __forceinline__ uint32_t neg() const { return meta >> 2; }
};

// round-half-even(a * 10^6) for 0 <= a < 4294, exactly: a * 10^6 = t + e with t the rounded product and e the
// error term an FMA returns exactly; t = n + f (n integer, f exact).  f != 1/2 is at least ulp(t) >= 2|e| away
// from one half, so it decides alone; at f == 1/2 the sign of e decides and e == 0 is a true tie.
__device__ __forceinline__ uint32_t round6(double a) {
    const double t = a * 1.0e6;
    const double e = fma(a, 1.0e6, -t);
    const uint32_t n = (uint32_t)t;
    const double f = t - (double)n;
    const bool up = (f > 0.5) || (f == 0.5 && (e > 0.0 || (e == 0.0 && (n & 1u))));
    return n + (up ? 1u : 0u);
}

// exact |v| * 10^6 rounded half-to-even, split into integer part and six decimals
__device__ __forceinline__ Num6 classify(double v) {
    const uint64_t bits = (uint64_t)__double_as_longlong(v);
    Num6 r;
    const uint32_t neg = (uint32_t)(bits >> 63);
    r.ip = 0;
    r.fp = 0;
    const uint32_t ex = (uint32_t)(bits >> 52) & 0x7ffu;
    const uint64_t frac = bits & ((1ull << 52) - 1);
    if (ex == 0x7ffu) {
        r.meta = (frac ? K7_NAN : K7_INF) | (neg << 2);
        return r;
    }
    if (ex >= 1023u + 43u) {  // |v| >= 2^43: up to 309 integer digits, printed by the host
        r.meta = K7_EXOTIC;
        return r;
    }
    r.meta = K7_FINITE | (neg << 2);
    const double a = __longlong_as_double((long long)(bits & ~(1ull << 63)));
    if (a < 4294.0) {   // |v| * 10^6 < 2^32
        const uint32_t q32 = round6(a), i32 = q32 / 1000000u;
        r.ip = i32;
        r.fp = q32 - i32 * 1000000u;
        return r;
    }
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
    q += up ? 1 : 0;
    if ((q >> 32) == 0) {     // |v| < 4294.97: one 32-bit division
        const uint32_t q32 = (uint32_t)q, i32 = q32 / 1000000u;
        r.ip = i32;
        r.fp = q32 - i32 * 1000000u;
    } else {
        r.ip = q / 1000000ull;
        r.fp = (uint32_t)(q - r.ip * 1000000ull);
    }
    return r;
}

__device__ __forceinline__ int digits_u32(uint32_t w) {  // w < 10^7
    int n = 1;
    if (w >= 10000u) { w /= 10000u; n += 4; }
    if (w >= 100u) { w /= 100u; n += 2; }
    if (w >= 10u) n += 1;
    return n;
}

__device__ __forceinline__ int digits_ip(uint64_t v) {  // v < 10^14
    if (v < 10000000ull) return digits_u32((uint32_t)v);
    return 7 + digits_u32((uint32_t)(v / 10000000ull));
}

__device__ __forceinline__ int num_len(const Num6 &n) {
    if (n.kind() == K7_NAN) return 3;
    if (n.kind() == K7_INF) return 3 + (int)n.neg();
    return (int)n.neg() + digits_ip(n.ip) + 7;
}

// decimal digits of w, most significant first, exactly nd of them (zero padded)
template <class Put>
__device__ __forceinline__ void put_u32(uint32_t w, int nd, int pos, Put put) {
    for (int k = nd - 1; k >= 0; --k) {
        const uint32_t t = w / 10u;
        put(pos + k, (char)('0' + (int)(w - t * 10u)));
        w = t;
    }
}

template <class Put>
__device__ __forceinline__ int num_put(const Num6 &n, int pos, Put put) {
    if (n.kind() == K7_NAN) {
        put(pos, 'n'); put(pos + 1, 'a'); put(pos + 2, 'n');
        return pos + 3;
    }
    if (n.neg()) put(pos++, '-');
    if (n.kind() == K7_INF) {
        put(pos, 'i'); put(pos + 1, 'n'); put(pos + 2, 'f');
        return pos + 3;
    }
    if (n.ip < 10000000ull) {
        const int nd = digits_u32((uint32_t)n.ip);
        put_u32((uint32_t)n.ip, nd, pos, put);
        pos += nd;
    } else {
        const uint32_t top = (uint32_t)(n.ip / 10000000ull);
        const uint32_t low = (uint32_t)(n.ip - (uint64_t)top * 10000000ull);
        const int nd = digits_u32(top);
        put_u32(top, nd, pos, put);
        put_u32(low, 7, pos + nd, put);
        pos += nd + 7;
    }
    put(pos, '.');
    uint32_t fp = n.fp;
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
    const double2 lo2 = *reinterpret_cast<const double2 *>(b), hi2 = *reinterpret_cast<const double2 *>(b + 2);
    const double ax = lo2.x, ay = lo2.y, bx = hi2.x, by = hi2.y;
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
    l.exotic = l.v[0].kind() == K7_EXOTIC || l.v[1].kind() == K7_EXOTIC || l.v[2].kind() == K7_EXOTIC ||
               l.v[3].kind() == K7_EXOTIC;
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

__device__ __forceinline__ int line_len(const Line &l, int cd) {
    return cd + 4 + num_len(l.v[0]) + num_len(l.v[1]) + num_len(l.v[2]) + num_len(l.v[3]);
}

template <class Put>
__device__ __forceinline__ int line_put(const Line &l, uint32_t cid, int cd, int pos, Put put) {
    put_u32(cid, cd, pos, put);
    pos += cd;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        put(pos++, ' ');
        pos = num_put(l.v[k], pos, put);
    }
    return pos;
}

struct RowIn {
    int32_t b0, b1;
    double w, h;
    int32_t cid;
    bool host;   // zero width / height or negative class id: the host decides
};

struct RowState {   // what the measuring pass keeps for the printing pass
    uint32_t q[4];  // plain rows: the four values as round(v * 10^6) < 10^7
    uint32_t len;   // bytes of the row's text
    uint32_t flag;  // 0 text, 1 no line, 2 host
};

// A PLAIN row is the everyday one: a single box that gives a line whose four values are finite and print with
// one integer digit ("d.dddddd" or "-d.dddddd"); its line is cd + 36 bytes plus the minus signs.  Lanes holding a
// plain row keep the four rounded values (bit 31 = sign) and print them straight; the other lanes go through
// row_measure / row_print below.
// one box of a row of width w, height h: true when its line is plain; q = the four values as round(|v| * 10^6), bit 31 = sign
__device__ __forceinline__ bool plain_corners(double ax, double ay, double bx, double by, double w, double h, uint32_t q[4]);
__device__ __forceinline__ bool plain_box(const double *__restrict__ b, double w, double h, uint32_t q[4]) {
    const double2 lo2 = *reinterpret_cast<const double2 *>(b), hi2 = *reinterpret_cast<const double2 *>(b + 2);
    return plain_corners(lo2.x, lo2.y, hi2.x, hi2.y, w, h, q);
}
__device__ __forceinline__ bool plain_corners(double ax, double ay, double bx, double by, double w, double h, uint32_t q[4]) {
    const double x1 = (bx < ax) ? bx : ax, x2 = (bx > ax) ? bx : ax;
    const double y1 = (by < ay) ? by : ay, y2 = (by > ay) ? by : ay;
    const double bw = x2 - x1, bh = y2 - y1;            // max(d, 0.0) == d for the d > 0 accepted here
    bool ok = (bw > 0.0) && (bh > 0.0);
    const double v[4] = {(x1 + x2) / 2.0 / w, (y1 + y2) / 2.0 / h, bw / w, bh / h};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const double a = fabs(v[k]);
        const bool in_range = a < 9.9999994;             // false for NaN
        ok = ok && in_range;
        q[k] = round6(in_range ? a : 0.0) | ((uint32_t)((uint64_t)__double_as_longlong(v[k]) >> 63) << 31);
    }
    return ok;
}

__device__ __forceinline__ bool plain_row(const RowIn &r, const double *__restrict__ box4,
                                          const uint8_t *__restrict__ sel, uint32_t q[4]) {
    q[0] = q[1] = q[2] = q[3] = 0;
    if (r.host || r.b1 - r.b0 != 1 || (uint32_t)r.cid >= 100u) return false;
    if (sel && !sel[r.b0]) return false;
    return plain_box(box4 + 4 * (int64_t)r.b0, r.w, r.h, q);
}

// "d.dddddd" for q < 10^7
template <class Put>
__device__ __forceinline__ void put_plain(uint32_t q, int pos, Put put) {
    const uint32_t d0 = q / 1000000u, fp = q - d0 * 1000000u;
    const uint32_t ab = fp / 10000u, rest = fp - ab * 10000u;
    const uint32_t cd = rest / 100u, ef = rest - cd * 100u;
    const uint32_t a = (ab * 103u) >> 10, c = (cd * 103u) >> 10, e = (ef * 103u) >> 10;   // tens digit of a pair < 100
    put(pos, (char)('0' + d0));
    put(pos + 1, '.');
    put(pos + 2, (char)('0' + a));
    put(pos + 3, (char)('0' + (ab - a * 10u)));
    put(pos + 4, (char)('0' + c));
    put(pos + 5, (char)('0' + (cd - c * 10u)));
    put(pos + 6, (char)('0' + e));
    put(pos + 7, (char)('0' + (ef - e * 10u)));
}

__device__ __forceinline__ uint32_t plain_len(const RowIn &r, const uint32_t q[4]) {
    return ((uint32_t)r.cid >= 10u ? 38u : 37u) + (q[0] >> 31) + (q[1] >> 31) + (q[2] >> 31) + (q[3] >> 31);
}

template <class Put>
__device__ __forceinline__ void plain_print(const RowIn &r, const RowState &st, Put put) {
    const uint32_t cid = (uint32_t)r.cid, tens = (cid * 103u) >> 10;
    int pos = 0;
    if (cid >= 10u) put(pos++, (char)('0' + tens));
    put(pos++, (char)('0' + (cid - tens * 10u)));
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        put(pos++, ' ');
        if (st.q[k] >> 31) put(pos++, '-');
        put_plain(st.q[k] & 0x7fffffffu, pos, put);
        pos += 8;
    }
}

__device__ __forceinline__ void row_measure(const RowIn &r, const double *__restrict__ box4,
                                            const uint8_t *__restrict__ sel, RowState &st) {
    st.len = 0;
    if (r.host) {
        st.flag = 2;
        return;
    }
    uint32_t len = 0, lines = 0;
    const int cd = cid_digits((uint32_t)r.cid);
    for (int32_t b = r.b0; b < r.b1; ++b) {
        if (sel && !sel[b]) continue;
        const Line l = box_line(box4 + 4 * (int64_t)b, r.w, r.h);
        if (!l.valid) continue;
        if (l.exotic) {
            st.flag = 2;
            return;
        }
        len += (uint32_t)line_len(l, cd);
        ++lines;
    }
    st.flag = lines ? 0 : 1;
    st.len = lines ? len + lines - 1 : 0;
}

template <class Put>
__device__ __forceinline__ void row_print(const RowIn &r, const double *__restrict__ box4,
                                          const uint8_t *__restrict__ sel, Put put) {
    const int cd = cid_digits((uint32_t)r.cid);
    int pos = 0;
    for (int32_t b = r.b0; b < r.b1; ++b) {   // the lines are converted again: this path is the rare one
        if (sel && !sel[b]) continue;
        const Line l = box_line(box4 + 4 * (int64_t)b, r.w, r.h);
        if (!l.valid) continue;
        if (pos) put(pos++, '\n');
        pos = line_put(l, (uint32_t)r.cid, cd, pos, put);
    }
}

// state[0] = ticket counter, state[1] = error word, state[2 + t] = look-back word of tile t
template <int RPT>
__global__ __launch_bounds__(K7_BLOCK) void k7_yolo_kernel(const double *__restrict__ box4,
                                                           const int32_t *__restrict__ row_off,
                                                           const uint8_t *__restrict__ sel,
                                                           const double *__restrict__ width,
                                                           const double *__restrict__ height,
                                                           const int32_t *__restrict__ class_id, int64_t n_rows,
                                                           int64_t *__restrict__ text_off,
                                                           uint8_t *__restrict__ flag_out, uint8_t *text,
                                                           int64_t text_cap, unsigned long long *state,
                                                           unsigned long long *trace) {
    constexpr int TILE = K7_BLOCK * RPT;
    __shared__ __attribute__((aligned(16))) unsigned char s_text[K7_LDS_TEXT + 32];
    __shared__ uint32_t s_wave[RPT][K7_WAVES];
    __shared__ unsigned long long s_bcast[2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // Tiles are taken in ticket order: every tile before this one is then held by a running or finished workgroup
    // whatever order the hardware dispatches workgroups in, so waiting for their words below cannot deadlock.
    // (One counter word sustains ~88 tickets per microsecond; 512-row tiles need ~60.)
    if (tid == 0) s_bcast[0] = atomicAdd(&state[0], 1ull);
    __syncthreads();
    const int64_t tile = (int64_t)s_bcast[0];
    unsigned long long *words = state + 2;
    const int64_t n_tiles = (n_rows + TILE - 1) / TILE;
    if (tile >= n_tiles) return;
    const int64_t row0 = tile * TILE + tid;   // the thread's rows are row0 + k * K7_BLOCK
#define K7_STAMP(i) do { if (trace && tid == 0) trace[tile * 8 + (i)] = wall_clock64(); } while (0)
    K7_STAMP(0);

    // ---- measure --------------------------------------------------------------------------------
    RowIn r[RPT];
    RowState st[RPT];
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
        const int64_t row = row0 + (int64_t)k * K7_BLOCK;
        r[k].b0 = r[k].b1 = 0;
        r[k].w = r[k].h = 1.0;
        r[k].cid = 0;
        if (row < n_rows) {
            r[k].b0 = row_off[row];
            r[k].b1 = row_off[row + 1];
            r[k].w = width[row];
            r[k].h = height[row];
            r[k].cid = class_id[row];
        }
        r[k].host = (r[k].w == 0.0) || (r[k].h == 0.0) || (r[k].cid < 0);
    }
    K7_STAMP(1);
    bool plain[RPT];
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
        const bool live = row0 + (int64_t)k * K7_BLOCK < n_rows;
        plain[k] = live && plain_row(r[k], box4, sel, st[k].q);
        if (plain[k]) {
            st[k].len = plain_len(r[k], st[k].q);
            st[k].flag = 0;
        } else {
            row_measure(r[k], box4, sel, st[k]);
            if (!live) st[k].len = 0;
        }
    }
    if (trace && tid == 0) trace[tile * 8 + 7] = (unsigned long long)plain[0] | ((unsigned long long)plain[RPT - 1] << 1);
    K7_STAMP(2);

    // ---- exclusive scan of the lengths in row order (k major) ------------------------------------------
    uint32_t incl[RPT];
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
        incl[k] = st[k].len;
#pragma unroll
        for (int d = 1; d < kWave; d <<= 1) {
            const uint32_t up = __shfl_up(incl[k], d);
            if (lane >= d) incl[k] += up;
        }
        if (lane == kWave - 1) s_wave[k][wave] = incl[k];
    }
    __syncthreads();
    uint32_t toff[RPT], tile_bytes = 0;
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
        uint32_t before = tile_bytes;
#pragma unroll
        for (int w = 0; w < K7_WAVES; ++w) {
            if (w < wave) before += s_wave[k][w];
            tile_bytes += s_wave[k][w];
        }
        toff[k] = before + incl[k] - st[k].len;
    }

    K7_STAMP(3);
    // the tile's own byte count goes out first: the tiles after this one need only that to move on
    if (tid == 0) {
        const unsigned long long mine = (tile == 0 ? K7_FLAG_PFX : K7_FLAG_AGG) | (unsigned long long)tile_bytes;
        __hip_atomic_store(&words[tile], mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // ---- print into LDS at tile-relative offsets while the words of the earlier tiles arrive ----------
    const bool staged = text && tile_bytes && tile_bytes <= (uint32_t)K7_LDS_TEXT;
    if (staged) {
#pragma unroll
        for (int k = 0; k < RPT; ++k) {
            unsigned char *mine = s_text + toff[k];
            if (plain[k])
                plain_print(r[k], st[k], [&](int p, char c) { mine[p] = (unsigned char)c; });
            else if (st[k].len)
                row_print(r[k], box4, sel, [&](int p, char c) { mine[p] = (unsigned char)c; });
        }
    }
    K7_STAMP(4);
    // ---- decoupled look-back (wave 0) -------------------------------------------------------------
    if (wave == 0) {
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
    K7_STAMP(5);

#pragma unroll
    for (int k = 0; k < RPT; ++k) {
        const int64_t row = row0 + (int64_t)k * K7_BLOCK;
        if (row < n_rows) {
            text_off[row] = base + toff[k];
            flag_out[row] = (uint8_t)st[k].flag;
            if (row == n_rows - 1) text_off[n_rows] = base + toff[k] + st[k].len;
        }
    }
    if (!text || tile_bytes == 0) return;
    if (base + (int64_t)tile_bytes > text_cap) {   // the host sees the total in text_off[n_rows] and reports it
        if (tid == 0) atomicExch(&state[1], 2ull);
        return;
    }
    unsigned char *dst = text + base;
    if (staged) {
        // dst[i] = s_text[i].  The 16-byte store k goes to the aligned address dst - phase + 16k and takes the LDS
        // bytes from 16k - phase on: five aligned dwords funnel-shifted by (-phase) & 3 bytes.
        const uint32_t phase = (uint32_t)(reinterpret_cast<uintptr_t>(dst) & 15u);
        const uint32_t end = phase + tile_bytes;
        const uint32_t n_chunks = (end + 15u) >> 4;
        unsigned char *aligned = dst - phase;
        const uint32_t *s32 = reinterpret_cast<const uint32_t *>(s_text);
        const uint32_t sh = (0u - phase) & 3u;
        for (uint32_t k = tid; k < n_chunks; k += K7_BLOCK) {
            const uint32_t lo = k << 4, hi = lo + 16u;
            if (lo >= phase && hi <= end) {
                const uint32_t m = (lo - phase) >> 2;
                const uint32_t d0 = s32[m], d1 = s32[m + 1], d2 = s32[m + 2], d3 = s32[m + 3], d4 = s32[m + 4];
                uint4 v;
                v.x = __builtin_amdgcn_alignbyte(d1, d0, sh);
                v.y = __builtin_amdgcn_alignbyte(d2, d1, sh);
                v.z = __builtin_amdgcn_alignbyte(d3, d2, sh);
                v.w = __builtin_amdgcn_alignbyte(d4, d3, sh);
                *reinterpret_cast<uint4 *>(aligned + lo) = v;
            } else {
                const uint32_t a = lo < phase ? phase : lo, b = hi > end ? end : hi;
                for (uint32_t i = a; i < b; ++i) aligned[i] = s_text[i - phase];
            }
        }
        K7_STAMP(6);
    } else {   // a tile of very long rows: print straight to memory
#pragma unroll
        for (int k = 0; k < RPT; ++k) {
            unsigned char *mine = dst + toff[k];
            if (plain[k])
                plain_print(r[k], st[k], [&](int p, char c) { mine[p] = (unsigned char)c; });
            else if (st[k].len)
                row_print(r[k], box4, sel, [&](int p, char c) { mine[p] = (unsigned char)c; });
        }
    }
}

// ---- two tiles per ticket, software-pipelined -----------------------------------------------------------------
// The single-tile kernel above idles ~6 us per tile in its look-back: a byte count published by another XCD becomes
// visible two memory round trips later, and a tile has only its own printing (1.4 us) to do meanwhile.  Here a
// workgroup takes tiles 2T and 2T+1: it measures A, publishes A's count, prints A into LDS, then measures B and
// publishes B's count, and only then looks back for A — a whole tile's measuring later, when the counts of the
// tiles before A have long arrived.  B needs no look-back: its base is A's base plus A's bytes.  A's per-row
// offsets wait in LDS (the registers are B's by then).  Deadlock-free for the same reason as above: both counts of a
// ticket are published without waiting for anything.
template <int RPT>
__global__ __launch_bounds__(K7_BLOCK) void k7_yolo_pair_kernel(const double *__restrict__ box4,
                                                                const int32_t *__restrict__ row_off,
                                                                const uint8_t *__restrict__ sel,
                                                                const double *__restrict__ width,
                                                                const double *__restrict__ height,
                                                                const int32_t *__restrict__ class_id, int64_t n_rows,
                                                                int64_t *__restrict__ text_off,
                                                                uint8_t *__restrict__ flag_out, uint8_t *text,
                                                                int64_t text_cap, unsigned long long *state) {
    constexpr int TILE = K7_BLOCK * RPT;
    constexpr int LDS_TEXT = K7_LDS_TEXT * RPT / 2;   // 48 bytes of staging per row
    __shared__ __attribute__((aligned(16))) unsigned char s_text[LDS_TEXT + 32];
    __shared__ uint32_t s_wave[RPT][K7_WAVES];
    __shared__ uint32_t s_park_off[RPT][K7_BLOCK], s_park_meta[RPT][K7_BLOCK];   // tile A: offset, len << 2 | flag
    __shared__ unsigned long long s_bcast[2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) s_bcast[0] = atomicAdd(&state[0], 1ull);
    __syncthreads();
    const int64_t ticket = (int64_t)s_bcast[0];
    unsigned long long *words = state + 2;
    const int64_t n_tiles = (n_rows + TILE - 1) / TILE;
    const int64_t tile_a = 2 * ticket, tile_b = tile_a + 1;
    if (tile_a >= n_tiles) return;
    const bool have_b = tile_b < n_tiles;

    RowIn r[RPT];
    RowState st[RPT];
    bool plain[RPT];
    uint32_t toff[RPT];

    // loads + lengths + workgroup scan of one tile; leaves the tile's state in r / st / plain / toff
    auto measure = [&](int64_t tile) -> uint32_t {
        const int64_t row0 = tile * TILE + tid;
#pragma unroll
        for (int k = 0; k < RPT; ++k) {
            const int64_t row = row0 + (int64_t)k * K7_BLOCK;
            r[k].b0 = r[k].b1 = 0;
            r[k].w = r[k].h = 1.0;
            r[k].cid = 0;
            if (row < n_rows) {
                r[k].b0 = row_off[row];
                r[k].b1 = row_off[row + 1];
                r[k].w = width[row];
                r[k].h = height[row];
                r[k].cid = class_id[row];
            }
            r[k].host = (r[k].w == 0.0) || (r[k].h == 0.0) || (r[k].cid < 0);
        }
#pragma unroll
        for (int k = 0; k < RPT; ++k) {
            const bool live = row0 + (int64_t)k * K7_BLOCK < n_rows;
            plain[k] = live && plain_row(r[k], box4, sel, st[k].q);
            if (plain[k]) {
                st[k].len = plain_len(r[k], st[k].q);
                st[k].flag = 0;
            } else {
                row_measure(r[k], box4, sel, st[k]);
                if (!live) st[k].len = 0;
            }
        }
        uint32_t incl[RPT];
        __syncthreads();   // s_wave of the previous tile has been read by everyone
#pragma unroll
        for (int k = 0; k < RPT; ++k) {
            incl[k] = st[k].len;
#pragma unroll
            for (int d = 1; d < kWave; d <<= 1) {
                const uint32_t up = __shfl_up(incl[k], d);
                if (lane >= d) incl[k] += up;
            }
            if (lane == kWave - 1) s_wave[k][wave] = incl[k];
        }
        __syncthreads();
        uint32_t bytes = 0;
#pragma unroll
        for (int k = 0; k < RPT; ++k) {
            uint32_t before = bytes;
#pragma unroll
            for (int w = 0; w < K7_WAVES; ++w) {
                if (w < wave) before += s_wave[k][w];
                bytes += s_wave[k][w];
            }
            toff[k] = before + incl[k] - st[k].len;
        }
        return bytes;
    };
    auto print_rows = [&](unsigned char *where) {   // the current tile's rows at where + toff (LDS or memory)
#pragma unroll
        for (int k = 0; k < RPT; ++k) {
            unsigned char *mine = where + toff[k];
            if (plain[k])
                plain_print(r[k], st[k], [&](int p, char c) { mine[p] = (unsigned char)c; });
            else if (st[k].len)
                row_print(r[k], box4, sel, [&](int p, char c) { mine[p] = (unsigned char)c; });
        }
    };
    auto flush = [&](unsigned char *dst, uint32_t bytes) {   // dst[i] = s_text[i], 16-byte stores, see k7_yolo_kernel
        const uint32_t phase = (uint32_t)(reinterpret_cast<uintptr_t>(dst) & 15u);
        const uint32_t end = phase + bytes;
        const uint32_t n_chunks = (end + 15u) >> 4;
        unsigned char *aligned = dst - phase;
        const uint32_t *s32 = reinterpret_cast<const uint32_t *>(s_text);
        const uint32_t sh = (0u - phase) & 3u;
        for (uint32_t c = tid; c < n_chunks; c += K7_BLOCK) {
            const uint32_t lo = c << 4, hi = lo + 16u;
            if (lo >= phase && hi <= end) {
                const uint32_t m = (lo - phase) >> 2;
                const uint32_t d0 = s32[m], d1 = s32[m + 1], d2 = s32[m + 2], d3 = s32[m + 3], d4 = s32[m + 4];
                uint4 v;
                v.x = __builtin_amdgcn_alignbyte(d1, d0, sh);
                v.y = __builtin_amdgcn_alignbyte(d2, d1, sh);
                v.z = __builtin_amdgcn_alignbyte(d3, d2, sh);
                v.w = __builtin_amdgcn_alignbyte(d4, d3, sh);
                *reinterpret_cast<uint4 *>(aligned + lo) = v;
            } else {
                const uint32_t a = lo < phase ? phase : lo, b = hi > end ? end : hi;
                for (uint32_t i = a; i < b; ++i) aligned[i] = s_text[i - phase];
            }
        }
    };
    auto offsets_from_regs = [&](int64_t tile, int64_t base) {
        const int64_t row0 = tile * TILE + tid;
#pragma unroll
        for (int k = 0; k < RPT; ++k) {
            const int64_t row = row0 + (int64_t)k * K7_BLOCK;
            if (row < n_rows) {
                text_off[row] = base + toff[k];
                flag_out[row] = (uint8_t)st[k].flag;
                if (row == n_rows - 1) text_off[n_rows] = base + toff[k] + st[k].len;
            }
        }
    };

    // ---- tile A: measure, publish, print, park ---------------------------------------------------------------------
    const uint32_t bytes_a = measure(tile_a);
    if (tid == 0)
        __hip_atomic_store(&words[tile_a], (tile_a == 0 ? K7_FLAG_PFX : K7_FLAG_AGG) | (unsigned long long)bytes_a, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
    const bool staged_a = text && bytes_a && bytes_a <= (uint32_t)LDS_TEXT;
    const bool pipelined = have_b && (staged_a || !text || bytes_a == 0);   // a tile too long for LDS is finished first
    if (staged_a) print_rows(s_text);
    if (pipelined) {
#pragma unroll
        for (int k = 0; k < RPT; ++k) {
            s_park_off[k][tid] = toff[k];
            s_park_meta[k][tid] = (st[k].len << 2) | st[k].flag;
        }
    }
    // ---- tile B measured before A is looked back for ------------------------------------------------------------
    uint32_t bytes_b = 0;
    if (pipelined) {
        bytes_b = measure(tile_b);   // (its barriers also order A's printing before A's flush)
        if (tid == 0)
            __hip_atomic_store(&words[tile_b], K7_FLAG_AGG | (unsigned long long)bytes_b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // ---- look-back for A (wave 0) -----------------------------------------------------------------------------------
    if (wave == 0) {
        unsigned long long base = 0;
        int64_t look = tile_a - 1;
        bool failed = false;
        while (look >= 0) {
            const int64_t t = look - lane;
            unsigned long long wv = K7_FLAG_PFX;
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
            const int first = has_pfx ? __ffsll((long long)has_pfx) - 1 : kWave;
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
            if (tile_a != 0)
                __hip_atomic_store(&words[tile_a], K7_FLAG_PFX | ((base + bytes_a) & K7_VALUE), __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
            if (pipelined)   // B's prefix is known without a look-back of its own
                __hip_atomic_store(&words[tile_b], K7_FLAG_PFX | ((base + bytes_a + bytes_b) & K7_VALUE), __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
            s_bcast[1] = base;
        }
    }
    __syncthreads();
    const int64_t base_a = (int64_t)s_bcast[1];
    const bool fits_a = base_a + (int64_t)bytes_a <= text_cap;
    if (text && bytes_a && !fits_a && tid == 0) atomicExch(&state[1], 2ull);

    if (pipelined) {
        // ---- finish A from the parked offsets -----------------------------------------------------------------------
        const int64_t row0 = tile_a * TILE + tid;
#pragma unroll
        for (int k = 0; k < RPT; ++k) {
            const int64_t row = row0 + (int64_t)k * K7_BLOCK;   // (tile A is never the last tile here)
            if (row < n_rows) {
                text_off[row] = base_a + s_park_off[k][tid];
                flag_out[row] = (uint8_t)(s_park_meta[k][tid] & 3u);
            }
        }
        if (staged_a && fits_a) flush(text + base_a, bytes_a);
        // ---- finish B: its state is still in registers --------------------------------------------------------------
        const int64_t base_b = base_a + (int64_t)bytes_a;
        offsets_from_regs(tile_b, base_b);
        if (!text || bytes_b == 0) return;
        if (base_b + (int64_t)bytes_b > text_cap) {
            if (tid == 0) atomicExch(&state[1], 2ull);
            return;
        }
        if (bytes_b <= (uint32_t)LDS_TEXT) {
            __syncthreads();                 // A has left LDS
            print_rows(s_text);
            __syncthreads();
            flush(text + base_b, bytes_b);
        } else {
            print_rows(text + base_b);
        }
        return;
    }
    // ---- not pipelined: A alone (last tile of the grid, or a tile too long for LDS), then B the same way -----------
    offsets_from_regs(tile_a, base_a);
    if (text && bytes_a && fits_a) {
        if (staged_a) flush(text + base_a, bytes_a);
        else print_rows(text + base_a);
    }
    if (!have_b) return;
    const uint32_t bytes_b2 = measure(tile_b);
    const int64_t base_b = base_a + (int64_t)bytes_a;
    if (tid == 0)
        __hip_atomic_store(&words[tile_b], K7_FLAG_PFX | ((unsigned long long)(base_b + bytes_b2) & K7_VALUE), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
    offsets_from_regs(tile_b, base_b);
    if (!text || bytes_b2 == 0) return;
    if (base_b + (int64_t)bytes_b2 > text_cap) {
        if (tid == 0) atomicExch(&state[1], 2ull);
        return;
    }
    if (bytes_b2 <= (uint32_t)LDS_TEXT) {
        print_rows(s_text);
        __syncthreads();
        flush(text + base_b, bytes_b2);
    } else {
        print_rows(text + base_b);
    }
}

// ---- rows of many boxes: tiles cut by BOXES, one lane per box ----------------------------------------------------
// The row-tiled kernels give a lane a whole row; with 1..32 boxes per row a 512-row tile's text (300 KB) no longer fits
// LDS, the lanes loop over their boxes one after the other and print byte by byte to memory: 332 ms for 16.5 M lines
// (measured).  Here a tile is the set of rows whose FIRST box falls into a window of K7B_WINDOW consecutive boxes
// (two binary searches in row_off; every row, empty ones included, belongs to exactly one tile), a lane owns a box,
// and what a row needs from its boxes comes out of workgroup scans over the boxes of the tile: the number of lines
// before a box (the first line of a row has no "\n" in front, the others do) and the bytes before it.  One line the
// host must print (a value of 2^43 or more) makes the whole row the host's, so when a tile has such a line a third scan
// tells every box whether its row holds one.  Tiles of more than K7B_CAP boxes (a row of hundreds of boxes) take
// the row-per-lane route in chunks of 256 rows.
constexpr int K7B_WINDOW = 480;
constexpr int K7B_CAP = 512;                 // boxes the LDS arrays of a tile hold: two per lane
constexpr int K7B_PER = K7B_CAP / K7_BLOCK;
constexpr int K7B_LDS_TEXT = 22 * 1024;      // text staged per tile (480..511 lines of ~38 B); with the row table 5 workgroups fit a CU

// first i in [0, n] with off[i] >= x
__device__ __forceinline__ int64_t lower_bound_off(const int32_t *__restrict__ off, int64_t n, int64_t x) {
    int64_t lo = 0, hi = n;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (off[mid] < x) lo = mid + 1; else hi = mid;
    }
    return lo;
}

__device__ __forceinline__ RowIn k7_row(int64_t r, const int32_t *__restrict__ row_off, const double *__restrict__ width,
                                        const double *__restrict__ height, const int32_t *__restrict__ class_id) {
    RowIn ri;
    ri.b0 = row_off[r];
    ri.b1 = row_off[r + 1];
    ri.w = width[r];
    ri.h = height[r];
    ri.cid = class_id[r];
    ri.host = (ri.w == 0.0) || (ri.h == 0.0) || (ri.cid < 0);
    return ri;
}

// A whole WAVE measures / prints one row (the box kernel's tiles of more than K7B_CAP boxes: label files of hundreds of
// lines): the lanes take the row's boxes 64 at a time, lengths are summed, and while printing an exclusive wave scan of
// the line costs (length + the "\n" in front; the row's first line writes none) gives every line its place.
__device__ __forceinline__ void wave_row_measure(const RowIn &r, const double *__restrict__ box4, const uint8_t *__restrict__ sel,
                                                 int lane, uint32_t &len_out, uint32_t &flag_out) {   // wave-uniform results
    if (r.host) {
        len_out = 0;
        flag_out = 2;
        return;
    }
    uint32_t len = 0, lines = 0;
    bool exotic = false;
    const int cd = cid_digits((uint32_t)r.cid);
    for (int32_t b = r.b0 + lane; b < r.b1; b += kWave) {
        if (sel && !sel[b]) continue;
        const Line l = box_line(box4 + 4 * (int64_t)b, r.w, r.h);
        if (!l.valid) continue;
        if (l.exotic) exotic = true;
        else { len += (uint32_t)line_len(l, cd); ++lines; }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        len += __shfl_xor(len, d);
        lines += __shfl_xor(lines, d);
    }
    if (__any(exotic)) {
        len_out = 0;
        flag_out = 2;
        return;
    }
    flag_out = lines ? 0 : 1;
    len_out = lines ? len + lines - 1 : 0;
}

__device__ __forceinline__ void wave_row_print(const RowIn &r, const double *__restrict__ box4, const uint8_t *__restrict__ sel,
                                               int lane, unsigned char *dst) {
    const int cd = cid_digits((uint32_t)r.cid);
    uint32_t done = 0;   // cost of the lines printed so far (wave-uniform)
    for (int32_t b0 = r.b0; b0 < r.b1; b0 += kWave) {
        const int32_t b = b0 + lane;
        Line l;
        l.valid = false;
        if (b < r.b1 && (!sel || sel[b])) l = box_line(box4 + 4 * (int64_t)b, r.w, r.h);
        const uint32_t cost = l.valid ? (uint32_t)line_len(l, cd) + 1u : 0u;
        uint32_t incl = cost;
#pragma unroll
        for (int d = 1; d < kWave; d <<= 1) {
            const uint32_t up = __shfl_up(incl, d);
            if (lane >= d) incl += up;
        }
        const uint32_t total = __shfl(incl, kWave - 1);
        if (l.valid) {
            const uint32_t pos = done + incl - cost;   // where this line's "\n" would sit; its text follows
            if (pos) dst[pos - 1] = '\n';
            unsigned char *mine = dst + pos;   // (for the row's first line pos is 0: no "\n", the text starts the row)
            line_put(l, (uint32_t)r.cid, cd, 0, [&](int p, char c) { mine[p] = (unsigned char)c; });
        }
        done += total;
    }
}

// tile t owns the rows [tile_row[t], tile_row[t + 1]): one binary search per tile, all tiles at once (inside the tile
// kernel the 20 dependent loads of a search cost more than the rest of the tile)
__global__ __launch_bounds__(K7_BLOCK) void k7_tile_rows_kernel(const int32_t *__restrict__ row_off, int64_t n_rows, int64_t n_tiles,
                                                                int64_t *__restrict__ tile_row, int64_t *__restrict__ tile_box) {
    const int64_t t = (int64_t)blockIdx.x * K7_BLOCK + threadIdx.x;
    if (t > n_tiles) return;
    const int64_t r = (t == n_tiles) ? n_rows : lower_bound_off(row_off, n_rows, t * K7B_WINDOW);
    tile_row[t] = r;
    tile_box[t] = row_off[r];   // the tile kernel then needs no dependent load to know its boxes
}

__global__ __launch_bounds__(K7_BLOCK) void k7_yolo_box_kernel(const double *__restrict__ box4,
                                                               const int32_t *__restrict__ row_off,
                                                               const uint8_t *__restrict__ sel,
                                                               const double *__restrict__ width,
                                                               const double *__restrict__ height,
                                                               const int32_t *__restrict__ class_id, int64_t n_rows,
                                                               int64_t n_tiles, const int64_t *__restrict__ tile_row,
                                                               const int64_t *__restrict__ tile_box,
                                                               int64_t *__restrict__ text_off,
                                                               uint8_t *__restrict__ flag_out, uint8_t *text,
                                                               int64_t text_cap, unsigned long long *state,
                                                               unsigned long long *trace) {
    __shared__ __attribute__((aligned(16))) unsigned char s_text[K7B_LDS_TEXT + 32];
    __shared__ uint32_t s_cnt[K7B_CAP + 1], s_pos[K7B_CAP + 1], s_ex[K7B_CAP + 1];   // exclusive scans over the boxes: lines / bytes / host lines
    __shared__ uint32_t s_off[K7B_CAP + 1];     // the tile's row offsets, relative to its first box (when it has at most K7B_CAP rows)
    __shared__ uint8_t s_host[K7B_CAP];         // row is the host's by its sizes / class id
    __shared__ uint32_t s_wave[K7B_PER][K7_WAVES];
    __shared__ unsigned long long s_bcast[2];
    __shared__ uint32_t s_bad;           // the tile holds a line only the host can print
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) s_bcast[0] = atomicAdd(&state[0], 1ull);   // ticket order, as in the row kernels
    __syncthreads();
    const int64_t tile = (int64_t)s_bcast[0];
    unsigned long long *words = state + 2;
    if (tile >= n_tiles) return;
    if (tid == 0) s_bad = 0;
#define K7B_STAMP(i) do { if (trace && tid == 0) trace[tile * 8 + (i)] = wall_clock64(); } while (0)
    K7B_STAMP(0);
    const int64_t r_lo = tile_row[tile], r_hi = tile_row[tile + 1];           // rows [r_lo, r_hi), possibly none
    const int64_t b_lo = tile_box[tile], nb = tile_box[tile + 1] - b_lo;
    const int64_t nr = r_hi - r_lo;
    const bool small = nb <= K7B_CAP;

    // exclusive workgroup scan in box order j = k * K7_BLOCK + tid; out[j] for every j <= K7B_CAP, returns the total
    auto block_scan = [&](const uint32_t (&x)[K7B_PER], uint32_t *out) -> uint32_t {
        uint32_t incl[K7B_PER];
        __syncthreads();   // s_wave may still be read by the scan before
#pragma unroll
        for (int k = 0; k < K7B_PER; ++k) {
            incl[k] = x[k];
#pragma unroll
            for (int d = 1; d < kWave; d <<= 1) {
                const uint32_t up = __shfl_up(incl[k], d);
                if (lane >= d) incl[k] += up;
            }
            if (lane == kWave - 1) s_wave[k][wave] = incl[k];
        }
        __syncthreads();
        uint32_t total = 0;
#pragma unroll
        for (int k = 0; k < K7B_PER; ++k) {
            uint32_t before = total;
#pragma unroll
            for (int w = 0; w < K7_WAVES; ++w) {
                if (w < wave) before += s_wave[k][w];
                total += s_wave[k][w];
            }
            out[k * K7_BLOCK + tid] = before + incl[k] - x[k];
        }
        if (tid == 0) out[K7B_CAP] = total;
        __syncthreads();
        return total;
    };

    uint32_t tile_bytes = 0;
    // what a lane keeps of its boxes
    RowIn in[K7B_PER];          // the row of the box (b0 / b1 relative to b_lo)
    uint32_t q[K7B_PER][4], len[K7B_PER], line[K7B_PER];
    bool plain[K7B_PER];
    if (small) {
        // the lane's boxes are requested first; meanwhile the tile's rows (offsets relative to the tile, sizes, class ids)
        // go to LDS with coalesced loads (the text area is free until the printing), so that a box finds its row and
        // the row's numbers without another trip to memory
        double2 c_lo[K7B_PER], c_hi[K7B_PER];
#pragma unroll
        for (int k = 0; k < K7B_PER; ++k) {
            const int64_t j = (int64_t)k * K7_BLOCK + tid;
            c_lo[k] = c_hi[k] = make_double2(0.0, 0.0);
            if (j < nb) {
                const double2 *b = reinterpret_cast<const double2 *>(box4 + 4 * (b_lo + j));
                c_lo[k] = b[0];
                c_hi[k] = b[1];
            }
        }
        double *s_w = reinterpret_cast<double *>(s_text);                                    // [K7B_CAP]
        double *s_h = s_w + K7B_CAP;                                                         // [K7B_CAP]
        int32_t *s_cid = reinterpret_cast<int32_t *>(s_h + K7B_CAP);                         // [K7B_CAP]
        static_assert(20 * K7B_CAP <= K7B_LDS_TEXT, "the rows of a tile fit the text area");
        const bool rows_in_lds = nr <= K7B_CAP;
        if (rows_in_lds) {
            for (int64_t i = tid; i <= nr; i += K7_BLOCK) s_off[i] = (uint32_t)(row_off[r_lo + i] - b_lo);
            for (int64_t i = tid; i < nr; i += K7_BLOCK) {
                const double w = width[r_lo + i], h = height[r_lo + i];
                const int32_t c = class_id[r_lo + i];
                s_w[i] = w;
                s_h[i] = h;
                s_cid[i] = c;
                s_host[i] = (w == 0.0) || (h == 0.0) || (c < 0);
            }
        }
        __syncthreads();
        K7B_STAMP(1);
        uint32_t host_line[K7B_PER];
#pragma unroll
        for (int k = 0; k < K7B_PER; ++k) {
            const int64_t j = (int64_t)k * K7_BLOCK + tid;
            len[k] = line[k] = host_line[k] = 0;
            plain[k] = false;
            in[k].b0 = in[k].b1 = 0;
            in[k].w = in[k].h = 1.0;
            in[k].cid = 0;
            in[k].host = true;
            if (j < nb) {
                int64_t lo = 0, hi = nr;   // the last row of the tile that starts at or before box j
                if (rows_in_lds) {
                    while (hi - lo > 1) {
                        const int64_t mid = (lo + hi) >> 1;
                        if (s_off[mid] <= (uint32_t)j) lo = mid; else hi = mid;
                    }
                    in[k].b0 = (int32_t)s_off[lo];
                    in[k].b1 = (int32_t)s_off[lo + 1];
                    in[k].w = s_w[lo];
                    in[k].h = s_h[lo];
                    in[k].cid = s_cid[lo];
                } else {   // a run of empty rows longer than the table: searched and read in memory
                    while (hi - lo > 1) {
                        const int64_t mid = (lo + hi) >> 1;
                        if (row_off[r_lo + mid] - b_lo <= j) lo = mid; else hi = mid;
                    }
                    in[k].b0 = (int32_t)(row_off[r_lo + lo] - b_lo);
                    in[k].b1 = (int32_t)(row_off[r_lo + lo + 1] - b_lo);
                    in[k].w = width[r_lo + lo];
                    in[k].h = height[r_lo + lo];
                    in[k].cid = class_id[r_lo + lo];
                }
                in[k].host = (in[k].w == 0.0) || (in[k].h == 0.0) || (in[k].cid < 0);
                if (!in[k].host && (!sel || sel[b_lo + j])) {
                    if ((uint32_t)in[k].cid < 100u && plain_corners(c_lo[k].x, c_lo[k].y, c_hi[k].x, c_hi[k].y, in[k].w, in[k].h, q[k])) {
                        plain[k] = true;
                        line[k] = 1;
                        len[k] = plain_len(in[k], q[k]);
                    } else {
                        const Line l = box_line(box4 + 4 * (b_lo + j), in[k].w, in[k].h);
                        if (l.valid) {
                            line[k] = 1;
                            if (l.exotic) host_line[k] = 1;
                            else len[k] = (uint32_t)line_len(l, cid_digits((uint32_t)in[k].cid));
                        }
                    }
                }
            }
            if (host_line[k]) atomicOr(&s_bad, 1u);
        }
        K7B_STAMP(2);
        __syncthreads();
        const bool bad = s_bad != 0;
        if (bad) {   // rare: rows holding a host line give no text at all
            block_scan(host_line, s_ex);
#pragma unroll
            for (int k = 0; k < K7B_PER; ++k)
                if (s_ex[in[k].b1] > s_ex[in[k].b0]) len[k] = line[k] = 0;
        }
        block_scan(line, s_cnt);
#pragma unroll
        for (int k = 0; k < K7B_PER; ++k) {   // every line but the first of its row carries a "\n" in front
            const int64_t j = (int64_t)k * K7_BLOCK + tid;
            if (line[k] && s_cnt[j] > s_cnt[in[k].b0]) {
                len[k] += 1;
                line[k] = 2;
            }
        }
        tile_bytes = block_scan(len, s_pos);
    } else {
        // ---- more boxes than the LDS arrays hold: a WAVE per row (the rows here run to hundreds of boxes); offsets relative to the tile -----
        uint32_t carry = 0;
        for (int64_t c = 0; c < nr; c += K7_WAVES) {   // a wave per row, four rows at a time
            const int64_t r = r_lo + c + wave;
            uint32_t rlen = 0, rflag = 1;
            if (r < r_hi) {
                const RowIn ri = k7_row(r, row_off, width, height, class_id);
                wave_row_measure(ri, box4, sel, lane, rlen, rflag);
            }
            __syncthreads();
            if (lane == 0) s_wave[0][wave] = rlen;
            __syncthreads();
            uint32_t before = carry;
#pragma unroll
            for (int w = 0; w < K7_WAVES; ++w) {
                if (w < wave) before += s_wave[0][w];
                carry += s_wave[0][w];
            }
            if (r < r_hi && lane == 0) {
                text_off[r] = (int64_t)before;   // the tile's base is added below
                flag_out[r] = (uint8_t)rflag;
            }
        }
        tile_bytes = carry;
    }

    K7B_STAMP(3);
    // ---- publish; print into LDS while the tiles before publish theirs; look back (wave 0) ---------------------------------
    if (tid == 0)
        __hip_atomic_store(&words[tile], (tile == 0 ? K7_FLAG_PFX : K7_FLAG_AGG) | (unsigned long long)tile_bytes, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
    const bool staged = small && text && tile_bytes && tile_bytes <= (uint32_t)K7B_LDS_TEXT;
    auto print_box = [&](int k, int64_t j, unsigned char *where) {
        auto put = [&](int p, char c) { where[p] = (unsigned char)c; };
        int pos = 0;
        if (line[k] == 2) put(pos++, '\n');
        if (plain[k]) {
            RowState st;
            st.q[0] = q[k][0]; st.q[1] = q[k][1]; st.q[2] = q[k][2]; st.q[3] = q[k][3];
            plain_print(in[k], st, [&](int p, char c) { where[pos + p] = (unsigned char)c; });
        } else {
            const Line l = box_line(box4 + 4 * (b_lo + j), in[k].w, in[k].h);   // converted again: the rare kind of line
            line_put(l, (uint32_t)in[k].cid, cid_digits((uint32_t)in[k].cid), pos, put);
        }
    };
    if (staged) {   // (the row offsets parked in s_text are not needed any more: every lane holds b0 / b1 of its boxes)
#pragma unroll
        for (int k = 0; k < K7B_PER; ++k) {
            const int64_t j = (int64_t)k * K7_BLOCK + tid;
            if (small && line[k]) print_box(k, j, s_text + s_pos[j]);
        }
    }
    K7B_STAMP(4);
    if (wave == 0) {
        unsigned long long base = 0;
        int64_t look = tile - 1;
        bool failed = false;
        while (look >= 0) {
            const int64_t t = look - lane;
            unsigned long long wv = K7_FLAG_PFX;
            if (t >= 0) {
                int spins = 0;
                do {
                    wv = __hip_atomic_load(&words[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if ((wv >> 62) == 0) {
                        if (++spins > K7_SPIN_LIMIT) {
                            failed = true;
                            break;
                        }
                        __builtin_amdgcn_s_sleep(1);
                    }
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
    K7B_STAMP(5);
    const int64_t base = (int64_t)s_bcast[1];
    const bool fits = base + (int64_t)tile_bytes <= text_cap;
    if (text && tile_bytes && !fits && tid == 0) atomicExch(&state[1], 2ull);   // the host reads the size needed in text_off[n_rows]

    if (small) {
        // ---- the rows' offsets and verdicts from the scans --------------------------------------------------------------
        const bool bad = s_bad != 0;
        if (nr <= K7B_CAP) {   // everything a row needs is in LDS
            for (int64_t i = tid; i < nr; i += K7_BLOCK) {
                const uint32_t a = s_off[i], b = s_off[i + 1];
                const bool host = s_host[i] || (bad && s_ex[b] > s_ex[a]);
                text_off[r_lo + i] = base + s_pos[a];
                flag_out[r_lo + i] = host ? 2 : ((s_cnt[b] > s_cnt[a]) ? 0 : 1);
            }
        } else {
            for (int64_t r = r_lo + tid; r < r_hi; r += K7_BLOCK) {
                const uint32_t a = (uint32_t)(row_off[r] - b_lo), b = (uint32_t)(row_off[r + 1] - b_lo);
                const bool host = (width[r] == 0.0) || (height[r] == 0.0) || (class_id[r] < 0) || (bad && s_ex[b] > s_ex[a]);
                text_off[r] = base + s_pos[a];
                flag_out[r] = host ? 2 : ((s_cnt[b] > s_cnt[a]) ? 0 : 1);
            }
        }
        if (tile == n_tiles - 1 && tid == 0) text_off[n_rows] = base + tile_bytes;
        K7B_STAMP(6);
        if (!text || tile_bytes == 0 || !fits) return;
        unsigned char *dst = text + base;
        if (staged) {   // LDS -> memory in 16-byte stores; the words are realigned to the destination's phase
            const uint32_t phase = (uint32_t)(reinterpret_cast<uintptr_t>(dst) & 15u);
            const uint32_t end = phase + tile_bytes;
            const uint32_t n_chunks = (end + 15u) >> 4;
            unsigned char *aligned = dst - phase;
            const uint32_t *s32 = reinterpret_cast<const uint32_t *>(s_text);
            const uint32_t sh = (0u - phase) & 3u;
            for (uint32_t c = tid; c < n_chunks; c += K7_BLOCK) {
                const uint32_t lo = c << 4, hi = lo + 16u;
                if (lo >= phase && hi <= end) {
                    const uint32_t m = (lo - phase) >> 2;
                    const uint32_t d0 = s32[m], d1 = s32[m + 1], d2 = s32[m + 2], d3 = s32[m + 3], d4 = s32[m + 4];
                    uint4 v;
                    v.x = __builtin_amdgcn_alignbyte(d1, d0, sh);
                    v.y = __builtin_amdgcn_alignbyte(d2, d1, sh);
                    v.z = __builtin_amdgcn_alignbyte(d3, d2, sh);
                    v.w = __builtin_amdgcn_alignbyte(d4, d3, sh);
                    *reinterpret_cast<uint4 *>(aligned + lo) = v;
                } else {
                    const uint32_t a = lo < phase ? phase : lo, b = hi > end ? end : hi;
                    for (uint32_t i = a; i < b; ++i) aligned[i] = s_text[i - phase];
                }
            }
        } else {
#pragma unroll
            for (int k = 0; k < K7B_PER; ++k) {
                const int64_t j = (int64_t)k * K7_BLOCK + tid;
                if (line[k]) print_box(k, j, dst + s_pos[j]);
            }
        }
        K7B_STAMP(7);
        return;
    }
    for (int64_t r = r_lo + wave; r < r_hi; r += K7_WAVES) {   // each wave's lane 0 wrote these entries itself
        const int64_t at = base + __shfl((long long)text_off[r], 0);
        const int rflag = __shfl((int)flag_out[r], 0);
        if (lane == 0) text_off[r] = at;
        if (text && fits && rflag == 0) {
            const RowIn ri = k7_row(r, row_off, width, height, class_id);
            wave_row_print(ri, box4, sel, lane, text + at);
        }
    }
    if (tile == n_tiles - 1 && tid == 0) text_off[n_rows] = base + tile_bytes;
}

// dyd_set_option("k7_variant"): -1 = by the table's shape (default: 22 for one box per row, 30 from 1.02 boxes per row on),
// 2 = one 512-row tile per ticket, 22 = two, pipelined, 30 = tiles of 480 boxes, a lane per box
static int g_k7_variant = -1;
static unsigned long long *g_k7_trace = nullptr;   // tuning hook (single-tile kernel): 8 timestamps per tile
void set_k7_trace(void *p) { g_k7_trace = static_cast<unsigned long long *>(p); }
void set_k7_variant(int v) { g_k7_variant = (v == 2 || v == 22 || v == 30) ? v : -1; }

static int yolo_launch(const double *box4, const int32_t *row_off, const uint8_t *sel, const double *width,
                       const double *height, const int32_t *class_id, int64_t n_rows, int64_t n_boxes,
                       int64_t *text_off, uint8_t *flag, uint8_t *text, int64_t text_cap, int64_t *total_out, hipStream_t st) {
    // rows of several boxes go to the box-tiled kernel (a lane per box); it needs the box count, the last row offset
    bool by_box = g_k7_variant == 30;
    if (n_boxes < 0 && g_k7_variant != 2 && g_k7_variant != 22) {   // device-resident caller: fetch the last offset
        int32_t last = 0;
        DYD_HIP(hipMemcpyAsync(&last, row_off + n_rows, 4, hipMemcpyDeviceToHost, st));
        DYD_HIP(hipStreamSynchronize(st));
        n_boxes = last;
    }
    if (g_k7_variant != 2 && g_k7_variant != 22 && !by_box) by_box = n_boxes > n_rows + n_rows / 50;   // -1: by the table's shape (measured: at 1.05 boxes per row the row kernels already take 1.7x the box kernel's time, at 1.25 60x)
    const int64_t n_tiles = by_box ? (n_boxes > 0 ? ceil_div(n_boxes, (int64_t)K7B_WINDOW) : 1) : ceil_div(n_rows, (int64_t)K7_BLOCK * 2);
    void *scr = nullptr;
    const size_t state_bytes = (size_t)(n_tiles + 2) * 8;
    int rc = get_scratch(state_bytes + (by_box ? (size_t)(n_tiles + 1) * 16 : 0), &scr, st);
    if (rc) return rc;
    DYD_HIP(hipMemsetAsync(scr, 0, state_bytes, st));
    unsigned long long *state = static_cast<unsigned long long *>(scr);
    if (by_box) {   // (the first tile's search for box 0 gives row 0: leading empty rows are its own)
        int64_t *tile_row = reinterpret_cast<int64_t *>(state + n_tiles + 2), *tile_box = tile_row + n_tiles + 1;
        hipLaunchKernelGGL(k7_tile_rows_kernel, dim3((unsigned)ceil_div(n_tiles + 1, (int64_t)K7_BLOCK)), dim3(K7_BLOCK), 0, st, row_off,
                           n_rows, n_tiles, tile_row, tile_box);
        hipLaunchKernelGGL(k7_yolo_box_kernel, dim3((unsigned)n_tiles), dim3(K7_BLOCK), 0, st, box4, row_off, sel, width, height,
                           class_id, n_rows, n_tiles, tile_row, tile_box, text_off, flag, text, text_cap, state, g_k7_trace);
    }
    else if (g_k7_variant != 2)
        hipLaunchKernelGGL((k7_yolo_pair_kernel<2>), dim3((unsigned)ceil_div(n_tiles, 2)), dim3(K7_BLOCK), 0, st, box4, row_off, sel,
                           width, height, class_id, n_rows, text_off, flag, text, text_cap, state);
    else
        hipLaunchKernelGGL((k7_yolo_kernel<2>), dim3((unsigned)n_tiles), dim3(K7_BLOCK), 0, st, box4, row_off, sel, width, height,
                           class_id, n_rows, text_off, flag, text, text_cap, state, g_k7_trace);
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
                       const double *height, const int32_t *class_id, int64_t n_rows, int64_t n_boxes, int64_t *out_text_off,
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
    return yolo_launch(box4, row_off, sel_or_null, width, height, class_id, n_rows, n_boxes, out_text_off, out_flag,
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
                     d_cid.as<int32_t>(), n_rows, n_boxes, d_toff.as<int64_t>(), d_flag.as<uint8_t>(), nullptr, 0, &total, st);
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
                         d_cid.as<int32_t>(), n_rows, n_boxes, d_toff.as<int64_t>(), d_flag.as<uint8_t>(),
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
