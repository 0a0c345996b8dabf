// host_json_fast.h — the fast lane of the native flatten / emit (included by host_json.cpp, inside its anonymous namespace).
//
// The exact walker of host_json.cpp (Parser / walk_cell) parses every annotation cell TWICE — once to collect the points, once
// more, after K1, to re-emit the document around the new ptLists — through std::string appends, per-object key sets and
// exceptions.  On the 10M-row configuration that host work, not the kernels, decides rows/s, so regular cells
// take this lane instead:
//
//   pass 1 (scan)  ONE parse per cell.  Points go to the SoA arrays (value + "was an int token" flag); everything else of the
//                  document is written straight away in its canonical json.dumps(..., ensure_ascii=False) form into a
//                  per-thread SEGMENT buffer with a hole where each ptList stood (reference processor.py:262-279);
//   pass 2 (emit)  no parsing: per cell the segments are copied and each hole is filled with the two corner points K1 chose,
//                  printed from the stored VALUE — repr(float(tok)) and str(int(tok)) are functions of the value alone, so the
//                  selected token itself is not needed (ints are exact below 2^53, anything larger never gets here).
//
// The lane is deliberately narrow: at the first construct it does not reproduce byte for byte (escaped or repeated keys, odd
// number spellings, non-array ptList ...) it BAILS, the cell's partial output is rolled back, and the exact walker processes
// the cell as before — so classification (regular / undecodable / irregular) and every edge case remain the exact walker's.
#pragma once

// growable raw array: no zero fill, realloc growth (mremap for the multi-GB arrays of a 1M-row batch).  adopt() lends it memory
// it does not own (a staging slot's pinned arena): it fills that in place and moves to memory of its own only if it outgrows it.
template <class T>
struct Raw {
    T *p = nullptr;
    size_t n = 0, cap = 0;
    bool lent = false;               // p is somebody else's memory: never freed or realloc'ed here
    Raw() = default;
    Raw(const Raw &) = delete;
    Raw &operator=(const Raw &) = delete;
    ~Raw() { if (!lent) free(p); }
    void grow(size_t need) {
        size_t c = cap * 2;
        if (c < need) c = need;
        if (c < 1024) c = 1024;
        T *q;
        if (lent) {
            q = static_cast<T *>(malloc(c * sizeof(T)));
            if (!q) throw std::bad_alloc();
            if (n) memcpy(q, p, n * sizeof(T));
            lent = false;
        } else {
            q = static_cast<T *>(realloc(p, c * sizeof(T)));
            if (!q) throw std::bad_alloc();
        }
        p = q;
        cap = c;
        huge_hint(q, c * sizeof(T));
    }
    // The arrays of a million-cell pass are hundreds of megabytes per thread: in 4 KB pages that is a million first-touch faults
    // while scanning and a million page-table entries to tear down when the pass is freed (0.15 s per million cells).  Where the
    // kernel offers transparent huge pages on request (`madvise` mode) the 2 MB-aligned inside of a big array asks for them.
    static void huge_hint(void *q, size_t bytes) {
#if defined(__linux__) && defined(MADV_HUGEPAGE)
        if (bytes < (size_t)8 << 20) return;
        const uintptr_t a = (reinterpret_cast<uintptr_t>(q) + ((uintptr_t)2 << 20) - 1) & ~(((uintptr_t)2 << 20) - 1);
        const uintptr_t e = (reinterpret_cast<uintptr_t>(q) + bytes) & ~(((uintptr_t)2 << 20) - 1);
        if (e > a) (void)madvise(reinterpret_cast<void *>(a), e - a, MADV_HUGEPAGE);
#else
        (void)q; (void)bytes;
#endif
    }
    inline void need(size_t k) { if (n + k > cap) grow(n + k); }
    inline void push(T v) { if (n == cap) grow(n + 1); p[n++] = v; }
    inline void put(const T *s, size_t k) { if (!k) return; need(k); memcpy(p + n, s, k * sizeof(T)); n += k; }
    void clear_free() { if (!lent) free(p); p = nullptr; n = cap = 0; lent = false; }
    void adopt(T *mem, size_t elems) { clear_free(); p = mem; cap = elems; lent = elems > 0; if (!lent) p = nullptr; }   // borrowed
    void own(T *mem, size_t elems) { clear_free(); p = mem; cap = elems; }                                                // a malloc'ed block
    T *disown(size_t *elems) { T *q = lent ? nullptr : p; *elems = q ? cap : 0; if (!lent) { p = nullptr; n = cap = 0; } return q; }
};

inline bool fj_lean_points() {
    const char *env = getenv("DYD_JSON_FAST");
    return !(env && env[0] == '2');
}

struct FastPart {
    int64_t lo = 0, hi = 0;          // cell range [lo, hi)
    Raw<double> xy;                  // 2 per point
    Raw<uint8_t> isint;              // per point: bit 0 = x was an int token, bit 1 = y
    Raw<int32_t> npts;               // per box
    Raw<uint32_t> hole;              // per box: where its ptList goes inside the cell's segment text
    Raw<char> seg;                   // canonical text of the lane's cells, ptLists cut out
    Raw<int64_t> seg_off;            // per cell: start in seg (hi - lo + 1 entries)
    Raw<uint8_t> lane;               // per cell: 1 = segments (this lane), 0 = the exact walker re-parses it in pass 2
    Raw<int32_t> cell_boxes;         // per cell: boxes it contributed
    // views pass 2 reads the points through: xyv = the part's first point, local box b starts at point ptv[b] - ptv_bias
    const double *xyv = nullptr;
    const int32_t *ptv = nullptr;
    int32_t ptv_bias = 0;
    // pass 2
    Raw<char> out;
    Raw<int64_t> out_len;            // per cell
    Raw<int64_t> out_off;            // pipeline: prefix of out_len (cells + 1)
    int64_t lane_count = 0;          // pipeline: cells this lane took (kept after `lane` is freed)
    size_t box_base = 0, pt_base = 0;  // global index of the part's first box / point
    bool lean_points = fj_lean_points();   // DYD_JSON_FAST=2: every point through the general walk (tests: both must agree)
};

inline bool fj_is_digit(char c) { return c >= '0' && c <= '9'; }
inline bool fj_is_ws(char c) { return c == ' ' || c == '\n' || c == '\r' || c == '\t'; }

// integer value -> decimal text (|v| < 2^63)
inline size_t fj_put_int(char *dst, int64_t v) {
    char tmp[24];
    size_t k = 0;
    uint64_t u = v < 0 ? (uint64_t)(-(v + 1)) + 1u : (uint64_t)v;
    do { tmp[k++] = (char)('0' + u % 10); u /= 10; } while (u);
    size_t o = 0;
    if (v < 0) dst[o++] = '-';
    while (k) dst[o++] = tmp[--k];
    return o;
}

struct FastNum {
    const char *b = nullptr, *e = nullptr;   // the token
    bool is_int = false, exact = false, big = false;   // big: more digits than the lane evaluates (value not set)
    double v = 0.0;
    // canonical spelling: [b, canon_e) when canon_e != nullptr (ints, and floats whose trimmed token IS their repr), else print v
    const char *canon_e = nullptr;
    bool minus_zero_int = false;             // the int token "-0" prints as "0"
};

struct FastKeys {   // duplicate-key check of one JSON object (bail above 24 members)
    const char *b[24];
    uint32_t len[24];
    int n = 0;
    inline bool add(const char *kb, size_t kl) {   // false -> bail
        for (int i = 0; i < n; ++i)
            if (len[i] == kl && !memcmp(b[i], kb, kl)) return false;
        if (n == 24) return false;
        b[n] = kb;
        len[n] = (uint32_t)kl;
        ++n;
        return true;
    }
};

struct FastCell {
    const char *p, *end;
    FastPart &A;
    std::string &tmp;     // scratch of the thread (escaped strings, general float printing)
    int depth = 0;
    int box = 0;
    bool big_int = false;
    WH w, h;

    FastCell(const char *b, const char *e, FastPart &a, std::string &t) : p(b), end(e), A(a), tmp(t) {}

    inline void ws() { while (p < end && fj_is_ws(*p)) ++p; }
    inline void put(const char *s, size_t k) { A.seg.put(s, k); }
    inline void putc(char c) { A.seg.push(c); }
    template <size_t N>
    inline void lit(const char (&s)[N]) { A.seg.put(s, N - 1); }

    // p at '"': a string without escapes / control characters; b..e = the raw span between the quotes
    inline bool plain_string(const char *&b, const char *&e) {
        const char *q = p + 1;
        while (q < end) {
            const unsigned char c = (unsigned char)*q;
            if (c == '"') { b = p + 1; e = q; p = q + 1; return true; }
            if (c == '\\' || c < 0x20) return false;
            ++q;
        }
        return false;
    }

    // p at '"': copies the string in its canonical spelling
    template <bool OUT>
    bool string_value() {
        const char *b, *e;
        if (plain_string(b, e)) {
            if (OUT) put(b - 1, (size_t)(e - b) + 2);
            return true;
        }
        try {   // escapes: the exact decoder / re-encoder, for this one string
            Parser ps{p, end};
            const Span s = ps.string_token();
            if (OUT) {
                tmp.clear();
                ps.emit_string(tmp, s);
                put(tmp.data(), tmp.size());
            }
            p = ps.p;
            return true;
        } catch (Fail) {
            return false;
        }
    }

    // p at the first character of a number (digit, '-', 'N', 'I').  false -> bail.
    bool number(FastNum &t) {
        const char *q = p;
        t = FastNum();
        t.b = q;
        bool neg = false;
        if (*q == '-') { neg = true; ++q; if (q >= end) return false; }
        if (*q == 'N') {
            if (neg || end - q < 3 || memcmp(q, "NaN", 3)) return false;
            q += 3;
            t.v = NAN; t.canon_e = q;
        } else if (*q == 'I') {
            if (end - q < 8 || memcmp(q, "Infinity", 8)) return false;
            q += 8;
            t.v = neg ? -INFINITY : INFINITY; t.canon_e = q;
        } else {
            const char *ib = q;
            if (*q == '0') {
                ++q;
                if (q < end && fj_is_digit(*q)) return false;   // "01": NUMBER_RE stops after the 0
            } else if (*q >= '1' && *q <= '9') {
                while (q < end && fj_is_digit(*q)) ++q;
            } else {
                return false;
            }
            const char *ie = q, *fb = nullptr, *fe = nullptr;
            if (q < end && *q == '.') {
                if (q + 1 < end && fj_is_digit(q[1])) {
                    fb = ++q;
                    while (q < end && fj_is_digit(*q)) ++q;
                    fe = q;
                } else {
                    return false;
                }
            }
            bool has_exp = false;
            if (q < end && (*q == 'e' || *q == 'E')) {
                const char *r = q + 1;
                if (r < end && (*r == '+' || *r == '-')) ++r;
                if (r < end && fj_is_digit(*r)) {
                    while (r < end && fj_is_digit(*r)) ++r;
                    q = r;
                    has_exp = true;
                } else {
                    return false;
                }
            }
            const size_t n_int = (size_t)(ie - ib), n_frac = fb ? (size_t)(fe - fb) : 0;
            if (!fb && !has_exp) {   // an int token
                t.is_int = true;
                t.canon_e = q;
                if (n_int <= 18) {
                    uint64_t m = 0;
                    for (const char *d = ib; d < ie; ++d) m = m * 10 + (uint64_t)(*d - '0');
                    t.exact = m <= 9007199254740992ull;
                    t.v = neg ? -(double)m : (double)m;
                    t.minus_zero_int = neg && m == 0;
                } else {
                    t.big = true;
                    if (q - t.b > 4000) return false;   // int() refuses more than 4300 digits: the exact walker's business
                }
            } else if (has_exp || n_int + n_frac > 19) {
                char buf[64];
                const size_t tl = (size_t)(q - t.b);
                if (tl >= sizeof(buf)) return false;
                memcpy(buf, t.b, tl);
                buf[tl] = 0;
                t.v = strtod(buf, nullptr);   // glibc: correctly rounded, like float()
            } else {
                uint64_t m = 0;
                int first_nz = -1, last_nz = -1, idx = 0;
                for (const char *d = ib; d < ie; ++d, ++idx) { m = m * 10 + (uint64_t)(*d - '0'); if (*d != '0') { if (first_nz < 0) first_nz = idx; last_nz = idx; } }
                for (const char *d = fb; d < fe; ++d, ++idx) { m = m * 10 + (uint64_t)(*d - '0'); if (*d != '0') { if (first_nz < 0) first_nz = idx; last_nz = idx; } }
                static const double p10[] = {1e0, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8, 1e9, 1e10, 1e11,
                                             1e12, 1e13, 1e14, 1e15, 1e16, 1e17, 1e18, 1e19, 1e20, 1e21, 1e22};
                if (m <= 9007199254740992ull) {
                    const double d = (double)m / p10[n_frac];   // exact operands, one correctly rounded division (Clinger)
                    t.v = neg ? -d : d;
                } else {
                    char buf[64];
                    const size_t tl = (size_t)(q - t.b);
                    memcpy(buf, t.b, tl);
                    buf[tl] = 0;
                    t.v = strtod(buf, nullptr);
                }
                // repr(float(tok)) == the token minus trailing fractional zeros when it has at most 15 significant digits
                // (every such decimal survives the round trip through f64, so the shortest round-tripping spelling is the
                // token itself) and repr prints in fixed notation: 1e-4 <= |v| < 1e16, or v == 0
                const int sig = first_nz < 0 ? 0 : last_nz - first_nz + 1;
                const int lead_frac_zeros = (first_nz < 0) ? 0 : first_nz - (int)n_int;   // zeros between the point and the first digit
                const bool fixed = first_nz < 0 || first_nz < (int)n_int || lead_frac_zeros <= 3;
                if (sig <= 15 && n_int <= 16 && fixed) {
                    const char *ce = fe;
                    while (ce > fb + 1 && ce[-1] == '0') --ce;
                    t.canon_e = ce;
                }
            }
        }
        if (q < end) {   // what follows must end the value, or the document is not JSON (the exact walker says which)
            const char c = *q;
            if (!(c == ',' || c == '}' || c == ']' || fj_is_ws(c))) return false;
        }
        t.e = q;
        p = q;
        return true;
    }

    inline void put_number(const FastNum &t) {
        if (t.minus_zero_int) { putc('0'); return; }
        if (t.canon_e) { put(t.b, (size_t)(t.canon_e - t.b)); return; }
        tmp.clear();
        append_py_float(tmp, t.v);
        put(tmp.data(), tmp.size());
    }

    inline bool literal(const char *s, size_t k) {
        if ((size_t)(end - p) < k || memcmp(p, s, k)) return false;
        p += k;
        return true;
    }

    // any JSON value, canonical copy (OUT) or validation only
    template <bool OUT>
    bool value() {
        ws();
        if (p >= end) return false;
        switch (*p) {
            case '{': {
                if (++depth > 200) return false;
                ++p;
                if (OUT) putc('{');
                ws();
                if (p < end && *p == '}') { ++p; if (OUT) putc('}'); --depth; return true; }
                FastKeys ks;
                bool first = true;
                while (true) {
                    ws();
                    if (p >= end || *p != '"') return false;
                    const char *kb, *ke;
                    if (!plain_string(kb, ke)) return false;
                    if (!ks.add(kb, (size_t)(ke - kb))) return false;
                    if (OUT) {
                        if (!first) lit(", ");
                        put(kb - 1, (size_t)(ke - kb) + 2);
                        lit(": ");
                    }
                    first = false;
                    ws();
                    if (p >= end || *p != ':') return false;
                    ++p;
                    if (!value<OUT>()) return false;
                    ws();
                    if (p < end && *p == ',') { ++p; continue; }
                    if (p < end && *p == '}') { ++p; break; }
                    return false;
                }
                if (OUT) putc('}');
                --depth;
                return true;
            }
            case '[': {
                if (++depth > 200) return false;
                ++p;
                if (OUT) putc('[');
                ws();
                if (p < end && *p == ']') { ++p; if (OUT) putc(']'); --depth; return true; }
                bool first = true;
                while (true) {
                    if (OUT && !first) lit(", ");
                    first = false;
                    if (!value<OUT>()) return false;
                    ws();
                    if (p < end && *p == ',') { ++p; continue; }
                    if (p < end && *p == ']') { ++p; break; }
                    return false;
                }
                if (OUT) putc(']');
                --depth;
                return true;
            }
            case '"': return string_value<OUT>();
            case 't': if (!literal("true", 4)) return false; if (OUT) lit("true"); return true;
            case 'f': if (!literal("false", 5)) return false; if (OUT) lit("false"); return true;
            case 'n': if (!literal("null", 4)) return false; if (OUT) lit("null"); return true;
            default: {
                const char c = *p;
                if (!(c == '-' || c == 'N' || c == 'I' || fj_is_digit(c))) return false;
                FastNum t;
                if (!number(t)) return false;
                if (OUT) put_number(t);
                return true;
            }
        }
    }

    // the hole of one ptList: npts points were pushed for it
    inline void close_box(int32_t npts) {
        A.npts.push(npts);
        A.hole.push((uint32_t)(A.seg.n - (size_t)A.seg_off.p[A.seg_off.n - 1]));
        ++box;
    }

    // A coordinate in the spelling json.dumps gives it: [-]digits[.digits], at most 19 digits, no exponent, mantissa <= 2^53 — the
    // value is then one exact int -> double conversion, or one correctly rounded division of exact operands (as in number()).
    // false: not that shape (the caller walks the point the general way).  The caller guarantees 48 readable bytes at q.
    // (A loop-free version — eight bytes at a time, SWAR digit test and the three-multiply reduction — was 12 % slower here.)
    static inline bool lean_coord(const char *&q, double &v, bool &is_int) {
        const char *s = q;
        const bool neg = *s == '-';
        s += neg;
        const char *ib = s;
        uint64_t m = 0;
        while (fj_is_digit(*s) && s - ib < 20) { m = m * 10 + (uint64_t)(*s - '0'); ++s; }
        const int n_int = (int)(s - ib);
        if (n_int == 0 || (ib[0] == '0' && n_int > 1)) return false;
        int n_frac = 0;
        if (*s == '.') {
            const char *fb = ++s;
            while (fj_is_digit(*s) && s - fb < 20) { m = m * 10 + (uint64_t)(*s - '0'); ++s; }
            n_frac = (int)(s - fb);
            if (n_frac == 0) return false;
        }
        if (n_int + n_frac > 19 || m > 9007199254740992ull || *s == 'e' || *s == 'E' || fj_is_digit(*s)) return false;
        static const double p10[] = {1e0, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8, 1e9, 1e10, 1e11, 1e12, 1e13, 1e14, 1e15, 1e16, 1e17, 1e18};
        const double d = n_frac ? (double)m / p10[n_frac] : (double)m;
        v = neg ? -d : d;
        is_int = n_frac == 0;
        q = s;
        return true;
    }

    // p at the '{' of a point spelled {"x": <coord>, "y": <coord>} (blanks after ':' and ',' optional): pushes it.  false, with
    // nothing consumed, for every other spelling.
    inline bool lean_point() {
        if (end - p < 112) return false;                       // every read below stays inside the cell
        const char *q = p;
        if (memcmp(q, "{\"x\":", 5)) return false;
        q += 5;
        q += *q == ' ';
        double x, y;
        bool xi, yi;
        if (!lean_coord(q, x, xi)) return false;
        if (*q != ',') return false;
        ++q;
        q += *q == ' ';
        if (memcmp(q, "\"y\":", 4)) return false;
        q += 4;
        q += *q == ' ';
        if (!lean_coord(q, y, yi)) return false;
        if (*q != '}') return false;
        p = q + 1;
        A.xy.need(2);
        A.xy.p[A.xy.n] = x;
        A.xy.p[A.xy.n + 1] = y;
        A.xy.n += 2;
        A.isint.push((uint8_t)((xi ? 1 : 0) | (yi ? 2 : 0)));
        if ((xi && std::fabs(x) > 33554432.0) || (yi && std::fabs(y) > 33554432.0)) big_int = true;
        return true;
    }

    // p at the '[' of a ptList (reference :253): pushes the valid points; no output
    bool ptlist(int32_t &count) {
        ++p;
        count = 0;
        ws();
        if (p < end && *p == ']') { ++p; return true; }
        while (true) {
            ws();
            if (p >= end) return false;
            if (*p == '{' && A.lean_points && lean_point()) {
                ++count;
            } else if (*p == '{') {
                ++p;
                bool hx = false, hy = false;
                FastNum nx, ny;
                ws();
                if (p < end && *p == '}') {
                    ++p;
                } else {
                    FastKeys ks;
                    while (true) {
                        ws();
                        if (p >= end || *p != '"') return false;
                        const char *kb, *ke;
                        if (!plain_string(kb, ke)) return false;
                        if (!ks.add(kb, (size_t)(ke - kb))) return false;
                        ws();
                        if (p >= end || *p != ':') return false;
                        ++p;
                        const bool isx = (ke - kb == 1 && *kb == 'x'), isy = (ke - kb == 1 && *kb == 'y');
                        if (isx || isy) {
                            ws();
                            if (p >= end) return false;
                            const char c = *p;
                            if (!(c == '-' || c == 'N' || c == 'I' || fj_is_digit(c))) return false;   // None / str / ...: exact walker
                            FastNum &t = isx ? nx : ny;
                            if (!number(t)) return false;
                            if (t.big || (t.is_int && !t.exact)) return false;
                            if (isx) hx = true; else hy = true;
                        } else if (!value<false>()) {
                            return false;
                        }
                        ws();
                        if (p < end && *p == ',') { ++p; continue; }
                        if (p < end && *p == '}') { ++p; break; }
                        return false;
                    }
                }
                if (hx && hy) {
                    A.xy.need(2);
                    A.xy.p[A.xy.n] = nx.v;
                    A.xy.p[A.xy.n + 1] = ny.v;
                    A.xy.n += 2;
                    A.isint.push((uint8_t)((nx.is_int ? 1 : 0) | (ny.is_int ? 2 : 0)));
                    if ((nx.is_int && std::fabs(nx.v) > 33554432.0) || (ny.is_int && std::fabs(ny.v) > 33554432.0)) big_int = true;
                    ++count;
                }
            } else if (!value<false>()) {   // a non-dict element is not a valid point
                return false;
            }
            ws();
            if (p < end && *p == ',') { ++p; continue; }
            if (p < end && *p == ']') { ++p; return true; }
            return false;
        }
    }

    // generic object walker with one special member; SPECIAL(key) handles the value and returns 0 bail / 1 handled / 2 not mine
    bool polygon() {   // p at '{'
        ++p;
        putc('{');
        bool first = true, seen = false;
        FastKeys ks;
        ws();
        if (p < end && *p == '}') {
            ++p;
        } else {
            while (true) {
                ws();
                if (p >= end || *p != '"') return false;
                const char *kb, *ke;
                if (!plain_string(kb, ke)) return false;
                if (!ks.add(kb, (size_t)(ke - kb))) return false;
                if (!first) lit(", ");
                first = false;
                put(kb - 1, (size_t)(ke - kb) + 2);
                lit(": ");
                ws();
                if (p >= end || *p != ':') return false;
                ++p;
                if (ke - kb == 6 && !memcmp(kb, "ptList", 6)) {
                    ws();
                    if (p >= end || *p != '[') return false;   // None / str / dict / number ptList: exact walker
                    seen = true;
                    int32_t cnt;
                    if (!ptlist(cnt)) return false;
                    close_box(cnt);
                } else if (!value<true>()) {
                    return false;
                }
                ws();
                if (p < end && *p == ',') { ++p; continue; }
                if (p < end && *p == '}') { ++p; break; }
                return false;
            }
        }
        if (!seen) {   // obj.get("polygon", {}).get("ptList", []) -> []; the ptList member is appended (:276)
            if (!first) lit(", ");
            lit("\"ptList\": ");
            close_box(0);
        }
        putc('}');
        return true;
    }

    bool object() {   // p at '{': one dict element of "objects"
        ++p;
        putc('{');
        bool first = true, seen = false;
        FastKeys ks;
        ws();
        if (p < end && *p == '}') {
            ++p;
        } else {
            while (true) {
                ws();
                if (p >= end || *p != '"') return false;
                const char *kb, *ke;
                if (!plain_string(kb, ke)) return false;
                if (!ks.add(kb, (size_t)(ke - kb))) return false;
                if (!first) lit(", ");
                first = false;
                put(kb - 1, (size_t)(ke - kb) + 2);
                lit(": ");
                ws();
                if (p >= end || *p != ':') return false;
                ++p;
                if (ke - kb == 7 && !memcmp(kb, "polygon", 7)) {
                    ws();
                    if (p >= end || *p != '{') return false;   // None / list ...: AttributeError in Python -> exact walker
                    seen = true;
                    if (!polygon()) return false;
                } else if (!value<true>()) {
                    return false;
                }
                ws();
                if (p < end && *p == ',') { ++p; continue; }
                if (p < end && *p == '}') { ++p; break; }
                return false;
            }
        }
        if (!seen) {   // :274-276 — a polygon dict is added at the end
            if (!first) lit(", ");
            lit("\"polygon\": {\"ptList\": ");
            close_box(0);
            putc('}');
        }
        putc('}');
        return true;
    }

    bool read_wh(WH &dst) {
        ws();
        if (p >= end) return false;
        const char c = *p;
        if (c == 'n') { dst.kind = 0; return value<true>(); }
        if (c == '-' || c == 'N' || c == 'I' || fj_is_digit(c)) {
            FastNum t;
            if (!number(t)) return false;
            put_number(t);
            if (t.big || (t.is_int && !t.exact)) { dst.kind = 3; return true; }
            dst.kind = t.is_int ? 1 : 2;
            dst.v = t.v;
            return true;
        }
        dst.kind = 3;
        return value<true>();
    }

    // the whole cell; false -> bail (the caller rolls the part back to its marks)
    bool cell() {
        ws();
        if (p >= end || *p != '{') return false;
        ++p;
        putc('{');
        bool first = true, seen = false;
        FastKeys ks;
        ws();
        if (p < end && *p == '}') {
            ++p;
        } else {
            while (true) {
                ws();
                if (p >= end || *p != '"') return false;
                const char *kb, *ke;
                if (!plain_string(kb, ke)) return false;
                if (!ks.add(kb, (size_t)(ke - kb))) return false;
                if (!first) lit(", ");
                first = false;
                put(kb - 1, (size_t)(ke - kb) + 2);
                lit(": ");
                ws();
                if (p >= end || *p != ':') return false;
                ++p;
                const size_t kl = (size_t)(ke - kb);
                if (kl == 7 && !memcmp(kb, "objects", 7)) {
                    ws();
                    if (p >= end || *p != '[') return false;   // null / dict / str / number "objects": exact walker
                    seen = true;
                    ++p;
                    putc('[');
                    bool efirst = true;
                    ws();
                    if (p < end && *p == ']') {
                        ++p;
                    } else {
                        while (true) {
                            ws();
                            if (p < end && *p == '{') {
                                if (!efirst) lit(", ");
                                efirst = false;
                                if (!object()) return false;
                            } else if (!value<false>()) {   // non-dict objects are dropped (:270)
                                return false;
                            }
                            ws();
                            if (p < end && *p == ',') { ++p; continue; }
                            if (p < end && *p == ']') { ++p; break; }
                            return false;
                        }
                    }
                    putc(']');
                } else if (kl == 5 && !memcmp(kb, "width", 5)) {
                    if (!read_wh(w)) return false;
                } else if (kl == 6 && !memcmp(kb, "height", 6)) {
                    if (!read_wh(h)) return false;
                } else if (!value<true>()) {
                    return false;
                }
                ws();
                if (p < end && *p == ',') { ++p; continue; }
                if (p < end && *p == '}') { ++p; break; }
                return false;
            }
        }
        ws();
        if (p != end) return false;   // "Extra data"
        if (!seen) { if (!first) lit(", "); lit("\"objects\": []"); }   // data["objects"] = [] (:278)
        putc('}');
        return true;
    }
};

// repr(v) for a double that is the nearest one to a decimal with k <= 4 fractional digits and fewer than 15 significant ones
// (coordinates usually are: pixels with a couple of decimals): c = round(v * 10^k) as an integer; if c / 10^k — one correctly
// rounded division of exact operands — gives v back, that decimal round-trips, and a decimal of at most 15 significant digits
// that round-trips IS the shortest one (two different such decimals never share a double), so repr(v) is c with the point put
// in and trailing zeros dropped.  Returns 0 when v is not of that kind (the caller then formats it the general way).
inline size_t fj_put_short_float(char *dst, double v) {
    const double a = v < 0 ? -v : v;
    if (!(a >= 1e-4 && a < 1e10)) return 0;   // keeps c below 1e14 and repr in fixed notation (zero and -0.0: the general way)
    static const double p10[] = {1e1, 1e2, 1e3, 1e4};
    for (int k = 1; k <= 4; ++k) {
        const double scaled = a * p10[k - 1];
        const int64_t c = (int64_t)(scaled + 0.5);
        if ((double)c / p10[k - 1] != a) continue;
        // digits of c, point k places from the right, trailing zeros of the fraction trimmed (one digit stays)
        char tmp[24];
        int nd = 0;
        int64_t u = c;
        do { tmp[nd++] = (char)('0' + u % 10); u /= 10; } while (u);
        while (nd <= k) tmp[nd++] = '0';            // 0.05 -> c = 5: leading zeros up to the units digit
        size_t o = 0;
        if (v < 0) dst[o++] = '-';
        for (int i = nd - 1; i >= k; --i) dst[o++] = tmp[i];
        dst[o++] = '.';
        int last = 0;                               // lowest fraction digit to print
        while (last < k - 1 && tmp[last] == '0') ++last;
        for (int i = k - 1; i >= last; --i) dst[o++] = tmp[i];
        return o;
    }
    return 0;
}

// pass 2 helper: one coordinate printed from its value
inline void fj_put_coord(Raw<char> &o, double v, bool is_int, std::string &tmp) {
    o.need(32);
    if (is_int) {
        o.n += fj_put_int(o.p + o.n, (int64_t)v);   // exact: |v| <= 2^53; -0.0 (the token "-0") prints as 0
        return;
    }
    const size_t k = fj_put_short_float(o.p + o.n, v);
    if (k) { o.n += k; return; }
    tmp.clear();
    append_py_float(tmp, v);
    o.put(tmp.data(), tmp.size());
}

// pass 2: the corner points of one box (:254-260)
inline bool fj_put_corners(Raw<char> &o, const double *xy, const uint8_t *isint, int32_t npts, const int32_t *a, std::string &tmp) {
    static const char kNull[] = "[{\"x\": null, \"y\": null}, {\"x\": null, \"y\": null}]";
    if (npts == 0) { o.put(kNull, sizeof(kNull) - 1); return true; }
    for (int i = 0; i < 4; ++i)
        if (a[i] < 0 || a[i] >= npts) return false;
    static const char *const lead[4] = {"[{\"x\": ", ", \"y\": ", "}, {\"x\": ", ", \"y\": "};
    for (int i = 0; i < 4; ++i) {
        o.put(lead[i], strlen(lead[i]));
        const int32_t k = a[i];
        fj_put_coord(o, xy[2 * (size_t)k + (i & 1)], (isint[k] >> (i & 1)) & 1, tmp);
    }
    o.put("}]", 2);
    return true;
}
