// host_json.cpp — native flatten / emit for the annotation cells (host code, no HIP).
//
// the reference spends ~45 % of its replace step and ~35 % of its IoU step inside
// CPython's json.loads / json.dumps (core/processor.py:266, :279, :289, :346).  This file is a
// schema-specialised JSON scanner + canonical re-emitter that produces, for REGULAR cells,
//   * the SoA buffers K1 / K2 consume (points + offsets, two-point boxes + row offsets), and
//   * byte-for-byte the text json.dumps(doc, ensure_ascii=False) would produce after the reference's
//     rewrite of every ptList (core/processor.py:262-279),
// and classifies every other cell as IRREGULAR so that the Python flatten/emit of flatten.py
// (which follows the reference accessor by accessor, including the exceptions it raises) handles
// it.  A cell is regular when: it decodes under Python's json grammar; the top level is an object;
// "objects" is absent or an array; every dict element's "polygon" is absent or an object whose
// "ptList" is absent or an array; every point that has both "x" and "y" carries plain numbers
// (ints of magnitude <= 2^53, floats, NaN/Infinity literals); no object on the rewritten path has
// a duplicate key; strings hold no lone surrogate.  Cells that fail to decode are regular too: the
// reference maps them to None (:280-281).
//
// Threads: cells are independent; both entry points split the cell range over std::threads.
#include <sys/mman.h>
#include <algorithm>
#include <charconv>
#include <chrono>
#include <cstdio>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/dyd.h"

namespace {

enum CellStatus : uint8_t { CELL_OK = 0, CELL_UNDECODABLE = 1, CELL_IRREGULAR = 2, CELL_MISSING = 3 };

struct Span {
    const char *b, *e;
};

struct Fail {  // thrown as plain ints to keep the parser small: 1 = JSON decode error, 2 = irregular
    int code;
};

inline bool is_ws(char c) { return c == ' ' || c == '\t' || c == '\n' || c == '\r'; }

// ---------------------------------------------------------------------------------------------------
// number / string canonicalisation (what json.dumps prints for the value json.loads produced)
// ---------------------------------------------------------------------------------------------------
// float.__repr__: shortest digits that round-trip; fixed notation for -4 < decpt <= 16, else d.ddde±XX
void append_py_float(std::string &out, double v) {
    if (std::isnan(v)) { out += "NaN"; return; }
    if (std::isinf(v)) { out += (v < 0 ? "-Infinity" : "Infinity"); return; }
    if (v == 0.0) { out += (std::signbit(v) ? "-0.0" : "0.0"); return; }
    char buf[64];
    auto r = std::to_chars(buf, buf + sizeof(buf) - 1, v, std::chars_format::scientific);  // d[.ddd]e±XX, shortest
    *r.ptr = 0;
    const char *p = buf, *end = r.ptr;
    if (*p == '-') { out += '-'; ++p; }
    const char *epos = static_cast<const char *>(memchr(p, 'e', end - p));
    std::string digits;
    digits += p[0];
    if (p + 1 < epos && p[1] == '.') digits.append(p + 2, epos);
    const int exp10 = atoi(epos + 1);
    const int nd = (int)digits.size();
    const int decpt = exp10 + 1;  // value = 0.DIGITS * 10^decpt
    if (decpt > -4 && decpt <= 16) {
        if (decpt <= 0) {
            out += "0.";
            out.append((size_t)(-decpt), '0');
            out += digits;
        } else if (decpt >= nd) {
            out += digits;
            out.append((size_t)(decpt - nd), '0');
            out += ".0";
        } else {
            out.append(digits, 0, (size_t)decpt);
            out += '.';
            out.append(digits, (size_t)decpt, std::string::npos);
        }
    } else {
        out += digits[0];
        if (nd > 1) {
            out += '.';
            out.append(digits, 1, std::string::npos);
        }
        char eb[16];
        const int e = decpt - 1;
        snprintf(eb, sizeof(eb), "e%c%02d", e < 0 ? '-' : '+', e < 0 ? -e : e);
        out += eb;
    }
}

}  // namespace
namespace dyd_host {
void append_py_float_public(std::string &out, double v) { append_py_float(out, v); }
}  // namespace dyd_host
namespace {

struct Num {
    bool is_int;
    double v;       // value as double (exact for ints within 2^53)
    bool exact;     // ints only: |value| <= 2^53
};

// token = a JSON number as matched by Python's NUMBER_RE, or NaN / Infinity / -Infinity
Num classify_number(Span t) {
    Num n{};
    const size_t len = (size_t)(t.e - t.b);
    if ((len == 3 && !memcmp(t.b, "NaN", 3))) { n.is_int = false; n.v = NAN; return n; }
    if ((len == 8 && !memcmp(t.b, "Infinity", 8))) { n.is_int = false; n.v = INFINITY; return n; }
    if ((len == 9 && !memcmp(t.b, "-Infinity", 9))) { n.is_int = false; n.v = -INFINITY; return n; }
    bool is_int = true;
    for (const char *p = t.b; p < t.e; ++p)
        if (*p == '.' || *p == 'e' || *p == 'E') { is_int = false; break; }
    n.is_int = is_int;
    // fast path: <= 15 significant digits and |10^k| <= 10^22 -> mantissa and power are exact doubles, so
    // one IEEE multiply / divide is correctly rounded (Clinger); everything else goes to strtod
    {
        const char *q = t.b;
        const bool neg = (*q == '-');
        if (neg) ++q;
        uint64_t mant = 0;
        int nd = 0, frac = 0;
        bool seen_dot = false, simple = true;
        for (; q < t.e; ++q) {
            const char c = *q;
            if (c >= '0' && c <= '9') {
                if (nd < 19) { mant = mant * 10 + (uint64_t)(c - '0'); }
                if (mant != 0 || nd > 0) ++nd;
                if (seen_dot) ++frac;
            } else if (c == '.') {
                seen_dot = true;
            } else {
                simple = false;  // exponent: rare, take strtod
                break;
            }
        }
        static const double p10[] = {1e0, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8, 1e9, 1e10, 1e11,
                                     1e12, 1e13, 1e14, 1e15, 1e16, 1e17, 1e18, 1e19, 1e20, 1e21, 1e22};
        if (simple && nd <= 15 && frac <= 22) {
            double v = (double)mant;
            if (frac) v /= p10[frac];
            n.v = neg ? -v : v;
            if (is_int) n.exact = true;
            return n;
        }
    }
    char tmp[48];
    const size_t tl = (size_t)(t.e - t.b);
    if (tl < sizeof(tmp)) {
        memcpy(tmp, t.b, tl);
        tmp[tl] = 0;
        n.v = strtod(tmp, nullptr);  // glibc: correctly rounded, like float()
    } else {
        std::string big(t.b, t.e);
        n.v = strtod(big.c_str(), nullptr);
    }
    if (is_int) {
        const char *d = t.b + (*t.b == '-');
        const size_t nd = (size_t)(t.e - d);
        if (nd <= 15) n.exact = true;
        else if (nd > 16) n.exact = false;
        else n.exact = std::fabs(n.v) <= 9007199254740992.0 && (nd < 16 || memcmp(d, "9007199254740992", 16) <= 0);
    }
    return n;
}

void append_number(std::string &out, Span t) {
    const Num n = classify_number(t);
    if (n.is_int) {
        if (t.e - t.b == 2 && t.b[0] == '-' && t.b[1] == '0') out += '0';  // int("-0") == 0
        else out.append(t.b, t.e);
    } else {
        append_py_float(out, n.v);
    }
}

void append_utf8(std::string &out, uint32_t cp) {
    if (cp < 0x80) out += (char)cp;
    else if (cp < 0x800) { out += (char)(0xC0 | (cp >> 6)); out += (char)(0x80 | (cp & 0x3F)); }
    else if (cp < 0x10000) { out += (char)(0xE0 | (cp >> 12)); out += (char)(0x80 | ((cp >> 6) & 0x3F)); out += (char)(0x80 | (cp & 0x3F)); }
    else { out += (char)(0xF0 | (cp >> 18)); out += (char)(0x80 | ((cp >> 12) & 0x3F)); out += (char)(0x80 | ((cp >> 6) & 0x3F)); out += (char)(0x80 | (cp & 0x3F)); }
}

// json.dumps(..., ensure_ascii=False) escaping of one decoded code point
void append_escaped_cp(std::string &out, uint32_t cp) {
    switch (cp) {
        case '"': out += "\\\""; return;
        case '\\': out += "\\\\"; return;
        case '\n': out += "\\n"; return;
        case '\r': out += "\\r"; return;
        case '\t': out += "\\t"; return;
        case '\b': out += "\\b"; return;
        case '\f': out += "\\f"; return;
        default: break;
    }
    if (cp < 0x20) {
        char b[8];
        snprintf(b, sizeof(b), "\\u%04x", cp);
        out += b;
    } else {
        append_utf8(out, cp);
    }
}

int hex4(const char *p) {
    int v = 0;
    for (int i = 0; i < 4; ++i) {
        const char c = p[i];
        int d;
        if (c >= '0' && c <= '9') d = c - '0';
        else if (c >= 'a' && c <= 'f') d = c - 'a' + 10;
        else if (c >= 'A' && c <= 'F') d = c - 'A' + 10;
        else return -1;
        v = v * 16 + d;
    }
    return v;
}

// Duplicate-key detection.  json.loads keeps the LAST value of a repeated key at the position of the
// FIRST, which a streaming re-emitter cannot reproduce, so any object with a repeated key makes the cell
// irregular (an escaped key spelling could alias another key, so it is treated the same way).
struct KeySet {
    Span small[12];
    int n = 0;
    std::vector<Span> more;
    static bool has_escape(Span s) { return memchr(s.b, '\\', (size_t)(s.e - s.b)) != nullptr; }
    bool add_is_dup(Span k) {  // true -> irregular
        if (has_escape(k)) return true;
        const size_t len = (size_t)(k.e - k.b);
        for (int i = 0; i < n && i < 12; ++i)
            if ((size_t)(small[i].e - small[i].b) == len && !memcmp(small[i].b, k.b, len)) return true;
        for (const Span &o : more)
            if ((size_t)(o.e - o.b) == len && !memcmp(o.b, k.b, len)) return true;
        if (n < 12) small[n] = k; else more.push_back(k);
        ++n;
        return false;
    }
    template <class P>
    void add(P &ps, Span k) { if (add_is_dup(k)) ps.irregular(); }
};

// ---------------------------------------------------------------------------------------------------
// the parser: Python's json grammar (strict=True), streaming, no DOM
// ---------------------------------------------------------------------------------------------------
struct Parser {
    const char *p, *end;
    int depth = 0;

    void ws() { while (p < end && is_ws(*p)) ++p; }
    [[noreturn]] void bad() { throw Fail{1}; }
    [[noreturn]] void irregular() { throw Fail{2}; }

    // scans a string token at p ('"' ... '"'); returns the raw span between the quotes
    Span string_token() {
        if (p >= end || *p != '"') bad();
        const char *b = ++p;
        while (true) {
            if (p >= end) bad();
            const unsigned char c = (unsigned char)*p;
            if (c == '"') break;
            if (c < 0x20) bad();  // strict: raw control characters are invalid
            if (c == '\\') {
                if (p + 1 >= end) bad();
                const char esc = p[1];
                if (esc == 'u') {
                    if (p + 6 > end) bad();
                    const int cp = hex4(p + 2);
                    if (cp < 0) bad();
                    p += 6;
                    if (cp >= 0xD800 && cp <= 0xDBFF) {
                        // a high surrogate must be completed by \uDC00-\uDFFF; a lone one survives json.dumps
                        // but makes to_csv raise UnicodeEncodeError -> leave such cells to the Python path
                        const int lo = (p + 6 <= end && p[0] == '\\' && p[1] == 'u') ? hex4(p + 2) : -1;
                        if (lo >= 0xDC00 && lo <= 0xDFFF) p += 6;
                        else irregular();
                    } else if (cp >= 0xDC00 && cp <= 0xDFFF) {
                        irregular();
                    }
                } else if (strchr("\"\\/bfnrt", esc) && esc != 0) {
                    p += 2;
                } else {
                    bad();
                }
            } else {
                ++p;
            }
        }
        Span s{b, p};
        ++p;
        return s;
    }

    // decoded + re-escaped form of a raw string span (between the quotes), with quotes
    void emit_string(std::string &out, Span s) {
        out += '"';
        const char *q = s.b;
        while (q < s.e) {
            const unsigned char c = (unsigned char)*q;
            if (c == '\\') {
                const char esc = q[1];
                if (esc == 'u') {
                    uint32_t cp = (uint32_t)hex4(q + 2);
                    q += 6;
                    if (cp >= 0xD800 && cp <= 0xDBFF) {
                        if (q + 6 <= s.e && q[0] == '\\' && q[1] == 'u') {
                            const int lo = hex4(q + 2);
                            if (lo >= 0xDC00 && lo <= 0xDFFF) {
                                cp = 0x10000 + ((cp - 0xD800) << 10) + ((uint32_t)lo - 0xDC00);
                                q += 6;
                            } else {
                                irregular();  // lone surrogate: to_csv would raise UnicodeEncodeError
                            }
                        } else {
                            irregular();
                        }
                    } else if (cp >= 0xDC00 && cp <= 0xDFFF) {
                        irregular();
                    }
                    append_escaped_cp(out, cp);
                } else {
                    uint32_t cp;
                    switch (esc) {
                        case 'b': cp = '\b'; break;
                        case 'f': cp = '\f'; break;
                        case 'n': cp = '\n'; break;
                        case 'r': cp = '\r'; break;
                        case 't': cp = '\t'; break;
                        default: cp = (unsigned char)esc; break;  // " \ /
                    }
                    q += 2;
                    append_escaped_cp(out, cp);
                }
            } else if (c == 0x7f || c >= 0x20) {
                // raw byte of a UTF-8 sequence (or ASCII): " and \ cannot occur raw here
                out += (char)c;
                ++q;
            } else {
                bad();
            }
        }
        out += '"';
    }

    // raw-span equality with an ASCII literal (a key written with escapes compares unequal -> such
    // a cell is sent to the Python path by the callers' `has_escape` check)
    static bool span_is(Span s, const char *lit) {
        const size_t n = strlen(lit);
        return (size_t)(s.e - s.b) == n && !memcmp(s.b, lit, n);
    }
    static bool has_escape(Span s) { return memchr(s.b, '\\', (size_t)(s.e - s.b)) != nullptr; }

    Span number_token() {
        const char *b = p;
        if (p < end && *p == '-') ++p;
        if (p < end && *p == 'I') {  // -Infinity / Infinity
            if (end - p >= 8 && !memcmp(p, "Infinity", 8)) { p += 8; return Span{b, p}; }
            bad();
        }
        if (p >= end) bad();
        if (*p == '0') ++p;
        else if (*p >= '1' && *p <= '9') { while (p < end && *p >= '0' && *p <= '9') ++p; }
        else bad();
        if (p + 1 < end && *p == '.' && p[1] >= '0' && p[1] <= '9') {
            ++p;
            while (p < end && *p >= '0' && *p <= '9') ++p;
        }
        if (p < end && (*p == 'e' || *p == 'E')) {
            const char *q = p + 1;
            if (q < end && (*q == '+' || *q == '-')) ++q;
            if (q < end && *q >= '0' && *q <= '9') {
                while (q < end && *q >= '0' && *q <= '9') ++q;
                p = q;
            }
        }
        return Span{b, p};
    }

    // skips one value, optionally emitting its canonical form
    void value(std::string *out) {
        ws();
        if (p >= end) bad();
        const char c = *p;
        if (c == '{') {
            if (++depth > 256) irregular();
            ++p;
            if (out) *out += '{';
            ws();
            if (p < end && *p == '}') { ++p; if (out) *out += '}'; --depth; return; }
            bool first = true;
            KeySet ks;
            while (true) {
                ws();
                Span k = string_token();
                if (ks.add_is_dup(k)) irregular();
                if (out) { if (!first) *out += ", "; emit_string(*out, k); *out += ": "; }
                first = false;
                ws();
                if (p >= end || *p != ':') bad();
                ++p;
                value(out);
                ws();
                if (p < end && *p == ',') { ++p; continue; }
                if (p < end && *p == '}') { ++p; break; }
                bad();
            }
            if (out) *out += '}';
            --depth;
        } else if (c == '[') {
            if (++depth > 256) irregular();
            ++p;
            if (out) *out += '[';
            ws();
            if (p < end && *p == ']') { ++p; if (out) *out += ']'; --depth; return; }
            bool first = true;
            while (true) {
                if (out && !first) *out += ", ";
                first = false;
                value(out);
                ws();
                if (p < end && *p == ',') { ++p; continue; }
                if (p < end && *p == ']') { ++p; break; }
                bad();
            }
            if (out) *out += ']';
            --depth;
        } else if (c == '"') {
            Span s = string_token();
            if (out) emit_string(*out, s);
        } else if (c == 't') {
            if (end - p >= 4 && !memcmp(p, "true", 4)) { p += 4; if (out) *out += "true"; } else bad();
        } else if (c == 'f') {
            if (end - p >= 5 && !memcmp(p, "false", 5)) { p += 5; if (out) *out += "false"; } else bad();
        } else if (c == 'n') {
            if (end - p >= 4 && !memcmp(p, "null", 4)) { p += 4; if (out) *out += "null"; } else bad();
        } else if (c == 'N') {
            if (end - p >= 3 && !memcmp(p, "NaN", 3)) { p += 3; if (out) *out += "NaN"; } else bad();
        } else if (c == '-' || c == 'I' || (c >= '0' && c <= '9')) {
            Span t = number_token();
            if (t.e - t.b > 4000) irregular();   // int() refuses more than 4300 digits (ValueError, not a decode error)
            if (out) append_number(*out, t);
        } else {
            bad();
        }
    }

    char peek() { ws(); return p < end ? *p : 0; }
};

// what a value token looks like from the outside
enum Kind { K_OBJECT, K_ARRAY, K_STRING, K_NUMBER, K_TRUE, K_FALSE, K_NULL };
Kind kind_of(char c) {
    switch (c) {
        case '{': return K_OBJECT;
        case '[': return K_ARRAY;
        case '"': return K_STRING;
        case 't': return K_TRUE;
        case 'f': return K_FALSE;
        case 'n': return K_NULL;
        default: return K_NUMBER;  // digits, '-', NaN, Infinity (validated by the parser)
    }
}

struct PointTok {
    Span x, y;
};

// Walks a ptList array (p at '['): collects the x/y number tokens of every valid point.
// Regular means: every element that is an object with both keys has NUMBER values there.
void walk_ptlist(Parser &ps, std::vector<PointTok> &pts) {
    ++ps.p;
    if (ps.peek() == ']') { ++ps.p; return; }
    while (true) {
        const char c = ps.peek();
        if (c == '{') {
            ++ps.p;
            KeySet ks;
            bool hx = false, hy = false, okx = false, oky = false;
            PointTok t{};
            if (ps.peek() == '}') {
                ++ps.p;
            } else {
                while (true) {
                    ps.ws();
                    const Span k = ps.string_token();
                    ks.add(ps, k);
                    ps.ws();
                    if (ps.p >= ps.end || *ps.p != ':') ps.bad();
                    ++ps.p;
                    const bool isx = Parser::span_is(k, "x"), isy = Parser::span_is(k, "y");
                    if (isx || isy) {
                        const char v = ps.peek();
                        const bool numeric = (kind_of(v) == K_NUMBER);
                        const char *b = ps.p;
                        ps.value(nullptr);
                        if (isx) { hx = true; okx = numeric; t.x = Span{b, ps.p}; }
                        else { hy = true; oky = numeric; t.y = Span{b, ps.p}; }
                    } else {
                        ps.value(nullptr);
                    }
                    const char d = ps.peek();
                    if (d == ',') { ++ps.p; continue; }
                    if (d == '}') { ++ps.p; break; }
                    ps.bad();
                }
            }
            if (hx && hy) {
                if (!okx || !oky) ps.irregular();  // None / str / bool / container coordinate -> Python path
                pts.push_back(t);
            }
        } else {
            ps.value(nullptr);  // non-dict element: not a valid point (:253)
        }
        const char d = ps.peek();
        if (d == ',') { ++ps.p; continue; }
        if (d == ']') { ++ps.p; break; }
        ps.bad();
    }
}

// ---------------------------------------------------------------------------------------------------
// replace step: one traversal that either collects points (scan) or writes the rewritten text (emit)
// ---------------------------------------------------------------------------------------------------
struct WH {
    uint8_t kind = 0;  // 0 absent / null, 1 int, 2 float, 3 something else (host re-reads it with CPython json)
    double v = 0;
};

struct CellSink {
    // scan mode
    std::vector<double> *xy = nullptr;
    std::vector<int32_t> *npts = nullptr;
    // emit mode
    std::string *out = nullptr;
    const int32_t *arg4 = nullptr;  // arg indices of this cell's boxes
    int box = 0;                    // boxes seen so far in this cell
    bool big_int = false;           // scan: some coordinate is an int beyond 2^25 (the IoU step's products leave f64's exact range)
    WH w, h;
};

void corner_points(Parser &ps, CellSink &sk, const std::vector<PointTok> &pts) {
    if (sk.xy) {  // scan: append the points
        for (const PointTok &t : pts) {
            const Num nx = classify_number(t.x), ny = classify_number(t.y);
            if ((nx.is_int && !nx.exact) || (ny.is_int && !ny.exact)) ps.irregular();  // exact big-int compare
            if ((nx.is_int && std::fabs(nx.v) > 33554432.0) || (ny.is_int && std::fabs(ny.v) > 33554432.0)) sk.big_int = true;
            sk.xy->push_back(nx.v);
            sk.xy->push_back(ny.v);
        }
        sk.npts->push_back((int32_t)pts.size());
    } else {  // emit: the two corner points, original tokens re-printed (:256-260)
        std::string &o = *sk.out;
        if (pts.empty()) {
            o += "[{\"x\": null, \"y\": null}, {\"x\": null, \"y\": null}]";
        } else {
            const int32_t *a = sk.arg4 + 4 * sk.box;
            const int n = (int)pts.size();
            for (int i = 0; i < 4; ++i)
                if (a[i] < 0 || a[i] >= n) ps.irregular();
            o += "[{\"x\": "; append_number(o, pts[a[0]].x);
            o += ", \"y\": "; append_number(o, pts[a[1]].y);
            o += "}, {\"x\": "; append_number(o, pts[a[2]].x);
            o += ", \"y\": "; append_number(o, pts[a[3]].y);
            o += "}]";
        }
    }
    ++sk.box;
}

// polygon object (p at '{'): emits its members, replacing / appending ptList
void walk_polygon(Parser &ps, CellSink &sk) {
    std::string *out = sk.out;
    ++ps.p;
    if (out) *out += '{';
    KeySet ks;
    bool first = true, seen = false;
    std::vector<PointTok> pts;
    if (ps.peek() == '}') {
        ++ps.p;
    } else {
        while (true) {
            ps.ws();
            const Span k = ps.string_token();
            ks.add(ps, k);
            if (out) { if (!first) *out += ", "; ps.emit_string(*out, k); *out += ": "; }
            first = false;
            ps.ws();
            if (ps.p >= ps.end || *ps.p != ':') ps.bad();
            ++ps.p;
            if (Parser::span_is(k, "ptList")) {
                if (kind_of(ps.peek()) != K_ARRAY) ps.irregular();  // None / str / dict / number ptList
                seen = true;
                pts.clear();
                walk_ptlist(ps, pts);
                corner_points(ps, sk, pts);
            } else {
                ps.value(out);
            }
            const char d = ps.peek();
            if (d == ',') { ++ps.p; continue; }
            if (d == '}') { ++ps.p; break; }
            ps.bad();
        }
    }
    if (!seen) {  // obj.get("polygon", {}).get("ptList", []) -> [] ; ptList is appended (:276)
        pts.clear();
        if (out) { if (!first) *out += ", "; *out += "\"ptList\": "; }
        corner_points(ps, sk, pts);
    }
    if (out) *out += '}';
}

// one element of "objects" that is an object (p at '{')
void walk_object(Parser &ps, CellSink &sk) {
    std::string *out = sk.out;
    ++ps.p;
    if (out) *out += '{';
    KeySet ks;
    bool first = true, seen = false;
    if (ps.peek() == '}') {
        ++ps.p;
    } else {
        while (true) {
            ps.ws();
            const Span k = ps.string_token();
            ks.add(ps, k);
            if (out) { if (!first) *out += ", "; ps.emit_string(*out, k); *out += ": "; }
            first = false;
            ps.ws();
            if (ps.p >= ps.end || *ps.p != ':') ps.bad();
            ++ps.p;
            if (Parser::span_is(k, "polygon")) {
                if (kind_of(ps.peek()) != K_OBJECT) ps.irregular();  // None / list ... -> AttributeError in Python
                seen = true;
                walk_polygon(ps, sk);
            } else {
                ps.value(out);
            }
            const char d = ps.peek();
            if (d == ',') { ++ps.p; continue; }
            if (d == '}') { ++ps.p; break; }
            ps.bad();
        }
    }
    if (!seen) {  // :274-276 — a polygon dict is added at the end
        std::vector<PointTok> none;
        if (out) { if (!first) *out += ", "; *out += "\"polygon\": {\"ptList\": "; }
        corner_points(ps, sk, none);
        if (out) *out += '}';
    }
    if (out) *out += '}';
}

void read_wh(Parser &ps, WH &dst, std::string *out) {
    const char c = ps.peek();
    const Kind kd = kind_of(c);
    const char *b = ps.p;
    ps.value(out);
    if (kd == K_NULL) { dst.kind = 0; return; }
    if (kd != K_NUMBER) { dst.kind = 3; return; }
    const Num n = classify_number(Span{b, ps.p});
    if (n.is_int && !n.exact) { dst.kind = 3; return; }
    dst.kind = n.is_int ? 1 : 2;
    dst.v = n.v;
}

// whole cell.  Throws Fail{1} (undecodable) or Fail{2} (irregular).
void walk_cell(Span cell, CellSink &sk) {
    Parser ps{cell.b, cell.e};
    std::string *out = sk.out;
    ps.ws();
    if (ps.p >= ps.end) ps.bad();
    if (*ps.p != '{') {  // a list / scalar document: data.get raises AttributeError -> Python path decides
        ps.value(nullptr);
        ps.ws();
        if (ps.p != ps.end) ps.bad();
        ps.irregular();
    }
    ++ps.p;
    if (out) *out += '{';
    KeySet ks;
    bool first = true, seen = false;
    if (ps.peek() == '}') {
        ++ps.p;
    } else {
        while (true) {
            ps.ws();
            const Span k = ps.string_token();
            ks.add(ps, k);
            if (out) { if (!first) *out += ", "; ps.emit_string(*out, k); *out += ": "; }
            first = false;
            ps.ws();
            if (ps.p >= ps.end || *ps.p != ':') ps.bad();
            ++ps.p;
            if (Parser::span_is(k, "objects")) {
                if (kind_of(ps.peek()) != K_ARRAY) ps.irregular();  // null / dict / str / number "objects"
                seen = true;
                ++ps.p;
                if (out) *out += '[';
                bool efirst = true;
                if (ps.peek() == ']') {
                    ++ps.p;
                } else {
                    while (true) {
                        if (ps.peek() == '{') {
                            if (out && !efirst) *out += ", ";
                            efirst = false;
                            walk_object(ps, sk);
                        } else {
                            ps.value(nullptr);  // non-dict objects are dropped (:270)
                        }
                        const char d = ps.peek();
                        if (d == ',') { ++ps.p; continue; }
                        if (d == ']') { ++ps.p; break; }
                        ps.bad();
                    }
                }
                if (out) *out += ']';
            } else if (Parser::span_is(k, "width")) {
                read_wh(ps, sk.w, out);
            } else if (Parser::span_is(k, "height")) {
                read_wh(ps, sk.h, out);
            } else {
                ps.value(out);
            }
            const char d = ps.peek();
            if (d == ',') { ++ps.p; continue; }
            if (d == '}') { ++ps.p; break; }
            ps.bad();
        }
    }
    ps.ws();
    if (ps.p != ps.end) ps.bad();  // "Extra data"
    if (!seen && out) { if (!first) *out += ", "; *out += "\"objects\": []"; }  // data["objects"] = [] (:278)
    if (out) *out += '}';
}

// ---------------------------------------------------------------------------------------------------
// IoU step: two-point boxes of one cell with the reference's prefix-on-exception rule (:341-366)
// ---------------------------------------------------------------------------------------------------
enum BoxResult { BR_OK, BR_STOP, BR_IRREGULAR };  // STOP = the reference raises inside its try: keep the prefix

// p at the '{' of a ptList element; consumes it; true when it has both "x" and "y"
bool point_tokens(Parser &ps, PointTok &t) {
    ++ps.p;
    KeySet ks;
    bool hx = false, hy = false;
    if (ps.peek() == '}') { ++ps.p; return false; }
    while (true) {
        ps.ws();
        const Span k = ps.string_token();
        ks.add(ps, k);
        ps.ws();
        ++ps.p;  // ':' (the document was validated before)
        ps.ws();
        const char *b = ps.p;
        ps.value(nullptr);
        if (Parser::span_is(k, "x")) { hx = true; t.x = Span{b, ps.p}; }
        if (Parser::span_is(k, "y")) { hy = true; t.y = Span{b, ps.p}; }
        if (ps.peek() == ',') { ++ps.p; continue; }
        ++ps.p;  // '}'
        return hx && hy;
    }
}

// p at the value of "ptList"; on BR_OK `is_box` tells whether v[4] holds a box
BoxResult ptlist_box(Parser &ps, double v[4], bool &is_box) {
    is_box = false;
    const Kind lk = kind_of(ps.peek());
    if (lk == K_NULL || lk == K_NUMBER || lk == K_TRUE || lk == K_FALSE) return BR_STOP;  // len() raises TypeError
    if (lk != K_ARRAY) return BR_IRREGULAR;  // str / dict have a len(): leave them to Python
    ++ps.p;
    int n_elems = 0;
    bool both = true;
    PointTok pt[2] = {};
    if (ps.peek() == ']') {
        ++ps.p;
    } else {
        while (true) {
            const bool dict = (ps.peek() == '{');
            PointTok t{};
            bool has_xy = false;
            if (dict) has_xy = point_tokens(ps, t);
            else ps.value(nullptr);
            if (n_elems < 2) { pt[n_elems] = t; both = both && dict && has_xy; }
            ++n_elems;
            if (ps.peek() == ',') { ++ps.p; continue; }
            ++ps.p;  // ']'
            break;
        }
    }
    if (n_elems != 2 || !both) return BR_OK;  // :352-358 -> continue
    const Span toks[4] = {pt[0].x, pt[0].y, pt[1].x, pt[1].y};
    bool any_null = false;
    for (int i = 0; i < 4; ++i) {
        const Kind kd = kind_of(*toks[i].b);
        if (kd == K_NULL) { any_null = true; continue; }
        if (kd != K_NUMBER) return BR_IRREGULAR;  // str / bool / container coordinates
        const Num n = classify_number(toks[i]);
        if (n.is_int && std::fabs(n.v) > 33554432.0) return BR_IRREGULAR;  // > 2^25: exact big-int arithmetic
        v[i] = n.v;
    }
    if (any_null) {
        // min(None, number) and min(None, None) both raise TypeError, but only if no earlier argument pair of
        // the four min/max calls is fine first: every one of them involves x or y of BOTH points, and the
        // first call, min(p1["x"], p2["x"]), raises iff one of the two x is None; otherwise a y is None and the
        // second call raises.  Either way nothing was appended: prefix.
        return BR_STOP;
    }
    is_box = true;
    return BR_OK;
}

// p at the '{' of an "objects" element
BoxResult object_box(Parser &ps, double v[4], bool &is_box) {
    is_box = false;
    ++ps.p;
    KeySet ks;
    BoxResult res = BR_OK;
    if (ps.peek() == '}') { ++ps.p; return BR_OK; }
    while (true) {
        ps.ws();
        const Span k = ps.string_token();
        ks.add(ps, k);
        ps.ws();
        ++ps.p;
        if (Parser::span_is(k, "polygon")) {
            const Kind pk = kind_of(ps.peek());
            if (pk != K_OBJECT) return BR_STOP;  // None / list / str / number .get -> AttributeError
            ++ps.p;
            KeySet pks;
            if (ps.peek() == '}') {
                ++ps.p;
            } else {
                while (true) {
                    ps.ws();
                    const Span pkey = ps.string_token();
                    pks.add(ps, pkey);
                    ps.ws();
                    ++ps.p;
                    if (Parser::span_is(pkey, "ptList")) {
                        res = ptlist_box(ps, v, is_box);
                        if (res != BR_OK) return res;
                    } else {
                        ps.value(nullptr);
                    }
                    if (ps.peek() == ',') { ++ps.p; continue; }
                    ++ps.p;
                    break;
                }
            }
        } else {
            ps.value(nullptr);
        }
        if (ps.peek() == ',') { ++ps.p; continue; }
        ++ps.p;
        return BR_OK;
    }
}

// returns false when the cell must go to the Python path
bool boxes_of_cell(Span cell, std::vector<double> &box4, int32_t &count) {
    count = 0;
    const size_t mark = box4.size();
    Parser ps{cell.b, cell.e};
    try {  // validate the whole document first: an undecodable cell yields no boxes at all
        ps.value(nullptr);
        ps.ws();
        if (ps.p != ps.end) ps.bad();
    } catch (Fail f) {
        return f.code == 1;  // JSONDecodeError is swallowed by the blanket except -> []
    }
    ps = Parser{cell.b, cell.e};
    try {
        if (ps.peek() != '{') return true;  // list / scalar document: .get raises inside the try -> []
        ++ps.p;
        KeySet ks;
        if (ps.peek() == '}') { return true; }
        while (true) {
            ps.ws();
            const Span k = ps.string_token();
            ks.add(ps, k);
            ps.ws();
            ++ps.p;
            if (Parser::span_is(k, "objects")) {
                const Kind kd = kind_of(ps.peek());
                if (kd == K_NULL || kd == K_NUMBER || kd == K_TRUE || kd == K_FALSE) return true;  // iteration raises -> []
                if (kd != K_ARRAY) { box4.resize(mark); count = 0; return false; }  // dict / str iterate: Python path
                ++ps.p;
                if (ps.peek() == ']') return true;
                while (true) {
                    if (ps.peek() == '{') {
                        double v[4];
                        bool is_box = false;
                        const BoxResult r = object_box(ps, v, is_box);
                        if (r == BR_STOP) return true;  // keep the prefix collected so far
                        if (r == BR_IRREGULAR) { box4.resize(mark); count = 0; return false; }
                        if (is_box) { box4.insert(box4.end(), v, v + 4); ++count; }
                    } else {
                        ps.value(nullptr);
                    }
                    if (ps.peek() == ',') { ++ps.p; continue; }
                    return true;
                }
            }
            ps.value(nullptr);
            if (ps.peek() == ',') { ++ps.p; continue; }
            return true;  // no "objects" member: []
        }
    } catch (Fail f) {
        box4.resize(mark);
        count = 0;
        return false;
    }
}

template <class F>
void parallel_cells(int64_t n, int n_threads, F fn) {
    if (n_threads <= 0) n_threads = (int)std::min<unsigned>(16u, std::max(1u, std::thread::hardware_concurrency()));
    n_threads = (int)std::min<int64_t>(n_threads, std::max<int64_t>(1, n / 256));
    if (n_threads <= 1) { fn(0, (int64_t)0, n); return; }
    std::vector<std::thread> th;
    for (int t = 0; t < n_threads; ++t) {
        const int64_t lo = n * t / n_threads, hi = n * (t + 1) / n_threads;
        th.emplace_back([=] { fn(t, lo, hi); });
    }
    for (auto &x : th) x.join();
}

#include "host_json_fast.h"

// where the cells of a scan live: one flat buffer + offsets, or one pointer + length per cell
struct CellSrc {
    const uint8_t *text = nullptr;
    const int64_t *off = nullptr;
    const uint8_t *const *ptr = nullptr;
    const int64_t *len = nullptr;
    inline Span get(int64_t i) const {
        if (ptr) return Span{(const char *)ptr[i], (const char *)ptr[i] + len[i]};
        return Span{(const char *)text + off[i], (const char *)text + off[i + 1]};
    }
    size_t bytes(int64_t lo, int64_t hi) const {
        if (hi <= lo) return 0;
        if (!ptr) return (size_t)(off[hi] - off[lo]);
        size_t s = 0;
        for (int64_t i = lo; i < hi; ++i) s += (size_t)len[i];
        return s;
    }
};

// as parallel_cells, but a bad_alloc inside a worker is reported instead of ending the process
template <class F>
bool parallel_cells_safe(int64_t n, int n_threads, F fn) {
    int failed = 0;
    parallel_cells(n, n_threads, [&](int t, int64_t lo, int64_t hi) {
        try {
            fn(t, lo, hi);
        } catch (const std::bad_alloc &) {
            __atomic_store_n(&failed, 1, __ATOMIC_RELAXED);
        }
    });
    return failed == 0;
}

template <class F>
bool parallel_index_safe(int n, F fn) {   // fn(k) for k in [0, n), one thread each
    int failed = 0;
    std::vector<std::thread> th;
    for (int k = 1; k < n; ++k)
        th.emplace_back([&, k] {
            try { fn(k); } catch (const std::bad_alloc &) { __atomic_store_n(&failed, 1, __ATOMIC_RELAXED); }
        });
    if (n > 0) {
        try { fn(0); } catch (const std::bad_alloc &) { __atomic_store_n(&failed, 1, __ATOMIC_RELAXED); }
    }
    for (auto &x : th) x.join();
    return failed == 0;
}

}  // namespace

// ===================================================================================================
// C ABI
// ===================================================================================================
struct dyd_scan {
    int64_t n_cells = 0;
    CellSrc src;                        // polygon scan: where the cells are (kept for the exact walker's pass 2)
    std::vector<double> xy;
    std::vector<int32_t> pt_off;        // [n_boxes + 1]
    std::vector<int32_t> cell_box_off;  // [n_cells + 1]
    std::vector<uint8_t> status;        // CellStatus per cell
    std::vector<uint8_t> w_kind, h_kind;
    std::vector<double> w_val, h_val;
    std::vector<uint8_t> sel;           // labelled scan: box carries the row's label
    std::vector<uint8_t> iou_host;      // polygon scan: the cell's IoU flag needs CPython's exact int arithmetic (see dyd_scan_iou_host)
    // emit output
    std::string text;
    std::vector<int64_t> text_off;
    // polygon scan (host_json_fast.h): per-thread parts, gathered SoA arrays, gathered text
    bool fast = false;
    std::vector<std::unique_ptr<FastPart>> parts;
    Raw<double> f_xy;
    Raw<int32_t> f_pt_off;
    Raw<char> f_text;
    // replace -> IoU pipeline (dyd_json_replace_iou): per-cell flag, totals, phase times (the slowest part's)
    bool pipelined = false;
    std::vector<uint8_t> high;
    int64_t tot_boxes = 0, tot_points = 0;
    double t_scan = 0, t_device = 0, t_emit = 0;
    ~dyd_scan();                        // parks the parts' text blocks in the host pool
};

// ---------------------------------------------------------------------------------------------------
// polygon scan / emit over per-thread parts (host_json_fast.h)
// ---------------------------------------------------------------------------------------------------
namespace {

int default_threads() {   // the process's CPU share: cgroup quota when there is one (a 16-CPU slice of a 256-thread host)
    static int cached = 0;
    if (cached) return cached;
    unsigned hw = std::thread::hardware_concurrency();
    if (hw == 0) hw = 1;
    double quota = 0;
    if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
        char a[64] = {0};
        double per = 0;
        if (fscanf(f, "%63s %lf", a, &per) == 2 && strcmp(a, "max") != 0 && per > 0) quota = atof(a) / per;
        fclose(f);
    }
    int n = (int)hw;
    if (quota >= 1.0 && quota < n) n = (int)(quota + 0.5);
    if (const char *e = getenv("DYD_HOST_THREADS")) { const int v = atoi(e); if (v > 0) n = v; }
    cached = std::max(1, std::min(n, 64));
    return cached;
}

inline bool use_fast_lane() {
    const char *env = getenv("DYD_JSON_FAST");
    return !(env && env[0] == '0');
}

// Blocks of emitted text outlive the pass that wrote them (the caller builds str objects from them) and are then 2 GB of
// page tables to tear down (0.17 s per million rows, "release") and as many first-touch faults for the next pass to take again:
// a freed handle parks them here instead, up to DYD_HOST_POOL_MB (default 3072) in total, and the next pass writes into warm
// memory.  dyd_host_pool_trim() gives everything back.
struct HostPool {
    std::mutex m;
    std::vector<std::pair<char *, size_t>> blocks;
    size_t total = 0;
    static size_t limit() {
        static size_t cached = 0;
        if (!cached) {
            size_t mb = 3072;
            if (const char *e = getenv("DYD_HOST_POOL_MB")) { const long v = atol(e); if (v >= 0) mb = (size_t)v; }
            cached = (mb << 20) + 1;
        }
        return cached - 1;
    }
    // the smallest parked block of at least `need` bytes, else the largest one (the caller grows it), else nothing
    char *take(size_t need, size_t *cap) {
        std::lock_guard<std::mutex> lk(m);
        int best = -1;
        for (int i = 0; i < (int)blocks.size(); ++i) {
            if (best < 0) { best = i; continue; }
            const bool fi = blocks[(size_t)i].second >= need, fb = blocks[(size_t)best].second >= need;
            if ((fi && !fb) || (fi && fb && blocks[(size_t)i].second < blocks[(size_t)best].second) ||
                (!fi && !fb && blocks[(size_t)i].second > blocks[(size_t)best].second))
                best = i;
        }
        if (best < 0) { *cap = 0; return nullptr; }
        char *p = blocks[(size_t)best].first;
        *cap = blocks[(size_t)best].second;
        total -= *cap;
        blocks.erase(blocks.begin() + best);
        return p;
    }
    void give(char *p, size_t cap) {
        if (!p) return;
        {
            std::lock_guard<std::mutex> lk(m);
            if (cap >= ((size_t)1 << 20) && total + cap <= limit() && blocks.size() < 256) {
                blocks.emplace_back(p, cap);
                total += cap;
                return;
            }
        }
        free(p);
    }
    void trim() {
        std::lock_guard<std::mutex> lk(m);
        for (auto &b : blocks) free(b.first);
        blocks.clear();
        total = 0;
    }
};
HostPool &host_pool() {
    static HostPool p;
    return p;
}

// pass 1 over the part's cell range: fills A and the per-cell arrays of h
void scan_part(dyd_scan *h, const CellSrc &src, const uint8_t *missing, FastPart &A, bool use_fast) {
    std::string tmp;
    std::vector<double> sxy;
    std::vector<int32_t> snp;
    const int64_t lo = A.lo, hi = A.hi;
    const size_t bytes = src.bytes(lo, hi);
    A.seg_off.n = A.lane.n = A.cell_boxes.n = A.seg.n = A.xy.n = A.isint.n = A.npts.n = A.hole.n = 0;   // a part object serves chunk after chunk
    A.seg_off.need((size_t)(hi - lo) + 1);
    A.lane.need((size_t)(hi - lo));
    A.cell_boxes.need((size_t)(hi - lo));
    A.seg.need(bytes / 3 + 64);          // the usual share of a cell that is not point text
    A.xy.need(bytes / 12 + 64);          // ~ one point (2 doubles) per 35 bytes of text
    for (int64_t i = lo; i < hi; ++i) {
        A.seg_off.push((int64_t)A.seg.n);
        if (missing && missing[i]) { h->status[(size_t)i] = CELL_MISSING; A.lane.push(0); A.cell_boxes.push(0); continue; }
        const size_t m_xy = A.xy.n, m_ii = A.isint.n, m_np = A.npts.n, m_ho = A.hole.n, m_sg = A.seg.n;
        const Span cell = src.get(i);
        if (use_fast) {
            FastCell fc(cell.b, cell.e, A, tmp);
            if (fc.cell()) {
                A.lane.push(1);
                A.cell_boxes.push(fc.box);
                h->iou_host[(size_t)i] = fc.big_int ? 1 : 0;
                h->w_kind[(size_t)i] = fc.w.kind; h->w_val[(size_t)i] = fc.w.v;
                h->h_kind[(size_t)i] = fc.h.kind; h->h_val[(size_t)i] = fc.h.v;
                continue;
            }
            A.xy.n = m_xy; A.isint.n = m_ii; A.npts.n = m_np; A.hole.n = m_ho; A.seg.n = m_sg;
        }
        A.lane.push(0);
        sxy.clear(); snp.clear();
        CellSink sk;
        sk.xy = &sxy; sk.npts = &snp;
        try {
            walk_cell(cell, sk);
            A.xy.put(sxy.data(), sxy.size());
            if (!sxy.empty()) {
                A.isint.need(sxy.size() / 2);
                memset(A.isint.p + A.isint.n, 0, sxy.size() / 2);
                A.isint.n += sxy.size() / 2;
            }
            A.npts.put(snp.data(), snp.size());
            for (size_t k = 0; k < snp.size(); ++k) A.hole.push(0);
            A.cell_boxes.push(sk.box);
            h->iou_host[(size_t)i] = sk.big_int ? 1 : 0;
            h->w_kind[(size_t)i] = sk.w.kind; h->w_val[(size_t)i] = sk.w.v;
            h->h_kind[(size_t)i] = sk.h.kind; h->h_val[(size_t)i] = sk.h.v;
        } catch (Fail f) {
            A.cell_boxes.push(0);
            h->status[(size_t)i] = (f.code == 1) ? CELL_UNDECODABLE : CELL_IRREGULAR;
        }
    }
    A.seg_off.push((int64_t)A.seg.n);
}

// pass 2 over the part: arg4 = K1's arg indices of the PART's boxes (local order); xy / pt_off as the part's views say
// (A.xyv: the part's first point; A.ptv[b] - A.ptv_bias: first point of local box b).  The text is APPENDED to `out`, one length
// per cell to `out_len`.  false: arg4 does not fit the scan.
bool emit_part(const dyd_scan *h, const CellSrc &src, FastPart &A, const int32_t *arg4, Raw<char> &out, Raw<int64_t> &out_len) {
    std::string tmp, slow;
    bool good = true;
    out_len.need((size_t)(A.hi - A.lo));
    out.need(A.seg.n + A.seg.n / 2 + 64);
    size_t lb = 0;   // local box index
    for (int64_t i = A.lo; i < A.hi; ++i) {
        const size_t mark = out.n;
        const size_t nb = (size_t)A.cell_boxes.p[i - A.lo];
        if (h->status[(size_t)i] != CELL_OK) { out_len.push(0); lb += nb; continue; }
        if (A.lane.p[i - A.lo]) {
            const char *sg = A.seg.p + A.seg_off.p[i - A.lo];
            const size_t sg_len = (size_t)(A.seg_off.p[i - A.lo + 1] - A.seg_off.p[i - A.lo]);
            size_t prev = 0;
            for (size_t b = lb; b < lb + nb; ++b) {
                const size_t hb = A.hole.p[b];
                out.put(sg + prev, hb - prev);
                prev = hb;
                const size_t p0 = (size_t)(A.ptv[b] - A.ptv_bias), p1 = (size_t)(A.ptv[b + 1] - A.ptv_bias);
                if (!fj_put_corners(out, A.xyv + 2 * p0, A.isint.p + p0, (int32_t)(p1 - p0), arg4 + 4 * b, tmp)) good = false;
            }
            out.put(sg + prev, sg_len - prev);
        } else {
            slow.clear();
            CellSink sk;
            sk.out = &slow;
            sk.arg4 = arg4 + 4 * lb;
            try {
                walk_cell(src.get(i), sk);
                out.put(slow.data(), slow.size());
            } catch (Fail) {
                out.n = mark;
                good = false;
            }
        }
        lb += nb;
        out_len.push((int64_t)(out.n - mark));
    }
    return good;
}

// the parts' texts -> one buffer + offsets (one thread per part)
bool gather_text(dyd_scan *h) {
    size_t total = 0;
    std::vector<size_t> base(h->parts.size() + 1, 0);
    for (size_t k = 0; k < h->parts.size(); ++k) { base[k] = total; total += h->parts[k]->out.n; }
    h->f_text.n = 0;
    h->f_text.need(total + 1);
    h->f_text.n = total;
    h->text_off.resize((size_t)h->n_cells + 1);
    h->text_off[0] = 0;
    return parallel_index_safe((int)h->parts.size(), [&](int k) {
        FastPart &A = *h->parts[(size_t)k];
        if (A.out.n) memcpy(h->f_text.p + base[(size_t)k], A.out.p, A.out.n);
        int64_t run = (int64_t)base[(size_t)k];
        for (int64_t i = A.lo; i < A.hi; ++i) { run += A.out_len.p[i - A.lo]; h->text_off[(size_t)i + 1] = run; }
        A.out.clear_free();
    });
}

void init_polygon_handle(dyd_scan *h, int64_t n_cells, int n_threads, const CellSrc &src) {
    h->n_cells = n_cells;
    h->fast = true;
    h->src = src;
    h->status.assign((size_t)n_cells, CELL_OK);
    h->w_kind.assign((size_t)n_cells, 0); h->h_kind.assign((size_t)n_cells, 0);
    h->w_val.assign((size_t)n_cells, 0.0); h->h_val.assign((size_t)n_cells, 0.0);
    h->iou_host.assign((size_t)n_cells, 0);
    h->cell_box_off.assign((size_t)n_cells + 1, 0);
    if (n_threads <= 0) n_threads = default_threads();
    n_threads = (int)std::min<int64_t>(n_threads, std::max<int64_t>(1, n_cells / 256));
    n_threads = std::min(n_threads, 64);
    for (int t = 0; t < n_threads; ++t) {
        h->parts.emplace_back(new FastPart());
        h->parts.back()->lo = n_cells * t / n_threads;
        h->parts.back()->hi = n_cells * (t + 1) / n_threads;
    }
}

}  // namespace

dyd_scan::~dyd_scan() {
    for (auto &pp : parts) {
        if (!pp) continue;
        size_t cap = 0;
        char *blk = pp->out.disown(&cap);
        host_pool().give(blk, cap);
    }
}

extern "C" {

void dyd_host_pool_trim(void) { host_pool().trim(); }

// Scan annotation cells for the replace step (processor.py:262-281).  text/cell_off: concatenated UTF-8
// cells; missing[i] != 0 marks a NaN cell.  The handle owns every output array.
// Regular cells go through the single-parse lane of host_json_fast.h; a cell that lane does not take is walked by the
// exact parser (walk_cell), which also decides between undecodable and irregular.  DYD_JSON_FAST=0 sends every cell there.
static int scan_polygons_src(const CellSrc &src, const uint8_t *missing, int64_t n_cells, int n_threads, dyd_scan **out) {
    dyd_scan *h = new (std::nothrow) dyd_scan();
    if (!h) return DYD_ERR_OOM;
    const bool use_fast = use_fast_lane();
    const bool timing = getenv("DYD_JSON_TIMING") != nullptr;
    auto T0 = std::chrono::steady_clock::now();
    try {
        init_polygon_handle(h, n_cells, n_threads, src);
        if (!parallel_index_safe((int)h->parts.size(), [&](int k) { scan_part(h, src, missing, *h->parts[(size_t)k], use_fast); }))
            throw std::bad_alloc();
        if (timing) fprintf(stderr, "scan parallel part: %.3f s\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - T0).count());
        size_t nb = 0, npnt = 0;
        for (auto &pp : h->parts) { pp->box_base = nb; pp->pt_base = npnt; nb += pp->npts.n; npnt += pp->xy.n / 2; }
        if (npnt >= (size_t)1 << 31) { delete h; return DYD_ERR_RANGE; }
        h->f_xy.need(2 * npnt + 2);
        h->f_xy.n = 2 * npnt;
        h->f_pt_off.need(nb + 1);
        h->f_pt_off.n = nb + 1;
        h->f_pt_off.p[0] = 0;
        // gather the parts into the contiguous arrays K1 takes, one thread per part
        if (!parallel_index_safe((int)h->parts.size(), [&](int k) {
                FastPart &A = *h->parts[(size_t)k];
                if (A.xy.n) memcpy(h->f_xy.p + 2 * A.pt_base, A.xy.p, A.xy.n * sizeof(double));
                int32_t run = (int32_t)A.pt_base;
                int32_t *po = h->f_pt_off.p + A.box_base + 1;
                for (size_t b = 0; b < A.npts.n; ++b) { run += A.npts.p[b]; po[b] = run; }
                int32_t *cb = h->cell_box_off.data() + 1;
                for (int64_t i = A.lo; i < A.hi; ++i) cb[i] = A.cell_boxes.p[i - A.lo];
                A.xy.clear_free();      // pass 2 reads the gathered copies
                A.npts.clear_free();
                A.xyv = h->f_xy.p + 2 * A.pt_base;
                A.ptv = h->f_pt_off.p + A.box_base;
                A.ptv_bias = (int32_t)A.pt_base;
            }))
            throw std::bad_alloc();
        for (int64_t i = 0; i < n_cells; ++i) h->cell_box_off[(size_t)i + 1] += h->cell_box_off[(size_t)i];
        if (timing) fprintf(stderr, "scan total: %.3f s\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - T0).count());
    } catch (const std::bad_alloc &) {
        delete h;
        return DYD_ERR_OOM;
    }
    *out = h;
    return DYD_OK;
}

int dyd_json_scan_polygons(const uint8_t *text, const int64_t *cell_off, const uint8_t *missing, int64_t n_cells,
                           int n_threads, dyd_scan **out) {
    if (!out || n_cells < 0 || (n_cells > 0 && !cell_off)) return DYD_ERR_INVALID;
    CellSrc src;
    src.text = text; src.off = cell_off;
    return scan_polygons_src(src, missing, n_cells, n_threads, out);
}

// the same over one (pointer, length) pair per cell — e.g. the UTF-8 views of the str objects of a DataFrame column, so that
// no cell is copied; the pointers must stay valid until the handle is freed
int dyd_json_scan_polygons_v(const uint8_t *const *cell_ptr, const int64_t *cell_len, const uint8_t *missing, int64_t n_cells,
                             int n_threads, dyd_scan **out) {
    if (!out || n_cells < 0 || (n_cells > 0 && (!cell_ptr || !cell_len))) return DYD_ERR_INVALID;
    CellSrc src;
    src.ptr = cell_ptr; src.len = cell_len;
    return scan_polygons_src(src, missing, n_cells, n_threads, out);
}

// The replace step and the IoU step in one native pass (reference processor.py:262-281 then :341-376; the processing page runs
// them back to back, ui/pages/processing.py:580-598).  The cell range is cut into one part per thread and every thread takes its
// part through all three stages on its own — single-parse scan, ONE fused K1+K2 launch on the part's arrays (dyd_bbox_iou_fused:
// staged straight from the part's buffers, no gathered copy of the points), segment emit — and frees the part's point and segment
// buffers before it returns.  What stays in the handle: per cell status / flag / width / height / iou_host, and per part the
// emitted text (dyd_scan_part), which dyd_scan_text gathers into one buffer on request.
int dyd_json_replace_iou(const uint8_t *text, const int64_t *cell_off, const uint8_t *const *cell_ptr, const int64_t *cell_len,
                         const uint8_t *missing, int64_t n_cells, int32_t min_boxes, double thr, int n_threads, dyd_scan **out) {
    if (!out || n_cells < 0) return DYD_ERR_INVALID;
    CellSrc src;
    if (cell_ptr && cell_len) { src.ptr = cell_ptr; src.len = cell_len; }
    else if (cell_off) { src.text = text; src.off = cell_off; }
    else if (n_cells > 0) return DYD_ERR_INVALID;
    dyd_scan *h = new (std::nothrow) dyd_scan();
    if (!h) return DYD_ERR_OOM;
    const bool use_fast = use_fast_lane();
    int first_rc = DYD_OK;
    try {
        // twice the CPU share's worth of parts: a part's thread sleeps through its copies and its launches, and under a cgroup quota
        // idle time is not charged — tools/threads_ab.sh, 1 M rows on a 16-CPU slice: 12 / 16 / 24 / 32 threads 1.57 / 1.46 /
        // 1.40 / 1.36 s for the whole DataFrame route.  An explicit count (argument or DYD_HOST_THREADS) is taken as given.
        if (n_threads <= 0 && !getenv("DYD_HOST_THREADS") && n_cells >= 65536) n_threads = std::min(64, 2 * default_threads());
        init_polygon_handle(h, n_cells, n_threads, src);
        h->pipelined = true;
        h->high.assign((size_t)n_cells, 0);
        // A part's thread takes its cells through scan -> launch -> emit CHUNK by chunk (DYD_PIPE_CHUNK_KB of cell text, default 8192):
        // the point / segment buffers of a chunk are reused by the next one and stay small, and the points are scanned straight into
        // the pinned arena of the staging slot the thread holds for the whole pass (dyd_stage_acquire: stream, events, device arena
        // kept by the context), so a chunk's way to the device is three DMA copies from memory that is already pinned, one fused
        // launch and two copies back — nothing is created, allocated or freed per launch.
        size_t chunk_bytes = (size_t)8 << 20;
        if (const char *e = getenv("DYD_PIPE_CHUNK_KB")) { const long v = atol(e); if (v > 0) chunk_bytes = (size_t)v << 10; }
        std::vector<double> ts((size_t)h->parts.size() * 3, 0.0);
        auto up256 = [](size_t v) { return (v + 255) & ~(size_t)255; };
        if (!parallel_index_safe((int)h->parts.size(), [&](int k) {
                FastPart &A = *h->parts[(size_t)k];
                const size_t part_cells = (size_t)(A.hi - A.lo);
                if (!part_cells) return;
                auto fail = [&](int rc) {
                    int expected = DYD_OK;
                    __atomic_compare_exchange_n(&first_rc, &expected, rc, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED);
                };
                const size_t part_bytes = src.bytes(A.lo, A.hi);
                {   // the part's output text: a warm block of an earlier pass if one is parked
                    size_t cap = 0;
                    if (char *blk = host_pool().take(part_bytes / 2 + 64, &cap)) A.out.own(blk, cap);
                    A.out.need(part_bytes / 2 + 64);
                    A.out_len.need(part_cells);
                }
                const size_t chunk_cap = std::min(chunk_bytes, part_bytes) + 4096;
                dyd_stage *stage = nullptr;
                void *pin = nullptr;
                size_t pin_cap = 0;
                {   // pinned: the points (scan_part reserves 2/3 of the text bytes) and, behind them, offsets / arg indices / flags
                    const int rc = dyd_stage_acquire(chunk_cap - chunk_cap / 8, &stage, &pin, &pin_cap);
                    if (rc) { fail(rc); return; }
                }
                FastPart C;                               // the chunk in work; its buffers serve chunk after chunk
                Raw<int32_t> v_pt_off, v_box_off, v_arg4; // used when a chunk does not fit the pinned arena
                Raw<uint8_t> v_high;
                size_t tot_boxes = 0, tot_pts = 0;
                int64_t lane_cells = 0;
                double t_scan = 0, t_dev = 0, t_emit = 0;
                int64_t c0 = A.lo;
                while (c0 < A.hi && !__atomic_load_n(&first_rc, __ATOMIC_RELAXED)) {
                    int64_t c1 = c0;
                    size_t bytes = 0;
                    while (c1 < A.hi && (c1 == c0 || bytes + src.bytes(c1, c1 + 1) <= chunk_bytes)) { bytes += src.bytes(c1, c1 + 1); ++c1; }
                    C.lo = c0; C.hi = c1;
                    const size_t ncell = (size_t)(c1 - c0);
                    const size_t xy_room = bytes / 12 + 64 + 64;     // doubles scan_part asks for up front
                    if (pin && 8 * xy_room + 4096 <= pin_cap) C.xy.adopt(static_cast<double *>(pin), (pin_cap - 4096) / 8);
                    else C.xy.clear_free();
                    auto t0 = std::chrono::steady_clock::now();
                    scan_part(h, src, missing, C, use_fast);
                    const size_t nb = C.npts.n;
                    // where the small arrays live: behind the points in the pinned arena when they fit
                    int32_t *pt_off, *box_off, *arg4;
                    uint8_t *high;
                    const size_t o_po = up256(8 * C.xy.n), o_bo = o_po + up256(4 * (nb + 1)), o_arg = o_bo + up256(4 * (ncell + 1)),
                                 o_high = o_arg + up256(16 * nb + 16), o_end = o_high + up256(ncell);
                    if (C.xy.lent && o_end <= pin_cap) {
                        char *base = static_cast<char *>(pin);
                        pt_off = reinterpret_cast<int32_t *>(base + o_po);
                        box_off = reinterpret_cast<int32_t *>(base + o_bo);
                        arg4 = reinterpret_cast<int32_t *>(base + o_arg);
                        high = reinterpret_cast<uint8_t *>(base + o_high);
                    } else {
                        v_pt_off.n = v_box_off.n = v_arg4.n = v_high.n = 0;
                        v_pt_off.need(nb + 1); v_box_off.need(ncell + 1); v_arg4.need(4 * nb + 4); v_high.need(ncell);
                        pt_off = v_pt_off.p; box_off = v_box_off.p; arg4 = v_arg4.p; high = v_high.p;
                    }
                    int64_t run = 0;
                    pt_off[0] = 0;
                    for (size_t b = 0; b < nb; ++b) { run += C.npts.p[b]; pt_off[b + 1] = (int32_t)run; }
                    int64_t runb = 0;
                    box_off[0] = 0;
                    for (size_t c = 0; c < ncell; ++c) { runb += C.cell_boxes.p[c]; box_off[c + 1] = (int32_t)runb; }
                    auto t1 = std::chrono::steady_clock::now();
                    int rc = (run >= ((int64_t)1 << 31)) ? DYD_ERR_RANGE : DYD_OK;
                    if (!rc) rc = dyd_bbox_iou_fused_staged(stage, C.xy.p, pt_off, box_off, (int64_t)ncell, min_boxes, thr, nullptr, arg4, high);
                    auto t2 = std::chrono::steady_clock::now();
                    if (rc) { fail(rc); break; }
                    memcpy(h->high.data() + c0, high, ncell);
                    C.xyv = C.xy.p;
                    C.ptv = pt_off;
                    C.ptv_bias = 0;
                    if (!emit_part(h, src, C, arg4, A.out, A.out_len)) { fail(DYD_ERR_INVALID); break; }
                    C.xyv = nullptr; C.ptv = nullptr;
                    tot_boxes += nb;
                    tot_pts += (size_t)run;
                    for (size_t c = 0; c < C.lane.n; ++c) lane_cells += C.lane.p[c];
                    auto t3 = std::chrono::steady_clock::now();
                    t_scan += std::chrono::duration<double>(t1 - t0).count();
                    t_dev += std::chrono::duration<double>(t2 - t1).count();
                    t_emit += std::chrono::duration<double>(t3 - t2).count();
                    c0 = c1;
                }
                C.xy.clear_free();                        // lent memory is only forgotten
                dyd_stage_release(stage);
                A.box_base = tot_boxes;                   // totals are summed below
                A.pt_base = tot_pts;
                A.lane_count = lane_cells;
                ts[(size_t)k * 3 + 0] = t_scan;
                ts[(size_t)k * 3 + 1] = t_dev;
                ts[(size_t)k * 3 + 2] = t_emit;
            }))
            throw std::bad_alloc();
        if (first_rc) { delete h; return first_rc; }
        for (size_t k = 0; k < h->parts.size(); ++k) {
            h->tot_boxes += (int64_t)h->parts[k]->box_base;
            h->tot_points += (int64_t)h->parts[k]->pt_base;
            h->t_scan = std::max(h->t_scan, ts[k * 3]);
            h->t_device = std::max(h->t_device, ts[k * 3 + 1]);
            h->t_emit = std::max(h->t_emit, ts[k * 3 + 2]);
            // per part offsets of the emitted text (cells + 1 entries) for dyd_scan_part
            FastPart &A = *h->parts[k];
            const size_t ncell = (size_t)(A.hi - A.lo);
            A.out_off.need(ncell + 1);
            A.out_off.p[0] = 0;
            for (size_t c = 0; c < ncell; ++c) A.out_off.p[c + 1] = A.out_off.p[c] + A.out_len.p[c];
            A.out_off.n = ncell + 1;
        }
    } catch (const std::bad_alloc &) {
        delete h;
        return DYD_ERR_OOM;
    }
    *out = h;
    return DYD_OK;
}

const uint8_t *dyd_scan_high(const dyd_scan *h) { return (h && h->pipelined) ? h->high.data() : nullptr; }
int32_t dyd_scan_parts(const dyd_scan *h) { return h ? (int32_t)h->parts.size() : 0; }
// part k covers cells [*lo, *hi); its emitted text is text[off[c - *lo] .. off[c - *lo + 1]) (empty for cells that are not regular)
int dyd_scan_part(const dyd_scan *h, int32_t k, int64_t *lo, int64_t *hi, const uint8_t **text, const int64_t **off) {
    if (!h || !h->pipelined || k < 0 || (size_t)k >= h->parts.size() || !lo || !hi || !text || !off) return DYD_ERR_INVALID;
    const FastPart &A = *h->parts[(size_t)k];
    if (!A.out_off.n) return DYD_ERR_INVALID;   // already gathered by dyd_scan_text
    *lo = A.lo; *hi = A.hi;
    *text = reinterpret_cast<const uint8_t *>(A.out.p);
    *off = A.out_off.p;
    return DYD_OK;
}
// the parts' texts as one buffer + offsets [n_cells + 1] (for the CSV writer); the per-part views end here
int dyd_scan_text(dyd_scan *h, const uint8_t **text, const int64_t **off) {
    if (!h || !h->pipelined || !text || !off) return DYD_ERR_INVALID;
    try {
        if (h->text_off.empty()) {
            if (!gather_text(h)) return DYD_ERR_OOM;
            for (auto &pp : h->parts) pp->out_off.clear_free();
        }
    } catch (const std::bad_alloc &) {
        return DYD_ERR_OOM;
    }
    *text = reinterpret_cast<const uint8_t *>(h->f_text.p);
    *off = h->text_off.data();
    return DYD_OK;
}
// n_boxes, n_points, fast-lane cells, and the slowest part's seconds in scan / device stage / emit
void dyd_scan_totals(const dyd_scan *h, int64_t *counts3, double *seconds3) {
    if (!h) return;
    if (counts3) {
        counts3[0] = h->tot_boxes; counts3[1] = h->tot_points;
        int64_t lanes = 0;
        for (const auto &pp : h->parts) lanes += pp->lane_count;
        counts3[2] = lanes;
    }
    if (seconds3) { seconds3[0] = h->t_scan; seconds3[1] = h->t_device; seconds3[2] = h->t_emit; }
}

int64_t dyd_scan_n_boxes(const dyd_scan *h) { return h ? (h->fast ? (int64_t)h->f_pt_off.n - 1 : (int64_t)h->pt_off.size() - 1) : 0; }
int64_t dyd_scan_n_points(const dyd_scan *h) { return h ? (h->fast ? (int64_t)h->f_xy.n / 2 : (int64_t)h->xy.size() / 2) : 0; }
const double *dyd_scan_xy(const dyd_scan *h) { return h->fast ? h->f_xy.p : h->xy.data(); }
const int32_t *dyd_scan_pt_off(const dyd_scan *h) { return h->fast ? h->f_pt_off.p : h->pt_off.data(); }
const int32_t *dyd_scan_cell_box_off(const dyd_scan *h) { return h->cell_box_off.data(); }
const uint8_t *dyd_scan_status(const dyd_scan *h) { return h->status.data(); }
const uint8_t *dyd_scan_wh_kind(const dyd_scan *h, int which) { return which ? h->h_kind.data() : h->w_kind.data(); }
const double *dyd_scan_wh_value(const dyd_scan *h, int which) { return which ? h->h_val.data() : h->w_val.data(); }
const uint8_t *dyd_scan_iou_host(const dyd_scan *h) { return h->iou_host.empty() ? nullptr : h->iou_host.data(); }
int64_t dyd_scan_fast_cells(const dyd_scan *h) {
    int64_t n = 0;
    if (h)
        for (const auto &pp : h->parts)
            for (size_t i = 0; i < pp->lane.n; ++i) n += pp->lane.p[i];
    return n;
}

// Emit the rewritten JSON text of every CELL_OK cell (empty text for the others).  arg4 = K1's arg indices
// for the boxes of the scan, in scan order.  Output stays owned by the handle.  Cells of the fast lane are assembled from
// their segments (no parsing); the others are re-walked by the exact parser.  text / cell_off may be NULL: the cells the
// scan was given are used (they must still be alive).
int dyd_json_emit_polygons(dyd_scan *h, const uint8_t *text, const int64_t *cell_off, const int32_t *arg4,
                           int n_threads, const uint8_t **out_text, const int64_t **out_off) {
    if (!h || !out_text || !out_off || !h->fast) return DYD_ERR_INVALID;
    (void)n_threads;   // one thread per part of the scan
    CellSrc src = h->src;
    if (text && cell_off) { src = CellSrc(); src.text = text; src.off = cell_off; }
    int bad = 0;
    try {
        if (!parallel_index_safe((int)h->parts.size(), [&](int k) {
                FastPart &A = *h->parts[(size_t)k];
                A.out.n = 0;
                A.out_len.n = 0;
                if (!emit_part(h, src, A, arg4 + 4 * A.box_base, A.out, A.out_len)) __atomic_store_n(&bad, 1, __ATOMIC_RELAXED);
            }))
            return DYD_ERR_OOM;
        if (bad) return DYD_ERR_INVALID;  // arg4 inconsistent with the scan
        if (!gather_text(h)) return DYD_ERR_OOM;
    } catch (const std::bad_alloc &) {
        return DYD_ERR_OOM;
    }
    *out_text = reinterpret_cast<const uint8_t *>(h->f_text.p);
    *out_off = h->text_off.data();
    return DYD_OK;
}

// ---- synthetic annotation cells (measurement / test aid) ------------------------------------------------------
// The JSON text json.dumps(..., ensure_ascii=False) gives for the rows of a SynthTable (synth.py: row_json), written by
// all cores, so that bench.py and the parity tests can build million-row JSON tables in seconds instead of minutes:
//   {"width": W, "height": H, "objects": [{"id": k, "name": "c<label>", "polygon": {"ptList": [{"x": .., "y": ..}, ...]},
//   "type": "polygon"}, ...]}   — coordinates of an int_row printed as ints, the others as float repr.
int dyd_synth_json(const double *xy, const int32_t *pt_off, const int32_t *box_off, const int32_t *label, const uint8_t *int_row,
                   int64_t n_rows, int64_t width, int64_t height, int n_threads, uint8_t **out_text, int64_t *out_off) {
    if (!out_text || !out_off || n_rows < 0 || (n_rows > 0 && (!pt_off || !box_off || !label || !int_row))) return DYD_ERR_INVALID;
    *out_text = nullptr;
    if (n_threads <= 0) n_threads = default_threads();
    n_threads = (int)std::max<int64_t>(1, std::min<int64_t>(std::min(n_threads, 64), n_rows / 64));
    std::vector<Raw<char>> parts((size_t)n_threads);
    std::vector<int64_t> lens((size_t)n_rows, 0);
    if (!parallel_index_safe(n_threads, [&](int t) {
            Raw<char> &o = parts[(size_t)t];
            std::string tmp;
            char head[96];
            const int hl = snprintf(head, sizeof(head), "{\"width\": %lld, \"height\": %lld, \"objects\": [", (long long)width, (long long)height);
            const int64_t lo = n_rows * t / n_threads, hi = n_rows * (t + 1) / n_threads;
            for (int64_t r = lo; r < hi; ++r) {
                const size_t mark = o.n;
                o.put(head, (size_t)hl);
                const bool as_int = int_row[r] != 0;
                for (int32_t b = box_off[r]; b < box_off[r + 1]; ++b) {
                    char buf[96];
                    const int k = snprintf(buf, sizeof(buf), "%s{\"id\": %d, \"name\": \"c%d\", \"polygon\": {\"ptList\": [", b > box_off[r] ? ", " : "",
                                           (int)(b - box_off[r]), (int)label[b]);
                    o.put(buf, (size_t)k);
                    for (int32_t p = pt_off[b]; p < pt_off[b + 1]; ++p) {
                        o.put(p > pt_off[b] ? ", {\"x\": " : "{\"x\": ", p > pt_off[b] ? 8 : 6);
                        fj_put_coord(o, xy[2 * (size_t)p], as_int, tmp);
                        o.put(", \"y\": ", 7);
                        fj_put_coord(o, xy[2 * (size_t)p + 1], as_int, tmp);
                        o.push('}');
                    }
                    o.put("]}, \"type\": \"polygon\"}", 22);
                }
                o.put("]}", 2);
                lens[(size_t)r] = (int64_t)(o.n - mark);
            }
        }))
        return DYD_ERR_OOM;
    size_t total = 0;
    std::vector<size_t> base((size_t)n_threads, 0);
    for (int t = 0; t < n_threads; ++t) { base[(size_t)t] = total; total += parts[(size_t)t].n; }
    uint8_t *buf = static_cast<uint8_t *>(malloc(total ? total : 1));
    if (!buf) return DYD_ERR_OOM;
    parallel_index_safe(n_threads, [&](int t) { if (parts[(size_t)t].n) memcpy(buf + base[(size_t)t], parts[(size_t)t].p, parts[(size_t)t].n); });
    out_off[0] = 0;
    for (int64_t r = 0; r < n_rows; ++r) out_off[r + 1] = out_off[r] + lens[(size_t)r];
    *out_text = buf;
    return DYD_OK;
}

void dyd_scan_free(dyd_scan *h) { delete h; }

// Scan bbox-JSON cells for the IoU step (processor.py:341-366): two-point boxes per row with the
// prefix-on-exception rule.  status[i] = 0 regular, 2 = Python path.  Arrays are owned by the handle
// (xy = box4 [4*B], pt_off unused, cell_box_off = row offsets).
int dyd_json_scan_boxes(const uint8_t *text, const int64_t *cell_off, const uint8_t *missing, int64_t n_cells,
                        int n_threads, dyd_scan **out) {
    if (!out || n_cells < 0 || (n_cells > 0 && !cell_off)) return DYD_ERR_INVALID;
    dyd_scan *h = new (std::nothrow) dyd_scan();
    if (!h) return DYD_ERR_OOM;
    h->n_cells = n_cells;
    h->status.assign((size_t)n_cells, CELL_OK);
    std::vector<int32_t> counts((size_t)n_cells, 0);
    struct Part { std::vector<double> b; int64_t lo = 0, hi = 0; };
    std::vector<Part> parts(64);
    try {
        parallel_cells(n_cells, n_threads, [&](int t, int64_t lo, int64_t hi) {
            Part &pt = parts[(size_t)t];
            pt.lo = lo; pt.hi = hi;
            for (int64_t i = lo; i < hi; ++i) {
                if (missing && missing[i]) continue;  // NaN cell: no boxes (:344-345)
                int32_t c = 0;
                if (!boxes_of_cell(Span{(const char *)text + cell_off[i], (const char *)text + cell_off[i + 1]}, pt.b, c)) {
                    h->status[(size_t)i] = CELL_IRREGULAR;
                    c = 0;
                }
                counts[(size_t)i] = c;
            }
        });
        std::sort(parts.begin(), parts.end(), [](const Part &a, const Part &b) { return a.lo < b.lo; });
        size_t tot = 0;
        for (auto &pt : parts) tot += pt.b.size();
        if (tot / 4 >= (size_t)1 << 31) { delete h; return DYD_ERR_RANGE; }
        h->xy.reserve(tot);
        for (auto &pt : parts) h->xy.insert(h->xy.end(), pt.b.begin(), pt.b.end());
        h->cell_box_off.resize((size_t)n_cells + 1);
        h->cell_box_off[0] = 0;
        for (int64_t i = 0; i < n_cells; ++i) h->cell_box_off[(size_t)i + 1] = h->cell_box_off[(size_t)i] + counts[(size_t)i];
        h->pt_off.assign(1, 0);
    } catch (const std::bad_alloc &) {
        delete h;
        return DYD_ERR_OOM;
    }
    *out = h;
    return DYD_OK;
}

}  // extern "C"

// ===================================================================================================
// split step: expansion of every row into one record per (object, label in the rules)
// (reference core/processor.py:712-792; helpers utils.py:645-662)
// ===================================================================================================
#include <string_view>
#include <unordered_map>

namespace {

enum SplitStatus : uint8_t {
    SP_OK = 0,            // expanded (possibly into zero records)
    SP_EMPTY = 1,         // "空数据": NaN / non-str / "" cell (decided by the caller through `missing`)
    SP_UNDECODABLE = 2,   // "JSON解析失败"
    SP_NOT_A_LIST = 3,    // "objects不是列表"
    SP_NO_OBJECTS = 4,    // "标注字段objects为空"
    SP_IRREGULAR = 5      // the Python path decides (non-dict document, odd "name" values, duplicate keys ...)
};
enum SplitEvent : uint8_t { EV_NO_NAME = 1, EV_UNDEFINED = 2, EV_NOTHING_CLASSIFIED = 3 };

// str.isspace() code points: what str.strip() removes
bool py_space(uint32_t c) {
    return (c >= 0x09 && c <= 0x0d) || (c >= 0x1c && c <= 0x20) || c == 0x85 || c == 0xa0 || c == 0x1680 ||
           (c >= 0x2000 && c <= 0x200a) || c == 0x2028 || c == 0x2029 || c == 0x202f || c == 0x205f || c == 0x3000;
}

// decodes one UTF-8 code point of a VALID sequence (the buffers come from Python's str.encode)
uint32_t next_cp(const char *&q, const char *e) {
    const unsigned char c = (unsigned char)*q;
    if (c < 0x80) { ++q; return c; }
    int n = (c >= 0xf0) ? 3 : (c >= 0xe0) ? 2 : 1;
    uint32_t cp = c & (0x3f >> n);
    ++q;
    while (n-- > 0 && q < e) cp = (cp << 6) | ((unsigned char)*q++ & 0x3f);
    return cp;
}

// the decoded (UTF-8) value of a raw JSON string span
void decode_string(Parser &ps, Span s, std::string &out) {
    out.clear();
    const char *q = s.b;
    while (q < s.e) {
        const unsigned char c = (unsigned char)*q;
        if (c != '\\') { out += (char)c; ++q; continue; }
        const char esc = q[1];
        if (esc == 'u') {
            uint32_t cp = (uint32_t)hex4(q + 2);
            q += 6;
            if (cp >= 0xD800 && cp <= 0xDBFF) {   // string_token accepted only completed pairs
                const int lo = hex4(q + 2);
                cp = 0x10000 + ((cp - 0xD800) << 10) + ((uint32_t)lo - 0xDC00);
                q += 6;
            } else if (cp >= 0xDC00 && cp <= 0xDFFF) {
                ps.irregular();
            }
            append_utf8(out, cp);
        } else {
            switch (esc) {
                case 'b': out += '\b'; break;
                case 'f': out += '\f'; break;
                case 'n': out += '\n'; break;
                case 'r': out += '\r'; break;
                case 't': out += '\t'; break;
                default: out += esc; break;   // " \ /
            }
            q += 2;
        }
    }
}

// re.split(r"[,，;；|]", name), every token stripped, empty tokens dropped (utils.py:659-662)
void split_labels(const std::string &name, std::vector<std::string> &labels) {
    labels.clear();
    const char *q = name.data(), *e = q + name.size();
    const char *tok_b = q;
    auto flush = [&](const char *tb, const char *te) {
        const char *a = tb;   // strip from the left
        while (a < te) {
            const char *n = a;
            if (!py_space(next_cp(n, te))) break;
            a = n;
        }
        const char *z = te;   // strip from the right: walk code points, remember the end of the last non-space
        const char *last = a;
        for (const char *w = a; w < z;) {
            const char *n = w;
            const uint32_t cp = next_cp(n, z);
            if (!py_space(cp)) last = n;
            w = n;
        }
        if (last > a) labels.emplace_back(a, last);
    };
    while (q < e) {
        const char *n = q;
        const uint32_t cp = next_cp(n, e);
        if (cp == ',' || cp == ';' || cp == '|' || cp == 0xFF0C || cp == 0xFF1B) {
            flush(tok_b, q);
            tok_b = n;
        }
        q = n;
    }
    flush(tok_b, e);
}

// json.dumps(str, ensure_ascii=False) of decoded UTF-8 text
void quote_utf8(std::string &out, const std::string &s) {
    out += '"';
    for (const char ch : s) {
        const unsigned char c = (unsigned char)ch;
        if (c == '"' || c == '\\' || c < 0x20) append_escaped_cp(out, c);
        else out += ch;
    }
    out += '"';
}

struct ObjText {          // canonical text of one dict element of "objects", the value of "name" cut out
    std::string before, after;
    bool has_name = false;
    std::string name;     // decoded, "" when the value is falsy (no labels either way)
};

struct SplitPart {        // per-thread outputs, concatenated in cell order afterwards
    int64_t lo = 0, hi = 0;
    std::string json, ev_text, combo, reasons;
    std::vector<int64_t> json_end, row_cell, ev_cell, ev_text_end;
    std::vector<int32_t> row_label;
    std::vector<uint8_t> ev_kind;
};

using LabelMap = std::unordered_map<std::string_view, int32_t>;

// one dict element of "objects" (p at '{') -> canonical text around the name value
void split_object(Parser &ps, ObjText &o) {
    o.before.clear(); o.after.clear(); o.has_name = false; o.name.clear();
    std::string *cur = &o.before;
    ++ps.p;
    *cur += '{';
    KeySet ks;
    bool first = true;
    if (ps.peek() == '}') { ++ps.p; *cur += '}'; return; }
    while (true) {
        ps.ws();
        const Span k = ps.string_token();
        ks.add(ps, k);
        if (!first) *cur += ", ";
        first = false;
        ps.emit_string(*cur, k);
        *cur += ": ";
        ps.ws();
        if (ps.p >= ps.end || *ps.p != ':') ps.bad();
        ++ps.p;
        if (Parser::span_is(k, "name")) {
            o.has_name = true;
            const Kind kd = kind_of(ps.peek());
            if (kd == K_STRING) {
                ps.ws();
                const Span v = ps.string_token();
                decode_string(ps, v, o.name);
            } else if (kd == K_NULL || kd == K_FALSE) {
                ps.value(nullptr);                       // falsy: no labels
            } else if (kd == K_ARRAY || kd == K_OBJECT) {
                const char open = *ps.p;
                ++ps.p;
                if (ps.peek() != (open == '[' ? ']' : '}')) ps.irregular();   // str(list / dict) as a label: Python path
                ++ps.p;
            } else {
                ps.irregular();                          // numbers (0 is falsy, 7 -> "7") and true: Python path
            }
            cur = &o.after;
        } else {
            ps.value(cur);
        }
        const char d = ps.peek();
        if (d == ',') { ++ps.p; continue; }
        if (d == '}') { ++ps.p; break; }
        ps.bad();
    }
    *cur += '}';
}

// whole cell.  Throws Fail.  Returns the status; fills part with the cell's records / events.
SplitStatus split_cell(Span cell, int64_t ci, const LabelMap &map, SplitPart &pt, int32_t &n_out) {
    Parser ps{cell.b, cell.e};
    n_out = 0;
    ps.ws();
    if (ps.p >= ps.end) ps.bad();
    if (*ps.p != '{') {   // list / scalar document: data.get raises -> str(exception) is the reason: Python path
        ps.value(nullptr);
        ps.ws();
        if (ps.p != ps.end) ps.bad();
        ps.irregular();
    }
    ++ps.p;
    std::string head;     // canonical "key": value of every member except "objects", ", "-joined
    std::vector<ObjText> objs;
    int64_t n_elements = 0;
    int objects_kind = -1;   // -1 absent, 0 array, 1 something else
    KeySet ks;
    if (ps.peek() == '}') {
        ++ps.p;
    } else {
        while (true) {
            ps.ws();
            const Span k = ps.string_token();
            ks.add(ps, k);
            ps.ws();
            if (ps.p >= ps.end || *ps.p != ':') ps.bad();
            ++ps.p;
            if (Parser::span_is(k, "objects")) {
                if (kind_of(ps.peek()) != K_ARRAY) {
                    objects_kind = 1;
                    ps.value(nullptr);
                } else {
                    objects_kind = 0;
                    ++ps.p;
                    if (ps.peek() == ']') {
                        ++ps.p;
                    } else {
                        while (true) {
                            ++n_elements;
                            if (ps.peek() == '{') {
                                objs.emplace_back();
                                split_object(ps, objs.back());
                            } else {
                                ps.value(nullptr);   // non-dict elements are skipped (:742)
                            }
                            const char d = ps.peek();
                            if (d == ',') { ++ps.p; continue; }
                            if (d == ']') { ++ps.p; break; }
                            ps.bad();
                        }
                    }
                }
            } else {
                if (!head.empty()) head += ", ";
                ps.emit_string(head, k);
                head += ": ";
                ps.value(&head);
            }
            const char d = ps.peek();
            if (d == ',') { ++ps.p; continue; }
            if (d == '}') { ++ps.p; break; }
            ps.bad();
        }
    }
    ps.ws();
    if (ps.p != ps.end) ps.bad();
    if (objects_kind == 1) return SP_NOT_A_LIST;
    if (n_elements == 0) return SP_NO_OBJECTS;

    std::vector<std::string> labels, all, undefined;
    std::vector<std::vector<std::string>> per_obj(objs.size());
    for (size_t i = 0; i < objs.size(); ++i) {
        if (objs[i].has_name && !objs[i].name.empty()) split_labels(objs[i].name, per_obj[i]);
        all.insert(all.end(), per_obj[i].begin(), per_obj[i].end());
    }
    std::sort(all.begin(), all.end());
    all.erase(std::unique(all.begin(), all.end()), all.end());
    for (size_t i = 0; i < all.size(); ++i) {   // "，".join(sorted(raw_label_set)) (:736)
        if (i) pt.combo += "\xef\xbc\x8c";
        pt.combo += all[i];
    }
    auto event = [&](uint8_t kind, const std::string &text) {
        pt.ev_cell.push_back(ci);
        pt.ev_kind.push_back(kind);
        pt.ev_text += text;
        pt.ev_text_end.push_back((int64_t)pt.ev_text.size());
    };
    for (size_t i = 0; i < objs.size(); ++i) {
        if (per_obj[i].empty()) { event(EV_NO_NAME, std::string()); continue; }   // "标注框缺少name字段" (:746-749)
        for (const std::string &label : per_obj[i]) {
            const auto it = map.find(std::string_view(label));
            if (it == map.end()) {                                               // :752-758
                event(EV_UNDEFINED, label);
                undefined.push_back(label);
                continue;
            }
            std::string &o = pt.json;                                            // :760-767
            o += '{';
            o += head;
            if (!head.empty()) o += ", ";
            o += "\"objects\": [";
            o += objs[i].before;
            quote_utf8(o, label);
            o += objs[i].after;
            o += "]}";
            pt.json_end.push_back((int64_t)o.size());
            pt.row_cell.push_back(ci);
            pt.row_label.push_back(it->second);
            ++n_out;
        }
    }
    std::sort(undefined.begin(), undefined.end());
    undefined.erase(std::unique(undefined.begin(), undefined.end()), undefined.end());
    for (size_t i = 0; i < undefined.size(); ++i) {   // "；".join(sorted(row_reason_set)) (:779, :790)
        if (i) pt.reasons += "\xef\xbc\x9b";
        pt.reasons += "\xe6\xa0\x87\xe7\xad\xbe";                      // 标签
        pt.reasons += undefined[i];
        pt.reasons += "\xe6\x9c\xaa\xe5\x9c\xa8\xe8\xa7\x84\xe5\x88\x99\xe4\xb8\xad\xe5\xae\x9a\xe4\xb9\x89";   // 未在规则中定义
    }
    if (n_out == 0) event(EV_NOTHING_CLASSIFIED, std::string());
    return SP_OK;
}

#include "host_split_fast.h"

}  // namespace

struct dyd_split {
    int64_t n_cells = 0;
    std::vector<uint8_t> status;
    std::vector<int32_t> n_expanded;
    std::vector<std::unique_ptr<SplitPartF>> parts;     // own the record texts (rec_ptr points into them)
    // gathered by all threads at once
    Raw<int64_t> row_cell, ev_cell, rec_len;
    Raw<uint64_t> rec_ptr;
    Raw<int32_t> row_label, ev_code;
    Raw<uint8_t> ev_kind;
    Raw<char> combo, reasons;
    std::vector<int64_t> combo_off, reasons_off;
    std::vector<int32_t> reason_code;      // per cell: index of its reasons text among the distinct ones (-1: none); empty: not made
    std::vector<int64_t> reason_first;     // per distinct reasons text the first cell carrying it
    std::vector<std::string> undef_names;               // distinct undefined labels, first-appearance order over the parts
    std::string undef_text;
    std::vector<int64_t> undef_off;
    std::vector<int64_t> label_first, label_count;      // per label of the rules: first record carrying it (-1), records
    int64_t fast_cells = 0;
    bool all_ascii = true;                              // every record text is pure ASCII
    // on request (the older accessors): one flat copy of the record texts / one text per event
    Raw<char> json_flat;
    std::vector<int64_t> json_off;
    std::string ev_text;
    std::vector<int64_t> ev_text_off;
    double t_parse = 0, t_gather = 0;
};

namespace {

int split_expand_src(const CellSrc &src, const uint8_t *missing, int64_t n_cells, const uint8_t *label_text, const int64_t *label_off,
                     int32_t n_labels, int n_threads, dyd_split **out) {
    dyd_split *h = new (std::nothrow) dyd_split();
    if (!h) return DYD_ERR_OOM;
    try {
        const auto T0 = std::chrono::steady_clock::now();
        h->n_cells = n_cells;
        h->status.assign((size_t)n_cells, SP_OK);
        h->n_expanded.assign((size_t)n_cells, 0);
        LabelMap map;
        map.reserve((size_t)n_labels * 2 + 1);
        for (int32_t i = 0; i < n_labels; ++i)
            map.emplace(std::string_view((const char *)label_text + label_off[i], (size_t)(label_off[i + 1] - label_off[i])), i);
        if (n_threads <= 0) n_threads = default_threads();
        n_threads = (int)std::min<int64_t>(n_threads, std::max<int64_t>(1, n_cells / 256));
        n_threads = std::min(n_threads, 64);
        for (int t = 0; t < n_threads; ++t) {
            h->parts.emplace_back(new SplitPartF());
            h->parts.back()->lo = n_cells * t / n_threads;
            h->parts.back()->hi = n_cells * (t + 1) / n_threads;
        }
        std::vector<int64_t> combo_len((size_t)n_cells, 0), reasons_len((size_t)n_cells, 0);
        const bool use_fast = use_fast_lane();
        if (!parallel_index_safe((int)h->parts.size(), [&](int k) {
                SplitPartF &pt = *h->parts[(size_t)k];
                SplitLane L;
                pt.label_first.assign((size_t)n_labels, -1);
                pt.label_count.assign((size_t)n_labels, 0);
                const size_t bytes = src.bytes(pt.lo, pt.hi);
                pt.json.need(bytes + bytes / 2 + 64);
                pt.json_end.need(bytes / 128 + 64);
                pt.row_cell.need(bytes / 128 + 64);
                pt.row_label.need(bytes / 128 + 64);
                for (int64_t i = pt.lo; i < pt.hi; ++i) {
                    if (missing && missing[i]) { h->status[(size_t)i] = SP_EMPTY; continue; }
                    const Span cell = src.get(i);
                    if (cell.e == cell.b) { h->status[(size_t)i] = SP_EMPTY; continue; }     // "" is not a usable cell (:716)
                    uint8_t st = SP_OK;
                    int32_t n_out = 0;
                    const size_t m_json = pt.json.n, m_rec = pt.json_end.n, m_ev = pt.ev_cell.n, m_combo = pt.combo.n, m_reasons = pt.reasons.n;
                    if (use_fast && split_cell_fast(cell, i, map, L, pt, st, n_out, combo_len[(size_t)i], reasons_len[(size_t)i])) {
                        ++pt.fast_cells;
                    } else {
                        // a bail may have left nothing behind: the lane writes to pt only after its parse succeeded
                        try {
                            split_cell_slow(cell, i, map, L, pt, st, n_out, combo_len[(size_t)i], reasons_len[(size_t)i]);
                        } catch (Fail f) {
                            pt.json.n = m_json; pt.json_end.n = m_rec; pt.row_cell.n = m_rec; pt.row_label.n = m_rec;
                            pt.ev_cell.n = m_ev; pt.ev_kind.n = m_ev; pt.ev_code.n = m_ev; pt.combo.n = m_combo; pt.reasons.n = m_reasons;
                            st = (f.code == 1) ? SP_UNDECODABLE : SP_IRREGULAR;
                            n_out = 0;
                            combo_len[(size_t)i] = reasons_len[(size_t)i] = 0;
                        }
                    }
                    h->status[(size_t)i] = st;
                    h->n_expanded[(size_t)i] = n_out;
                }
                pt.ascii = all_ascii(pt.json.p, pt.json.p + pt.json.n);
            }))
            throw std::bad_alloc();
        const auto T1 = std::chrono::steady_clock::now();
        // ---- gather: bases, the table of undefined labels, then every thread copies its part's fixed-width arrays ------------
        size_t n_rec = 0, n_ev = 0;
        std::map<std::string, int32_t, std::less<>> undef_ix;
        std::vector<std::vector<int32_t>> remap(h->parts.size());
        h->label_first.assign((size_t)n_labels, -1);
        h->label_count.assign((size_t)n_labels, 0);
        for (size_t k = 0; k < h->parts.size(); ++k) {
            SplitPartF &pt = *h->parts[k];
            pt.rec_base = n_rec; pt.ev_base = n_ev;
            n_rec += pt.json_end.n; n_ev += pt.ev_cell.n;
            h->fast_cells += pt.fast_cells;
            h->all_ascii = h->all_ascii && pt.ascii;
            remap[k].resize(pt.undef_names.size());
            for (size_t u = 0; u < pt.undef_names.size(); ++u) {
                auto it = undef_ix.find(pt.undef_names[u]);
                if (it == undef_ix.end()) {
                    it = undef_ix.emplace(pt.undef_names[u], (int32_t)h->undef_names.size()).first;
                    h->undef_names.push_back(pt.undef_names[u]);
                }
                remap[k][u] = it->second;
            }
            for (int32_t l = 0; l < n_labels; ++l) {
                if (pt.label_first[(size_t)l] >= 0 && h->label_first[(size_t)l] < 0)
                    h->label_first[(size_t)l] = (int64_t)pt.rec_base + pt.label_first[(size_t)l];
                h->label_count[(size_t)l] += pt.label_count[(size_t)l];
            }
        }
        h->undef_off.push_back(0);
        for (const std::string &u : h->undef_names) { h->undef_text += u; h->undef_off.push_back((int64_t)h->undef_text.size()); }
        h->row_cell.need(n_rec + 1); h->row_cell.n = n_rec;
        h->row_label.need(n_rec + 1); h->row_label.n = n_rec;
        h->rec_ptr.need(n_rec + 1); h->rec_ptr.n = n_rec;
        h->rec_len.need(n_rec + 1); h->rec_len.n = n_rec;
        h->ev_cell.need(n_ev + 1); h->ev_cell.n = n_ev;
        h->ev_kind.need(n_ev + 1); h->ev_kind.n = n_ev;
        h->ev_code.need(n_ev + 1); h->ev_code.n = n_ev;
        h->combo_off.resize((size_t)n_cells + 1);
        h->reasons_off.resize((size_t)n_cells + 1);
        h->combo_off[0] = h->reasons_off[0] = 0;
        for (int64_t i = 0; i < n_cells; ++i) {
            h->combo_off[(size_t)i + 1] = h->combo_off[(size_t)i] + combo_len[(size_t)i];
            h->reasons_off[(size_t)i + 1] = h->reasons_off[(size_t)i] + reasons_len[(size_t)i];
        }
        h->combo.need((size_t)h->combo_off[(size_t)n_cells] + 1); h->combo.n = (size_t)h->combo_off[(size_t)n_cells];
        h->reasons.need((size_t)h->reasons_off[(size_t)n_cells] + 1); h->reasons.n = (size_t)h->reasons_off[(size_t)n_cells];
        if (!parallel_index_safe((int)h->parts.size(), [&](int k) {
                SplitPartF &pt = *h->parts[(size_t)k];
                const size_t nr = pt.json_end.n, ne = pt.ev_cell.n;
                if (nr) {
                    memcpy(h->row_cell.p + pt.rec_base, pt.row_cell.p, nr * sizeof(int64_t));
                    memcpy(h->row_label.p + pt.rec_base, pt.row_label.p, nr * sizeof(int32_t));
                    int64_t prev = 0;
                    for (size_t r = 0; r < nr; ++r) {
                        h->rec_ptr.p[pt.rec_base + r] = (uint64_t)(uintptr_t)(pt.json.p + prev);
                        h->rec_len.p[pt.rec_base + r] = pt.json_end.p[r] - prev;
                        prev = pt.json_end.p[r];
                    }
                }
                if (ne) {
                    memcpy(h->ev_cell.p + pt.ev_base, pt.ev_cell.p, ne * sizeof(int64_t));
                    memcpy(h->ev_kind.p + pt.ev_base, pt.ev_kind.p, ne);
                    for (size_t e = 0; e < ne; ++e) {
                        const int32_t c = pt.ev_code.p[e];
                        h->ev_code.p[pt.ev_base + e] = c < 0 ? -1 : remap[(size_t)k][(size_t)c];
                    }
                }
                if (pt.combo.n) memcpy(h->combo.p + h->combo_off[(size_t)pt.lo], pt.combo.p, pt.combo.n);
                if (pt.reasons.n) memcpy(h->reasons.p + h->reasons_off[(size_t)pt.lo], pt.reasons.p, pt.reasons.n);
                pt.row_cell.clear_free(); pt.row_label.clear_free(); pt.ev_cell.clear_free(); pt.ev_kind.clear_free();
                pt.ev_code.clear_free(); pt.combo.clear_free(); pt.reasons.clear_free();
            }))
            throw std::bad_alloc();
        {   // The reasons repeat (they name the undefined labels of a row): a code per cell and the first cell of every distinct
            // text let the caller build each distinct str ONCE (they are Chinese: a decode per cell otherwise).  Given up beyond
            // 4096 distinct texts (reason_code stays empty: the caller makes the strings cell by cell).
            h->reason_code.assign((size_t)n_cells, -1);
            std::unordered_map<std::string_view, int32_t> seen;
            bool ok = true;
            for (int64_t i = 0; i < n_cells && ok; ++i) {
                const int64_t a = h->reasons_off[(size_t)i], b = h->reasons_off[(size_t)i + 1];
                if (b == a) continue;
                const std::string_view sv(h->reasons.p + a, (size_t)(b - a));
                auto it = seen.find(sv);
                if (it == seen.end()) {
                    if (h->reason_first.size() >= 4096) { ok = false; break; }
                    it = seen.emplace(sv, (int32_t)h->reason_first.size()).first;
                    h->reason_first.push_back(i);
                }
                h->reason_code[(size_t)i] = it->second;
            }
            if (!ok) { h->reason_code.clear(); h->reason_first.clear(); }
        }
        const auto T2 = std::chrono::steady_clock::now();
        h->t_parse = std::chrono::duration<double>(T1 - T0).count();
        h->t_gather = std::chrono::duration<double>(T2 - T1).count();
    } catch (const std::bad_alloc &) {
        delete h;
        return DYD_ERR_OOM;
    }
    *out = h;
    return DYD_OK;
}

}  // namespace

extern "C" {

// labels: the keys of label_to_category, concatenated UTF-8 with offsets.  `missing[i]` != 0 marks a row without
// a usable JSON cell (status SP_EMPTY; so does an empty cell).  The handle owns every output.
int dyd_json_split_expand(const uint8_t *text, const int64_t *cell_off, const uint8_t *missing, int64_t n_cells,
                          const uint8_t *label_text, const int64_t *label_off, int32_t n_labels, int n_threads,
                          dyd_split **out) {
    if (!out || n_cells < 0 || n_labels < 0 || (n_cells > 0 && !cell_off) || (n_labels > 0 && (!label_text || !label_off)))
        return DYD_ERR_INVALID;
    CellSrc src;
    src.text = text; src.off = cell_off;
    return split_expand_src(src, missing, n_cells, label_text, label_off, n_labels, n_threads, out);
}

// the same over one (pointer, length) view per cell (the str objects of a DataFrame column: nothing is copied); the views
// must stay valid for the call only — the handle keeps no reference to the cells
int dyd_json_split_expand_v(const uint8_t *const *cell_ptr, const int64_t *cell_len, const uint8_t *missing, int64_t n_cells,
                            const uint8_t *label_text, const int64_t *label_off, int32_t n_labels, int n_threads,
                            dyd_split **out) {
    if (!out || n_cells < 0 || n_labels < 0 || (n_cells > 0 && (!cell_ptr || !cell_len)) || (n_labels > 0 && (!label_text || !label_off)))
        return DYD_ERR_INVALID;
    CellSrc src;
    src.ptr = cell_ptr; src.len = cell_len;
    return split_expand_src(src, missing, n_cells, label_text, label_off, n_labels, n_threads, out);
}

const uint8_t *dyd_split_status(const dyd_split *h) { return h->status.data(); }
const int32_t *dyd_split_n_expanded(const dyd_split *h) { return h->n_expanded.data(); }
int64_t dyd_split_rows(const dyd_split *h) { return (int64_t)h->row_cell.n; }
const int64_t *dyd_split_row_cell(const dyd_split *h) { return h->row_cell.p; }
const int32_t *dyd_split_row_label(const dyd_split *h) { return h->row_label.p; }
int64_t dyd_split_events(const dyd_split *h) { return (int64_t)h->ev_cell.n; }
const int64_t *dyd_split_event_cell(const dyd_split *h) { return h->ev_cell.p; }
const uint8_t *dyd_split_event_kind(const dyd_split *h) { return h->ev_kind.p; }
const int32_t *dyd_split_event_code(const dyd_split *h) { return h->ev_code.p; }
int64_t dyd_split_undefined(const dyd_split *h) { return (int64_t)h->undef_names.size(); }
const int64_t *dyd_split_label_first(const dyd_split *h) { return h->label_first.data(); }
const int64_t *dyd_split_label_count(const dyd_split *h) { return h->label_count.data(); }
int64_t dyd_split_fast_cells(const dyd_split *h) { return h->fast_cells; }
int dyd_split_all_ascii(const dyd_split *h) { return h->all_ascii ? 1 : 0; }
const int32_t *dyd_split_reason_code(const dyd_split *h) { return h->reason_code.empty() ? nullptr : h->reason_code.data(); }
int64_t dyd_split_reason_distinct(const dyd_split *h) { return (int64_t)h->reason_first.size(); }
const int64_t *dyd_split_reason_first(const dyd_split *h) { return h->reason_first.data(); }
void dyd_split_seconds(const dyd_split *h, double *parse_gather2) {
    if (h && parse_gather2) { parse_gather2[0] = h->t_parse; parse_gather2[1] = h->t_gather; }
}
// one (address, length) view per record, in row order; valid until dyd_split_free
int dyd_split_rec_views(const dyd_split *h, const uint64_t **ptr, const int64_t **len) {
    if (!h || !ptr || !len) return DYD_ERR_INVALID;
    *ptr = h->rec_ptr.p;
    *len = h->rec_len.p;
    return DYD_OK;
}
// which: 0 record JSON [rows] (a flat copy made on the first request), 1 label combination per cell [n_cells], 2 joined
// reasons per cell [n_cells], 3 label of each event [events] (made on the first request), 4 the distinct undefined labels
// [dyd_split_undefined] that dyd_split_event_code indexes
int dyd_split_strings(dyd_split *h, int which, const uint8_t **data, const int64_t **off) {
    if (!h || !data || !off) return DYD_ERR_INVALID;
    try {
        switch (which) {
            case 0: {
                if (h->json_off.empty()) {
                    const size_t n_rec = h->row_cell.n;
                    h->json_off.resize(n_rec + 1);
                    h->json_off[0] = 0;
                    for (size_t r = 0; r < n_rec; ++r) h->json_off[r + 1] = h->json_off[r] + h->rec_len.p[r];
                    h->json_flat.need((size_t)h->json_off[n_rec] + 1);
                    h->json_flat.n = (size_t)h->json_off[n_rec];
                    if (!parallel_index_safe((int)h->parts.size(), [&](int k) {
                            const SplitPartF &pt = *h->parts[(size_t)k];
                            if (pt.json.n) memcpy(h->json_flat.p + h->json_off[pt.rec_base], pt.json.p, pt.json.n);
                        }))
                        return DYD_ERR_OOM;
                }
                *data = (const uint8_t *)h->json_flat.p; *off = h->json_off.data();
                return DYD_OK;
            }
            case 1: *data = (const uint8_t *)h->combo.p; *off = h->combo_off.data(); return DYD_OK;
            case 2: *data = (const uint8_t *)h->reasons.p; *off = h->reasons_off.data(); return DYD_OK;
            case 3: {
                if (h->ev_text_off.empty()) {
                    h->ev_text_off.reserve(h->ev_cell.n + 1);
                    h->ev_text_off.push_back(0);
                    for (size_t e = 0; e < h->ev_cell.n; ++e) {
                        const int32_t c = h->ev_code.p[e];
                        if (c >= 0) h->ev_text += h->undef_names[(size_t)c];
                        h->ev_text_off.push_back((int64_t)h->ev_text.size());
                    }
                }
                *data = (const uint8_t *)h->ev_text.data(); *off = h->ev_text_off.data();
                return DYD_OK;
            }
            case 4: *data = (const uint8_t *)h->undef_text.data(); *off = h->undef_off.data(); return DYD_OK;
            default: return DYD_ERR_INVALID;
        }
    } catch (const std::bad_alloc &) {
        return DYD_ERR_OOM;
    }
}
void dyd_split_free(dyd_split *h) { delete h; }

}  // extern "C"

// ===================================================================================================
// label_replace step: object names rewritten through a mapping, the document re-serialised
// (reference core/processor.py:565-609; helpers utils.py:659-679)
// ===================================================================================================
namespace {

enum RelabelStatus : uint8_t {
    RL_REWRITTEN = 0,     // the document is an object whose "objects" is a list: text re-serialised, names replaced
    RL_EMPTY = 1,         // NaN / non-str / "" cell (decided by the caller through `missing`)
    RL_UNDECODABLE = 2,   // json.JSONDecodeError: counted, the cell stays
    RL_UNCHANGED = 3,     // no "objects", or not a list: the cell stays as it is
    RL_IRREGULAR = 5      // the Python path decides (non-dict document, odd "name" values, duplicate keys ...)
};

using NameMap = std::unordered_map<std::string_view, std::string_view>;

struct RelabelPart {      // per-thread outputs, concatenated in cell order afterwards
    int64_t lo = 0, hi = 0;
    std::string text, before, after, tokens;
    std::vector<int64_t> tok_end, tok_cell;
};

struct RelabelCounts {
    int32_t objects = 0, missing_name = 0, labels = 0, replaced_labels = 0, replaced_objects = 0;
};

// one dict element of "objects" (p at '{'): canonical text into `out`, its name replaced when a label is mapped
void relabel_object(Parser &ps, std::string &out, const NameMap &map, int64_t ci, RelabelPart &pt, RelabelCounts &n, bool &any_change,
                    std::string &name, std::vector<std::string> &labels) {
    ++n.objects;
    ++ps.p;
    out += '{';
    KeySet ks;
    bool first = true, has_name = false, name_is_null = false;
    if (ps.peek() == '}') { ++ps.p; out += '}'; ++n.missing_name; return; }
    while (true) {
        ps.ws();
        const Span k = ps.string_token();
        ks.add(ps, k);
        if (!first) out += ", ";
        first = false;
        ps.emit_string(out, k);
        out += ": ";
        ps.ws();
        if (ps.p >= ps.end || *ps.p != ':') ps.bad();
        ++ps.p;
        if (Parser::span_is(k, "name")) {
            has_name = true;
            const Kind kd = kind_of(ps.peek());
            if (kd == K_STRING) {
                ps.ws();
                const Span v = ps.string_token();
                decode_string(ps, v, name);
                if (name.empty()) {                         // falsy: no labels, nothing changes (utils.py:665-666)
                    out += "\"\"";
                } else {
                    split_labels(name, labels);
                    int hits = 0;
                    for (std::string &l : labels) {
                        const auto it = map.find(std::string_view(l));
                        if (it == map.end()) {                  // :593-595
                            pt.tokens += l;
                            pt.tok_end.push_back((int64_t)pt.tokens.size());
                            pt.tok_cell.push_back(ci);
                        } else {
                            l.assign(it->second);
                            ++hits;
                        }
                    }
                    n.labels += (int32_t)labels.size();
                    std::sort(labels.begin(), labels.end());    // sorted(set(...)): code point order == UTF-8 byte order
                    labels.erase(std::unique(labels.begin(), labels.end()), labels.end());
                    std::string fresh;
                    for (size_t i = 0; i < labels.size(); ++i) {
                        if (i) fresh += ',';
                        fresh += labels[i];
                    }
                    if (hits) {                                 // :598-602
                        n.replaced_labels += hits;
                        ++n.replaced_objects;
                    }
                    if (fresh != name) {                        // :603-604 (also when nothing was replaced: the diff shows it,
                        if (any_change) {                       //           the cell keeps the old name)
                            pt.before += "\xef\xbc\x9b";
                            pt.after += "\xef\xbc\x9b";
                        }
                        pt.before += name;
                        pt.after += fresh;
                        any_change = true;
                    }
                    quote_utf8(out, hits ? fresh : name);
                }
            } else if (kd == K_NULL) {
                ps.value(&out);
                name_is_null = true;
            } else if (kd == K_FALSE) {
                ps.value(&out);                                 // falsy and not None: nothing to do
            } else if (kd == K_ARRAY || kd == K_OBJECT) {
                const char open = *ps.p;
                ++ps.p;
                if (ps.peek() != (open == '[' ? ']' : '}')) ps.irregular();   // str(list / dict) as a label: Python path
                ++ps.p;
                out += (open == '[') ? "[]" : "{}";
            } else {
                ps.irregular();                                 // numbers (0 is falsy, 7 -> "7") and true: Python path
            }
        } else {
            ps.value(&out);
        }
        const char d = ps.peek();
        if (d == ',') { ++ps.p; continue; }
        if (d == '}') { ++ps.p; break; }
        ps.bad();
    }
    out += '}';
    if (!has_name || name_is_null) ++n.missing_name;
}

// whole cell.  Throws Fail.  On RL_REWRITTEN the part holds the new text, the joined diff and the unmatched labels.
RelabelStatus relabel_cell(Span cell, int64_t ci, const NameMap &map, RelabelPart &pt, RelabelCounts &n, bool &any_change) {
    Parser ps{cell.b, cell.e};
    ps.ws();
    if (ps.p >= ps.end) ps.bad();
    if (*ps.p != '{') {   // list / scalar document: data.get raises AttributeError and ends the step: Python path
        ps.value(nullptr);
        ps.ws();
        if (ps.p != ps.end) ps.bad();
        ps.irregular();
    }
    ++ps.p;
    std::string &out = pt.text;
    out += '{';
    int objects_kind = -1;   // -1 absent, 0 array, 1 something else
    KeySet ks;
    std::string name;
    std::vector<std::string> labels;
    if (ps.peek() == '}') {
        ++ps.p;
    } else {
        bool first = true;
        while (true) {
            ps.ws();
            const Span k = ps.string_token();
            ks.add(ps, k);
            if (!first) out += ", ";
            first = false;
            ps.emit_string(out, k);
            out += ": ";
            ps.ws();
            if (ps.p >= ps.end || *ps.p != ':') ps.bad();
            ++ps.p;
            if (Parser::span_is(k, "objects")) {
                if (kind_of(ps.peek()) != K_ARRAY) {
                    objects_kind = 1;
                    ps.value(nullptr);
                } else {
                    objects_kind = 0;
                    ++ps.p;
                    out += '[';
                    if (ps.peek() == ']') {
                        ++ps.p;
                    } else {
                        bool first_el = true;
                        while (true) {
                            if (!first_el) out += ", ";
                            first_el = false;
                            if (ps.peek() == '{') relabel_object(ps, out, map, ci, pt, n, any_change, name, labels);
                            else ps.value(&out);     // non-dict elements are kept and not counted (:583-584)
                            const char d = ps.peek();
                            if (d == ',') { ++ps.p; continue; }
                            if (d == ']') { ++ps.p; break; }
                            ps.bad();
                        }
                    }
                    out += ']';
                }
            } else {
                ps.value(&out);
            }
            const char d = ps.peek();
            if (d == ',') { ++ps.p; continue; }
            if (d == '}') { ++ps.p; break; }
            ps.bad();
        }
    }
    ps.ws();
    if (ps.p != ps.end) ps.bad();
    out += '}';
    return objects_kind == 0 ? RL_REWRITTEN : RL_UNCHANGED;
}

}  // namespace

struct dyd_relabel {
    int64_t n_cells = 0;
    std::vector<uint8_t> status, has_diff;
    std::vector<int32_t> counts;          // 5 per cell: objects, missing names, labels, replaced labels, replaced objects
    std::string text, before, after, tokens;
    std::vector<int64_t> text_off, before_off, after_off, tok_off, tok_cell;
};

extern "C" {

// map_keys / map_vals: the old -> new label pairs, concatenated UTF-8 with offsets.  `missing[i]` != 0 marks a cell the
// step skips (NaN / not text / "").  Cells are independent; the caller walks the results in its own (row-major) order.
int dyd_json_relabel(const uint8_t *text, const int64_t *cell_off, const uint8_t *missing, int64_t n_cells,
                     const uint8_t *key_text, const int64_t *key_off, const uint8_t *val_text, const int64_t *val_off,
                     int32_t n_pairs, int n_threads, dyd_relabel **out) {
    if (!out || n_cells < 0 || n_pairs < 0 || (n_cells > 0 && !cell_off) || (n_pairs > 0 && (!key_text || !key_off || !val_text || !val_off)))
        return DYD_ERR_INVALID;
    dyd_relabel *h = new (std::nothrow) dyd_relabel();
    if (!h) return DYD_ERR_OOM;
    try {
        h->n_cells = n_cells;
        h->status.assign((size_t)n_cells, RL_EMPTY);
        h->has_diff.assign((size_t)n_cells, 0);
        h->counts.assign((size_t)n_cells * 5, 0);
        NameMap map;
        map.reserve((size_t)n_pairs * 2 + 1);
        for (int32_t i = 0; i < n_pairs; ++i)   // later pairs overwrite earlier ones, as the dict does
            map[std::string_view((const char *)key_text + key_off[i], (size_t)(key_off[i + 1] - key_off[i]))] =
                std::string_view((const char *)val_text + val_off[i], (size_t)(val_off[i + 1] - val_off[i]));
        std::vector<RelabelPart> parts(64);
        std::vector<int64_t> text_len((size_t)n_cells, 0), before_len((size_t)n_cells, 0), after_len((size_t)n_cells, 0);
        parallel_cells(n_cells, n_threads, [&](int t, int64_t lo, int64_t hi) {
            RelabelPart &pt = parts[(size_t)t];
            pt.lo = lo; pt.hi = hi;
            for (int64_t i = lo; i < hi; ++i) {
                if (missing && missing[i]) continue;
                const size_t m_text = pt.text.size(), m_before = pt.before.size(), m_after = pt.after.size(), m_tok = pt.tokens.size(),
                             m_te = pt.tok_end.size();
                RelabelCounts n;
                bool any_change = false;
                RelabelStatus st;
                try {
                    st = relabel_cell(Span{(const char *)text + cell_off[i], (const char *)text + cell_off[i + 1]}, i, map, pt, n, any_change);
                } catch (Fail f) {
                    st = (f.code == 1) ? RL_UNDECODABLE : RL_IRREGULAR;
                }
                if (st != RL_REWRITTEN) {   // nothing of this cell is kept: its text stays what it was
                    pt.text.resize(m_text); pt.before.resize(m_before); pt.after.resize(m_after); pt.tokens.resize(m_tok);
                    pt.tok_end.resize(m_te); pt.tok_cell.resize(m_te);
                    pt.text.append((const char *)text + cell_off[i], (size_t)(cell_off[i + 1] - cell_off[i]));
                } else {
                    int32_t *c = &h->counts[(size_t)i * 5];
                    c[0] = n.objects; c[1] = n.missing_name; c[2] = n.labels; c[3] = n.replaced_labels; c[4] = n.replaced_objects;
                    h->has_diff[(size_t)i] = any_change ? 1 : 0;
                }
                h->status[(size_t)i] = st;
                text_len[(size_t)i] = (int64_t)(pt.text.size() - m_text);
                before_len[(size_t)i] = (int64_t)(pt.before.size() - m_before);
                after_len[(size_t)i] = (int64_t)(pt.after.size() - m_after);
            }
        });
        std::sort(parts.begin(), parts.end(), [](const RelabelPart &a, const RelabelPart &b) { return a.lo < b.lo; });
        h->tok_off.push_back(0);
        for (auto &pt : parts) {
            const int64_t tb = (int64_t)h->tokens.size();
            h->text += pt.text;
            h->before += pt.before;
            h->after += pt.after;
            h->tokens += pt.tokens;
            for (int64_t e : pt.tok_end) h->tok_off.push_back(tb + e);
            h->tok_cell.insert(h->tok_cell.end(), pt.tok_cell.begin(), pt.tok_cell.end());
        }
        h->text_off.resize((size_t)n_cells + 1);
        h->before_off.resize((size_t)n_cells + 1);
        h->after_off.resize((size_t)n_cells + 1);
        h->text_off[0] = h->before_off[0] = h->after_off[0] = 0;
        for (int64_t i = 0; i < n_cells; ++i) {
            h->text_off[(size_t)i + 1] = h->text_off[(size_t)i] + text_len[(size_t)i];
            h->before_off[(size_t)i + 1] = h->before_off[(size_t)i] + before_len[(size_t)i];
            h->after_off[(size_t)i + 1] = h->after_off[(size_t)i] + after_len[(size_t)i];
        }
    } catch (const std::bad_alloc &) {
        delete h;
        return DYD_ERR_OOM;
    }
    *out = h;
    return DYD_OK;
}

const uint8_t *dyd_relabel_status(const dyd_relabel *h) { return h->status.data(); }
const uint8_t *dyd_relabel_has_diff(const dyd_relabel *h) { return h->has_diff.data(); }
const int32_t *dyd_relabel_counts(const dyd_relabel *h) { return h->counts.data(); }
int64_t dyd_relabel_tokens(const dyd_relabel *h) { return (int64_t)h->tok_cell.size(); }
const int64_t *dyd_relabel_token_cell(const dyd_relabel *h) { return h->tok_cell.data(); }
// which: 0 new text per cell [n_cells], 1 joined old names per cell [n_cells], 2 joined new names per cell [n_cells],
// 3 unmatched labels in order of appearance [tokens]
int dyd_relabel_strings(const dyd_relabel *h, int which, const uint8_t **data, const int64_t **off) {
    if (!h || !data || !off) return DYD_ERR_INVALID;
    switch (which) {
        case 0: *data = (const uint8_t *)h->text.data(); *off = h->text_off.data(); return DYD_OK;
        case 1: *data = (const uint8_t *)h->before.data(); *off = h->before_off.data(); return DYD_OK;
        case 2: *data = (const uint8_t *)h->after.data(); *off = h->after_off.data(); return DYD_OK;
        case 3: *data = (const uint8_t *)h->tokens.data(); *off = h->tok_off.data(); return DYD_OK;
        default: return DYD_ERR_INVALID;
    }
}
void dyd_relabel_free(dyd_relabel *h) { delete h; }

}  // extern "C"

// ===================================================================================================
// YOLO step: labelled boxes of a cell (reference utils.py:681-710) for K7
// ===================================================================================================
namespace {

// first-wins min / max over the number tokens of one coordinate list, CPython semantics on doubles
struct MinMax {
    bool any = false;
    double lo = 0, hi = 0;
    void add(double v) {
        if (!any) { any = true; lo = hi = v; return; }
        if (v < lo) lo = v;
        if (v > hi) hi = v;
    }
};

double coord_value(Parser &ps, Span tok) {
    if (kind_of(*tok.b) != K_NUMBER) ps.irregular();                      // None / str / bool / container: TypeError paths
    const Num n = classify_number(tok);
    if (n.is_int && std::fabs(n.v) > 4503599627370496.0) ps.irregular();  // > 2^52: exact int arithmetic in the label lines
    return n.v;
}

// regular cells only; anything the reference treats through an exception or a non-list container throws Fail{2}
void labelled_cell(Span cell, std::string_view label, std::vector<double> &box4, std::vector<uint8_t> &sel, int32_t &count) {
    count = 0;
    Parser ps{cell.b, cell.e};
    ps.ws();
    if (ps.p >= ps.end) ps.bad();
    if (*ps.p != '{') { ps.value(nullptr); ps.ws(); if (ps.p != ps.end) ps.bad(); ps.irregular(); }
    ++ps.p;
    KeySet ks;
    std::string name;
    if (ps.peek() == '}') { ++ps.p; ps.ws(); if (ps.p != ps.end) ps.bad(); return; }
    while (true) {
        ps.ws();
        const Span k = ps.string_token();
        ks.add(ps, k);
        ps.ws();
        if (ps.p >= ps.end || *ps.p != ':') ps.bad();
        ++ps.p;
        if (!Parser::span_is(k, "objects")) {
            ps.value(nullptr);
        } else {
            if (kind_of(ps.peek()) != K_ARRAY) ps.irregular();
            ++ps.p;
            if (ps.peek() == ']') {
                ++ps.p;
            } else {
                while (true) {
                    if (ps.peek() != '{') {
                        ps.value(nullptr);                      // non-dict objects are skipped (:689)
                    } else {
                        ++ps.p;
                        KeySet oks;
                        bool has_name = false, truthy = false, has_box = false;
                        MinMax xs, ys;
                        if (ps.peek() == '}') {
                            ++ps.p;
                        } else {
                            while (true) {
                                ps.ws();
                                const Span ok = ps.string_token();
                                oks.add(ps, ok);
                                ps.ws();
                                if (ps.p >= ps.end || *ps.p != ':') ps.bad();
                                ++ps.p;
                                if (Parser::span_is(ok, "name")) {
                                    has_name = true;
                                    const Kind kd = kind_of(ps.peek());
                                    if (kd == K_STRING) { ps.ws(); decode_string(ps, ps.string_token(), name); truthy = !name.empty(); }
                                    else if (kd == K_NULL || kd == K_FALSE) ps.value(nullptr);
                                    else ps.irregular();        // numbers, true, containers: never equal to a str label, Python decides
                                } else if (Parser::span_is(ok, "polygon")) {
                                    if (kind_of(ps.peek()) != K_OBJECT) ps.irregular();
                                    ++ps.p;
                                    KeySet pks;
                                    if (ps.peek() == '}') {
                                        ++ps.p;
                                    } else {
                                        while (true) {
                                            ps.ws();
                                            const Span pk = ps.string_token();
                                            pks.add(ps, pk);
                                            ps.ws();
                                            if (ps.p >= ps.end || *ps.p != ':') ps.bad();
                                            ++ps.p;
                                            if (!Parser::span_is(pk, "ptList")) {
                                                ps.value(nullptr);
                                            } else {
                                                if (kind_of(ps.peek()) != K_ARRAY) ps.irregular();
                                                ++ps.p;
                                                if (ps.peek() == ']') {
                                                    ++ps.p;
                                                } else {
                                                    while (true) {
                                                        if (ps.peek() != '{') {
                                                            ps.value(nullptr);      // not a dict: no coordinate (:699-700)
                                                        } else {
                                                            ++ps.p;
                                                            KeySet qks;
                                                            if (ps.peek() == '}') {
                                                                ++ps.p;
                                                            } else {
                                                                while (true) {
                                                                    ps.ws();
                                                                    const Span qk = ps.string_token();
                                                                    qks.add(ps, qk);
                                                                    ps.ws();
                                                                    if (ps.p >= ps.end || *ps.p != ':') ps.bad();
                                                                    ++ps.p;
                                                                    ps.ws();
                                                                    const char *b = ps.p;
                                                                    ps.value(nullptr);
                                                                    if (Parser::span_is(qk, "x")) xs.add(coord_value(ps, Span{b, ps.p}));
                                                                    else if (Parser::span_is(qk, "y")) ys.add(coord_value(ps, Span{b, ps.p}));
                                                                    const char d = ps.peek();
                                                                    if (d == ',') { ++ps.p; continue; }
                                                                    if (d == '}') { ++ps.p; break; }
                                                                    ps.bad();
                                                                }
                                                            }
                                                        }
                                                        const char d = ps.peek();
                                                        if (d == ',') { ++ps.p; continue; }
                                                        if (d == ']') { ++ps.p; break; }
                                                        ps.bad();
                                                    }
                                                }
                                                has_box = xs.any && ys.any;
                                            }
                                            const char d = ps.peek();
                                            if (d == ',') { ++ps.p; continue; }
                                            if (d == '}') { ++ps.p; break; }
                                            ps.bad();
                                        }
                                    }
                                } else {
                                    ps.value(nullptr);
                                }
                                const char d = ps.peek();
                                if (d == ',') { ++ps.p; continue; }
                                if (d == '}') { ++ps.p; break; }
                                ps.bad();
                            }
                        }
                        if (has_name && truthy && has_box) {
                            const double v[4] = {xs.lo, ys.lo, xs.hi, ys.hi};
                            box4.insert(box4.end(), v, v + 4);
                            sel.push_back(std::string_view(name) == label ? 1 : 0);
                            ++count;
                        }
                    }
                    const char d = ps.peek();
                    if (d == ',') { ++ps.p; continue; }
                    if (d == ']') { ++ps.p; break; }
                    ps.bad();
                }
            }
        }
        const char d = ps.peek();
        if (d == ',') { ++ps.p; continue; }
        if (d == '}') { ++ps.p; break; }
        ps.bad();
    }
    ps.ws();
    if (ps.p != ps.end) ps.bad();
}

}  // namespace

extern "C" {

// Labelled boxes for the YOLO step: per cell the (min x, min y, max x, max y) of every named object with a
// non-empty ptList (utils.py:681-710) and whether its name equals the row's label value (processor.py:1006).
// Undecodable cells give no boxes (the reference swallows the error); irregular cells (status 2) are left to
// the Python path.  Results: dyd_scan_xy = box4, dyd_scan_cell_box_off, dyd_scan_status, dyd_scan_sel.
int dyd_json_scan_labelled(const uint8_t *text, const int64_t *cell_off, const uint8_t *missing, int64_t n_cells,
                           const uint8_t *label_text, const int64_t *label_off, int n_threads, dyd_scan **out) {
    if (!out || n_cells < 0 || (n_cells > 0 && (!cell_off || !label_off))) return DYD_ERR_INVALID;
    dyd_scan *h = new (std::nothrow) dyd_scan();
    if (!h) return DYD_ERR_OOM;
    h->n_cells = n_cells;
    try {
        h->status.assign((size_t)n_cells, CELL_OK);
        std::vector<int32_t> counts((size_t)n_cells, 0);
        struct Part { std::vector<double> b; std::vector<uint8_t> s; int64_t lo = 0, hi = 0; };
        std::vector<Part> parts(64);
        parallel_cells(n_cells, n_threads, [&](int t, int64_t lo, int64_t hi) {
            Part &pt = parts[(size_t)t];
            pt.lo = lo; pt.hi = hi;
            for (int64_t i = lo; i < hi; ++i) {
                if (missing && missing[i]) { h->status[(size_t)i] = CELL_MISSING; continue; }
                const size_t mb = pt.b.size(), ms = pt.s.size();
                int32_t c = 0;
                try {
                    labelled_cell(Span{(const char *)text + cell_off[i], (const char *)text + cell_off[i + 1]},
                                  std::string_view((const char *)label_text + label_off[i], (size_t)(label_off[i + 1] - label_off[i])),
                                  pt.b, pt.s, c);
                } catch (Fail f) {
                    pt.b.resize(mb); pt.s.resize(ms);
                    c = 0;
                    h->status[(size_t)i] = (f.code == 1) ? CELL_UNDECODABLE : CELL_IRREGULAR;
                }
                counts[(size_t)i] = c;
            }
        });
        std::sort(parts.begin(), parts.end(), [](const Part &a, const Part &b) { return a.lo < b.lo; });
        size_t tot = 0;
        for (auto &pt : parts) tot += pt.s.size();
        if (tot >= (size_t)1 << 31) { delete h; return DYD_ERR_RANGE; }
        h->xy.reserve(tot * 4);
        h->sel.reserve(tot);
        for (auto &pt : parts) {
            h->xy.insert(h->xy.end(), pt.b.begin(), pt.b.end());
            h->sel.insert(h->sel.end(), pt.s.begin(), pt.s.end());
        }
        h->cell_box_off.resize((size_t)n_cells + 1);
        h->cell_box_off[0] = 0;
        for (int64_t i = 0; i < n_cells; ++i) h->cell_box_off[(size_t)i + 1] = h->cell_box_off[(size_t)i] + counts[(size_t)i];
        h->pt_off.assign(1, 0);
    } catch (const std::bad_alloc &) {
        delete h;
        return DYD_ERR_OOM;
    }
    *out = h;
    return DYD_OK;
}

const uint8_t *dyd_scan_sel(const dyd_scan *h) { return h->sel.data(); }

}  // extern "C"


// ===================================================================================================
// verification of hash equality (dedup / reference filter): the bytes of matched cells
// ===================================================================================================
extern "C" {

// cells a[i] = text_a[off_a[ia]..off_a[ia + 1]) with ia = idx_a ? idx_a[i] : i, likewise b; a pair with a negative index is skipped.
// out_first_mismatch: the smallest i whose cells differ (-1: all pairs are equal); returns the number of differing pairs.
int64_t dyd_host_cells_differ(const uint8_t *text_a, const int64_t *off_a, const int64_t *idx_a, const uint8_t *text_b, const int64_t *off_b,
                              const int64_t *idx_b, int64_t n, int n_threads, uint8_t *out_differs_or_null) {
    if (n <= 0 || !off_a || !off_b) return 0;
    if (n_threads <= 0) n_threads = default_threads();
    std::vector<int64_t> bad((size_t)std::max(1, std::min(n_threads, 64)), 0);
    parallel_cells(n, (int)bad.size(), [&](int t, int64_t lo, int64_t hi) {
        int64_t count = 0;
        for (int64_t i = lo; i < hi; ++i) {
            const int64_t ia = idx_a ? idx_a[i] : i, ib = idx_b ? idx_b[i] : i;
            uint8_t d = 0;
            if (ia >= 0 && ib >= 0) {
                const int64_t la = off_a[ia + 1] - off_a[ia], lb = off_b[ib + 1] - off_b[ib];
                d = (la != lb || (la > 0 && memcmp(text_a + off_a[ia], text_b + off_b[ib], (size_t)la) != 0)) ? 1 : 0;
            }
            if (out_differs_or_null) out_differs_or_null[i] = d;
            count += d;
        }
        bad[(size_t)t] += count;
    });
    int64_t total = 0;
    for (int64_t c : bad) total += c;
    return total;
}

}  // extern "C"
