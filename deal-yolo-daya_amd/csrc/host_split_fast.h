// host_split_fast.h — the split step's expansion at table scale (included by host_json.cpp inside its anonymous namespace,
// after split_cell).
//
// split_cell (the exact walker) parses a cell through std::string appends, one ObjText per object and a vector of label
// strings per object; its caller then concatenated every thread's output on ONE thread.  At configs[2] scale the split step
// expands 1 M rows into ~15 M records (2.4 GB of JSON text): that serial tail and the per-object allocations were most of
// the native time.  This lane keeps the exact walker for every cell it does not reproduce byte for byte and changes the rest:
//
//   * ONE parse per cell with the single-parse machinery of host_json_fast.h (FastCell::value<true> writes the canonical
//     json.dumps(..., ensure_ascii=False) text); the document's members other than "objects" and every dict element of
//     "objects" land in a per-thread scratch buffer, each object with the place where its name's value was cut out;
//   * records are memcpy'd together from those spans straight into the thread's part buffer — and STAY there: the handle
//     hands out one (address, length) view per record (dyd_split_rec_views), so the 2.4 GB are written once and read once
//     (by the str / Arrow builders), never gathered;
//   * the fixed-width per-record / per-event arrays are gathered by all threads at once;
//   * undefined labels are interned (per thread, merged at the end): an event carries a code into a table of distinct
//     labels, so the Python side builds one reason text per DISTINCT label instead of sorting a million strings;
//   * per label of the rules: the first record carrying it and the number of records — what the host needs for the
//     first-appearance order of the categories (reference processor.py:773, dict insertion order) without a pass over the records.
//
// Bails (-> exact walker, which also decides undecodable / irregular): escaped or repeated keys, a "name" that is neither a
// plain string nor null / false, anything FastCell::value bails on.
#pragma once

struct SplitObjF {
    uint32_t b = 0, name_at = 0, e = 0;      // span in the scratch buffer; where the name's value goes
    const char *nb = nullptr, *ne = nullptr; // raw name (a plain string: raw == decoded), null when absent / falsy
    bool has_name = false;
};

struct SplitPartF {       // per-thread outputs; records / events in cell order
    int64_t lo = 0, hi = 0;
    Raw<char> json;                  // record texts, back to back
    Raw<int64_t> json_end;           // per record: end inside json
    Raw<int64_t> row_cell;
    Raw<int32_t> row_label;
    Raw<char> combo, reasons;        // per cell texts, back to back (lengths in the handle's per-cell arrays)
    Raw<int64_t> ev_cell;
    Raw<uint8_t> ev_kind;
    Raw<int32_t> ev_code;            // EV_UNDEFINED: local code of the label (remapped to the global table afterwards), else -1
    std::map<std::string, int32_t, std::less<>> undef_ix;
    std::vector<std::string> undef_names;
    std::vector<int64_t> label_first;   // per label of the rules: first LOCAL record carrying it (-1 none)
    std::vector<int64_t> label_count;
    int64_t fast_cells = 0;
    bool ascii = true;               // every record text of the part is pure ASCII
    size_t rec_base = 0, ev_base = 0;
};

struct SplitLane {        // per-thread scratch, reused from cell to cell
    FastPart S;
    std::string tmp;
    std::vector<SplitObjF> objs;
    std::vector<std::string_view> labs;      // labels of all objects, object after object
    std::vector<uint32_t> lab_off;           // per object: its first label in labs (objs.size() + 1 entries)
    std::deque<std::string> own;             // labels of names with non-ASCII text (split_labels copies them)
    std::vector<std::string> own_tmp;
    std::vector<std::string_view> all, undef;
    SplitPart slow;                          // the exact walker's outputs for one cell
};

struct FastSplitCell : FastCell {
    SplitLane &L;
    uint32_t head_a_e = 0, head_b_b = 0;     // head = seg[0, head_a_e) + seg[head_b_b, end)
    int objects_kind = -1;                   // -1 absent, 0 array, 1 something else
    int64_t n_elements = 0;

    FastSplitCell(const char *b, const char *e, SplitLane &l) : FastCell(b, e, l.S, l.tmp), L(l) {}

    bool object_elem() {   // p at '{': one dict element of "objects"
        SplitObjF o;
        o.b = (uint32_t)A.seg.n;
        ++p;
        putc('{');
        bool first = true;
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
                if (ke - kb == 4 && !memcmp(kb, "name", 4)) {
                    ws();
                    if (p >= end) return false;
                    o.has_name = true;
                    o.name_at = (uint32_t)A.seg.n;
                    if (*p == '"') {
                        const char *b, *e;
                        if (!plain_string(b, e)) return false;
                        if (e > b) { o.nb = b; o.ne = e; }
                    } else if (*p == 'n') {
                        if (!literal("null", 4)) return false;
                    } else if (*p == 'f') {
                        if (!literal("false", 5)) return false;
                    } else {
                        return false;          // numbers, true, containers as a name: the exact walker / the Python path
                    }
                } else if (!value<true>()) {
                    return false;
                }
                ws();
                if (p < end && *p == ',') { ++p; continue; }
                if (p < end && *p == '}') { ++p; break; }
                return false;
            }
        }
        putc('}');
        o.e = (uint32_t)A.seg.n;
        L.objs.push_back(o);
        return true;
    }

    bool document() {
        ws();
        if (p >= end || *p != '{') return false;
        ++p;
        bool first = true;
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
                ws();
                if (p >= end || *p != ':') return false;
                ++p;
                if (ke - kb == 7 && !memcmp(kb, "objects", 7)) {
                    ws();
                    if (p >= end) return false;
                    if (*p != '[') {
                        objects_kind = 1;
                        if (!value<false>()) return false;
                    } else {
                        objects_kind = 0;
                        head_a_e = (uint32_t)A.seg.n;
                        ++p;
                        ws();
                        if (p < end && *p == ']') {
                            ++p;
                        } else {
                            while (true) {
                                ws();
                                ++n_elements;
                                if (p < end && *p == '{') {
                                    if (!object_elem()) return false;
                                } else if (!value<false>()) {   // non-dict elements are skipped (:742)
                                    return false;
                                }
                                ws();
                                if (p < end && *p == ',') { ++p; continue; }
                                if (p < end && *p == ']') { ++p; break; }
                                return false;
                            }
                        }
                        head_b_b = (uint32_t)A.seg.n;
                    }
                } else {
                    if (!first) lit(", ");
                    first = false;
                    put(kb - 1, (size_t)(ke - kb) + 2);
                    lit(": ");
                    if (!value<true>()) return false;
                }
                ws();
                if (p < end && *p == ',') { ++p; continue; }
                if (p < end && *p == '}') { ++p; break; }
                return false;
            }
        }
        ws();
        if (p != end) return false;
        if (A.seg.n >= ((size_t)1 << 32)) return false;
        return true;
    }
};

// re.split(r"[,，;；|]", name) of a pure-ASCII plain string: the tokens, blanks stripped (a plain JSON string holds no control
// character, so ' ' is the only str.strip() candidate), empty ones dropped (utils.py:659-662)
inline void split_labels_ascii(const char *b, const char *e, std::vector<std::string_view> &out) {
    const char *tb = b;
    for (const char *q = b;; ++q) {
        if (q == e || *q == ',' || *q == ';' || *q == '|') {
            const char *a = tb, *z = q;
            while (a < z && *a == ' ') ++a;
            while (z > a && z[-1] == ' ') --z;
            if (z > a) out.emplace_back(a, (size_t)(z - a));
            if (q == e) break;
            tb = q + 1;
        }
    }
}

inline bool all_ascii(const char *b, const char *e) {
    uint64_t acc = 0;
    for (; b + 8 <= e; b += 8) { uint64_t v; memcpy(&v, b, 8); acc |= v; }
    for (; b < e; ++b) acc |= (unsigned char)*b;
    return !(acc & 0x8080808080808080ull);
}

inline void split_event(SplitPartF &pt, int64_t ci, uint8_t kind, std::string_view label) {
    pt.ev_cell.push(ci);
    pt.ev_kind.push(kind);
    int32_t code = -1;
    if (kind == EV_UNDEFINED) {
        auto it = pt.undef_ix.find(label);
        if (it == pt.undef_ix.end()) {
            code = (int32_t)pt.undef_names.size();
            pt.undef_names.emplace_back(label);
            pt.undef_ix.emplace(std::string(label), code);
        } else {
            code = it->second;
        }
    }
    pt.ev_code.push(code);
}

inline void split_record_mark(SplitPartF &pt, int64_t ci, int32_t label) {
    if (pt.label_first[(size_t)label] < 0) pt.label_first[(size_t)label] = (int64_t)pt.json_end.n;
    ++pt.label_count[(size_t)label];
    pt.json_end.push((int64_t)pt.json.n);
    pt.row_cell.push(ci);
    pt.row_label.push(label);
}

// one cell through the lane.  false: bail, nothing was written to pt.
inline bool split_cell_fast(Span cell, int64_t ci, const LabelMap &map, SplitLane &L, SplitPartF &pt, uint8_t &status, int32_t &n_out,
                            int64_t &combo_len, int64_t &reasons_len) {
    L.S.seg.n = 0;
    L.objs.clear();
    FastSplitCell fc(cell.b, cell.e, L);
    if (!fc.document()) return false;
    n_out = 0;
    combo_len = reasons_len = 0;
    if (fc.objects_kind == 1) { status = SP_NOT_A_LIST; return true; }
    if (fc.n_elements == 0) { status = SP_NO_OBJECTS; return true; }
    status = SP_OK;
    const char *seg = L.S.seg.p;
    const size_t seg_n = L.S.seg.n;

    L.labs.clear(); L.lab_off.clear(); L.own.clear(); L.all.clear(); L.undef.clear();
    for (const SplitObjF &o : L.objs) {
        L.lab_off.push_back((uint32_t)L.labs.size());
        if (!o.nb) continue;
        if (all_ascii(o.nb, o.ne)) {
            split_labels_ascii(o.nb, o.ne, L.labs);
        } else {
            split_labels(std::string(o.nb, o.ne), L.own_tmp);
            for (std::string &s : L.own_tmp) { L.own.push_back(std::move(s)); L.labs.emplace_back(L.own.back()); }
        }
    }
    L.lab_off.push_back((uint32_t)L.labs.size());
    // "，".join(sorted(raw_label_set)) (:731-736)
    L.all.assign(L.labs.begin(), L.labs.end());
    std::sort(L.all.begin(), L.all.end());
    L.all.erase(std::unique(L.all.begin(), L.all.end()), L.all.end());
    const size_t combo_mark = pt.combo.n;
    for (size_t i = 0; i < L.all.size(); ++i) {
        if (i) pt.combo.put("\xef\xbc\x8c", 3);
        pt.combo.put(L.all[i].data(), L.all[i].size());
    }
    combo_len = (int64_t)(pt.combo.n - combo_mark);

    const size_t head_a = fc.head_a_e, head_b = seg_n - fc.head_b_b, head_len = head_a + head_b;
    static const char kObjects[] = "\"objects\": [";
    for (size_t i = 0; i < L.objs.size(); ++i) {
        const SplitObjF &o = L.objs[i];
        const uint32_t l0 = L.lab_off[i], l1 = L.lab_off[i + 1];
        if (l0 == l1) { split_event(pt, ci, EV_NO_NAME, std::string_view()); continue; }   // "标注框缺少name字段" (:744-749)
        for (uint32_t k = l0; k < l1; ++k) {
            const std::string_view label = L.labs[k];
            const auto it = map.find(label);
            if (it == map.end()) {                                                        // :752-758
                split_event(pt, ci, EV_UNDEFINED, label);
                L.undef.push_back(label);
                continue;
            }
            // {head, "objects": [before "label" after]}   (:760-767)
            const size_t before = o.name_at - o.b, after = o.e - o.name_at;
            pt.json.need(head_len + before + after + label.size() + 32);
            char *w = pt.json.p + pt.json.n;
            *w++ = '{';
            if (head_a) { memcpy(w, seg, head_a); w += head_a; }
            if (head_b) { memcpy(w, seg + fc.head_b_b, head_b); w += head_b; }
            if (head_len) { *w++ = ','; *w++ = ' '; }
            memcpy(w, kObjects, sizeof(kObjects) - 1); w += sizeof(kObjects) - 1;
            memcpy(w, seg + o.b, before); w += before;
            *w++ = '"';
            memcpy(w, label.data(), label.size()); w += label.size();
            *w++ = '"';
            memcpy(w, seg + o.name_at, after); w += after;
            *w++ = ']'; *w++ = '}';
            pt.json.n = (size_t)(w - pt.json.p);
            split_record_mark(pt, ci, it->second);
            ++n_out;
        }
    }
    std::sort(L.undef.begin(), L.undef.end());
    L.undef.erase(std::unique(L.undef.begin(), L.undef.end()), L.undef.end());
    const size_t reasons_mark = pt.reasons.n;
    for (size_t i = 0; i < L.undef.size(); ++i) {   // "；".join(sorted(row_reason_set)) (:779, :790)
        if (i) pt.reasons.put("\xef\xbc\x9b", 3);
        pt.reasons.put("\xe6\xa0\x87\xe7\xad\xbe", 6);                                             // 标签
        pt.reasons.put(L.undef[i].data(), L.undef[i].size());
        pt.reasons.put("\xe6\x9c\xaa\xe5\x9c\xa8\xe8\xa7\x84\xe5\x88\x99\xe4\xb8\xad\xe5\xae\x9a\xe4\xb9\x89", 21);   // 未在规则中定义
    }
    reasons_len = (int64_t)(pt.reasons.n - reasons_mark);
    if (n_out == 0) split_event(pt, ci, EV_NOTHING_CLASSIFIED, std::string_view());
    return true;
}

// a cell the lane bailed on: the exact walker, its outputs appended to the part.  Throws Fail like split_cell.
inline void split_cell_slow(Span cell, int64_t ci, const LabelMap &map, SplitLane &L, SplitPartF &pt, uint8_t &status, int32_t &n_out,
                            int64_t &combo_len, int64_t &reasons_len) {
    SplitPart &s = L.slow;
    s.json.clear(); s.ev_text.clear(); s.combo.clear(); s.reasons.clear();
    s.json_end.clear(); s.row_cell.clear(); s.ev_cell.clear(); s.ev_text_end.clear(); s.row_label.clear(); s.ev_kind.clear();
    n_out = 0;
    combo_len = reasons_len = 0;
    status = split_cell(cell, ci, map, s, n_out);     // throws before anything reaches pt
    size_t prev = 0;
    for (size_t r = 0; r < s.json_end.size(); ++r) {
        const size_t e = (size_t)s.json_end[r];
        pt.json.put(s.json.data() + prev, e - prev);
        prev = e;
        split_record_mark(pt, ci, s.row_label[r]);
    }
    size_t tprev = 0;
    for (size_t k = 0; k < s.ev_cell.size(); ++k) {
        const size_t e = (size_t)s.ev_text_end[k];
        split_event(pt, ci, s.ev_kind[k], std::string_view(s.ev_text.data() + tprev, e - tprev));
        tprev = e;
    }
    pt.combo.put(s.combo.data(), s.combo.size());
    pt.reasons.put(s.reasons.data(), s.reasons.size());
    combo_len = (int64_t)s.combo.size();
    reasons_len = (int64_t)s.reasons.size();
}
