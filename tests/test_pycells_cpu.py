"""The threaded column builders of csrc/pyhelpers.c (pycells.py) against numpy / CPython itself: values, identity of the objects
handed out, and the reference counts they leave behind.  Host glue only — no device involved."""
import gc
import pickle
import sys

import numpy as np
import pytest

from deal_yolo_daya_amd import pycells

pytestmark = pytest.mark.skipif(not pycells.available(), reason="_dydpy is not built for this interpreter")


def _views(texts):
    raw = [t.encode("utf-8") for t in texts]
    buf = np.frombuffer(b"".join(raw) or b"\0", np.uint8)
    lens = np.array([len(r) for r in raw], np.int64)
    off = np.zeros(len(raw) + 1, np.int64)
    np.cumsum(lens, out=off[1:])
    return (buf.ctypes.data + off[:-1]).astype(np.uint64), lens, buf, off, raw


def _text_class(t):
    """csrc/pyhelpers.c text_class: 1 ASCII, 2 widest character in U+0100..U+FFFF (written by the workers as a 2-byte str), 0 else"""
    if t.isascii():
        return 1
    return 2 if 0xFF < max(map(ord, t)) <= 0xFFFF else 0


def _texts(n, rng, big_every=11):
    out = []
    for i in range(n):
        if i % 97 == 5:
            out.append("中文，标签；" * (i % 5 + 1))
        elif i % 97 == 6:
            out.append(("c%d，c%d，猫%d" % (i, i + 1, i)) * (i % 3 + 1))          # ASCII with BMP separators: the label combos
        elif i % 211 == 9:
            out.append("café crème %d" % i)                                       # Latin-1 only: a 1-byte non-ASCII str (CPython decodes)
        elif i % 211 == 10:
            out.append("emoji \U0001F600 %d ，" % i)                              # beyond the BMP (CPython decodes)
        elif i % 211 == 11:
            out.append("é，\u07ff\u0800\uffff%d" % i)                            # 2- and 3-byte sequences at their boundaries
        elif i % 53 == 7:
            out.append("")
        elif i % big_every == 0:
            out.append(("{\"k\": %d, " % i) * 60)                      # > 512 bytes: beyond pymalloc's small-object sizes
        else:
            out.append("r%d-" % i * (i % 9 + 1))
    return out


def test_strings_from_views_in_any_order_are_real_strs():
    rng = np.random.default_rng(1)
    texts = _texts(120_000, rng)
    ptr, lens, keep, _, raw = _views(texts)
    idx, slot = rng.permutation(len(texts)), rng.permutation(len(texts))
    got = pycells.strings_from_views(ptr, lens, idx)
    assert got.tolist() == [texts[k] for k in idx.tolist()]
    got2 = pycells.strings_from_views(ptr, lens, None, slot=slot)
    want = [None] * len(texts)
    for i, k in enumerate(slot.tolist()):
        want[k] = texts[i]
    assert got2.tolist() == want
    got3 = pycells.strings_from_views(ptr, lens, idx, slot=slot)
    want3 = [None] * len(texts)
    for i, k in enumerate(slot.tolist()):
        want3[k] = texts[idx[i]]
    assert got3.tolist() == want3
    # they behave like the strings CPython makes: hash, size, concatenation (in-place resize when unshared), pickling, dict keys
    for k in (0, 11, 22, 5, 102, len(texts) - 1):
        s, t = got2[slot[k]], texts[k]
        assert type(s) is str and s == t and hash(s) == hash(t) and sys.getsizeof(s) == sys.getsizeof(t)
        assert s.encode("utf-8") == raw[k] and pickle.loads(pickle.dumps(s)) == t and s.isascii() == t.isascii()
    grown = pycells.strings_from_views(ptr, lens, np.array([0, 11, 22] * 30000, np.int64))
    acc = grown[1]
    grown = None
    acc += "tail"                                                     # refcount 1: resized in place
    assert acc == texts[11] + "tail"
    d = {s: i for i, s in enumerate(got.tolist()[:5000])}
    assert all(d[texts[idx[i]]] >= 0 for i in range(0, 5000, 37))
    ascii_ix = np.flatnonzero(np.array([t.isascii() for t in texts]))
    got4 = pycells.strings_from_views(ptr, lens, ascii_ix, all_ascii=True)
    assert got4.tolist() == [texts[k] for k in ascii_ix.tolist()]
    del got, got2, got3, got4, d, acc
    gc.collect()


def test_strings_from_flat_buffers_with_missing_cells():
    rng = np.random.default_rng(2)
    texts = _texts(70_000, rng, big_every=3)
    _, _, buf, off, _ = _views(texts)
    na = (rng.random(len(texts)) < 0.1).astype(np.uint8)
    got = pycells.strings(buf, off, na)
    assert got.tolist() == [None if m else t for t, m in zip(texts, na.tolist())]
    assert pycells.strings(buf, off).tolist() == texts


def test_a_used_output_array_is_refused_and_left_alone():
    texts = ["x" * 40 + str(i) for i in range(5000)]
    ptr, lens, keep, _, _ = _views(texts)
    from deal_yolo_daya_amd import pycells as pc
    out = np.empty(len(texts), object)
    out[17] = "occupied"
    with pytest.raises(ValueError):
        pc._dydpy.map_strs(ptr.ctypes.data, lens.ctypes.data, 0, 0, len(texts), out.ctypes.data, 4, 1)
    assert out[17] == "occupied" and out[16] is None


def test_object_take_counts_every_reference_once():
    rng = np.random.default_rng(3)
    vals = np.array([f"s{i}" for i in range(50_000)] + [None, 1.5, ("t",)], object)
    row = np.sort(rng.integers(0, len(vals), 400_000))
    slot = rng.permutation(len(row))
    before = [sys.getrefcount(vals[k]) for k in (5, 49_999, len(vals) - 1)]
    none_before = sys.getrefcount(None)
    a = pycells.take(vals, row)
    b = pycells.take(vals, row, slot=slot)
    want = vals[row]
    want_b = np.empty(len(row), object)
    want_b[slot] = want
    same = all(x is y for x, y in zip(a, want)) and all(x is y for x, y in zip(b, want_b))
    assert same
    for k, r0 in zip((5, 49_999, len(vals) - 1), before):
        delta = sys.getrefcount(vals[k]) - r0
        assert delta == 4 * int((row == k).sum())       # a, b, want, want_b
    del a, b, want, want_b
    gc.collect()
    assert [sys.getrefcount(vals[k]) for k in (5, 49_999, len(vals) - 1)] == before
    assert abs(sys.getrefcount(None) - none_before) < 50
    with pytest.raises(IndexError):
        pycells.take(vals, np.full(70_000, len(vals), np.int64))


def test_small_table_take_and_numeric_take_and_slots():
    rng = np.random.default_rng(4)
    table = np.array([f"lab{i}" for i in range(20)] + [None], object)
    codes = rng.integers(0, len(table), 300_000).astype(np.int32)
    idx, slot = rng.permutation(len(codes)), rng.permutation(len(codes))
    r0 = sys.getrefcount(table[3])
    a = pycells.take_small(table, codes, idx, slot=slot)
    want = np.empty(len(codes), object)
    want[slot] = table[codes[idx]]
    same = all(x is y for x, y in zip(a, want))
    delta = sys.getrefcount(table[3]) - r0
    assert same and delta == 2 * int((codes == 3).sum())
    del a, want
    delta = sys.getrefcount(table[3]) - r0
    assert delta == 0
    with pytest.raises(IndexError):
        pycells.take_small(table, np.full(70_000, 99, np.int32))
    for dtype in (np.float64, np.int64, np.int32, np.uint8, np.bool_, np.float32, np.int16):
        v = (rng.random(90_000) * 100).astype(dtype)
        i2 = rng.integers(0, len(v), 200_000)
        s2 = rng.permutation(len(i2))
        assert np.array_equal(pycells.take(v, i2), v[i2])
        w = np.empty(len(i2), dtype)
        w[s2] = v[i2]
        assert np.array_equal(pycells.take(v, i2, slot=s2), w)
    cat = rng.integers(0, 3, 250_000).astype(np.int32)
    sizes = np.bincount(cat, minlength=3)
    off = np.zeros(4, np.int64)
    np.cumsum(sizes, out=off[1:])
    pos = np.empty(len(cat), np.int64)
    for c in range(3):
        m = np.flatnonzero(cat == c)
        pos[m] = rng.permutation(len(m))
    slots = pycells.category_slots(cat, pos, off)
    assert np.array_equal(slots, off[cat] + pos) and np.array_equal(np.sort(slots), np.arange(len(cat)))


def test_gathered_text_for_arrow_columns():
    rng = np.random.default_rng(5)
    texts = _texts(80_000, rng)
    ptr, lens, keep, _, raw = _views(texts)
    idx, slot = rng.permutation(len(texts)), rng.permutation(len(texts))
    data, off = pycells.gather_text(ptr, lens, idx)
    assert bytes(data) == b"".join(raw[k] for k in idx.tolist()) and off[-1] == len(data)
    data, off = pycells.gather_text(ptr, lens, None, slot=slot)
    order = np.empty(len(texts), np.int64)
    order[slot] = np.arange(len(texts))
    assert bytes(data) == b"".join(raw[k] for k in order.tolist())
    assert np.array_equal(np.diff(off), lens[order])


def test_strings_allocated_first_and_written_later():
    """alloc_strings / fill_strings: the two halves of strings_from_views, with the places known only at the second"""
    rng = np.random.default_rng(6)
    texts = _texts(90_000, rng)
    ptr, lens, keep, _, raw = _views(texts)
    seq, asc = pycells.alloc_strings(ptr, lens)
    assert asc.tolist() == [_text_class(t) for t in texts]
    slot = rng.permutation(len(texts))
    out = np.empty(len(texts), object)
    got = pycells.fill_strings(ptr, lens, seq, asc, slot, out)
    want = [None] * len(texts)
    for i, k in enumerate(slot.tolist()):
        want[k] = texts[i]
    assert got is out and out.tolist() == want and all(v is None for v in seq.tolist())
    for k in (0, 5, 11, 102):
        s = out[slot[k]]
        assert type(s) is str and hash(s) == hash(texts[k]) and sys.getsizeof(s) == sys.getsizeof(texts[k]) and sys.getrefcount(s) == 3
    with pytest.raises(ValueError):                                   # the strings have moved on
        pycells.fill_strings(ptr, lens, seq, asc, slot, np.empty(len(texts), object))
    used = np.empty(len(texts), object)
    used[3] = "occupied"
    seq2, asc2 = pycells.alloc_strings(ptr, lens)
    with pytest.raises(ValueError):
        pycells.fill_strings(ptr, lens, seq2, asc2, slot, used)
    assert pycells.fill_strings(ptr, lens, seq2, asc2).tolist() == texts     # in place
    ascii_ix = np.flatnonzero(np.array([t.isascii() for t in texts]))
    seq3, none = pycells.alloc_strings(ptr[ascii_ix], lens[ascii_ix], all_ascii=True)
    assert none is None and pycells.fill_strings(ptr[ascii_ix], lens[ascii_ix], seq3).tolist() == [texts[k] for k in ascii_ix.tolist()]
    seq4, asc4 = pycells.alloc_strings(ptr, lens)                     # released unwritten: nothing reads the text of a dying str
    del seq4, asc4
    gc.collect()


def test_small_strings_from_prepared_arenas(monkeypatch):
    """the builders take pymalloc's new arenas from a slab they mapped, advised as huge pages and touched on all cores
    (csrc/pyhelpers.c prefault_begin; from 4 MB of strings on, here from the first byte): the strings are the same strings, the
    default arena allocator is back afterwards (ordinary objects come and go), freeing the strings unmaps the slab's arenas like
    any others, and the address space does not grow over repeated rounds"""
    import resource

    monkeypatch.setenv("DYD_PREFAULT_MIN_MB", "0")
    rng = np.random.default_rng(9)
    texts = _texts(150_000, rng, big_every=13)
    ptr, lens, keep, _, raw = _views(texts)
    slot = rng.permutation(len(texts))

    def vm_size():
        for line in open("/proc/self/status"):
            if line.startswith("VmSize:"):
                return int(line.split()[1])
        return 0

    sizes = []
    for rnd in range(6):
        monkeypatch.setenv("DYD_PREFAULT", "0" if rnd == 3 else "1")
        a = pycells.strings_from_views(ptr, lens, None, slot=slot)
        seq, asc = pycells.alloc_strings(ptr, lens)
        b = pycells.fill_strings(ptr, lens, seq, asc)
        want = [None] * len(texts)
        for i, k in enumerate(slot.tolist()):
            want[k] = texts[i]
        assert a.tolist() == want and b.tolist() == texts
        junk = [str(i) * 3 for i in range(200_000)]                     # ordinary allocations after the builders: pymalloc's own arenas
        assert junk[777] == "777777777"
        del a, b, seq, junk, want
        gc.collect()
        sizes.append(vm_size())
    assert max(sizes[2:]) - min(sizes[2:]) < 400_000                    # kB: no slab is left mapped round after round
    assert resource.getrusage(resource.RUSAGE_SELF).ru_maxrss > 0


def test_malformed_utf8_raises_what_cpython_raises():
    """texts whose class says "2-byte str" but whose bytes are not UTF-8 (truncated sequence, surrogate, overlong form, stray
    continuation byte): the builders let CPython decode that text, i.e. raise its UnicodeDecodeError, and leave nothing behind"""
    good = ["标签，c%d" % i for i in range(70_000)]
    for bad in (b"\xe4\xb8", b"ok\xed\xa0\x80\xe4\xb8\xad", b"\xe0\x80\xaf\xe4\xb8\xad", b"\xe4\xb8\xad\x80x", b"\xe4\xb8\xad\xc4"):
        raw = [t.encode("utf-8") for t in good]
        raw[31_337] = bad
        buf = np.frombuffer(b"".join(raw), np.uint8)
        lens = np.array([len(r) for r in raw], np.int64)
        off = np.zeros(len(raw) + 1, np.int64)
        np.cumsum(lens, out=off[1:])
        ptr = (buf.ctypes.data + off[:-1]).astype(np.uint64)
        with pytest.raises(UnicodeDecodeError):
            bad.decode("utf-8")
        with pytest.raises(UnicodeDecodeError):
            pycells.strings_from_views(ptr, lens)
        with pytest.raises(UnicodeDecodeError):
            pycells.strings(buf, off)
        try:
            seq, asc = pycells.alloc_strings(ptr, lens)
        except UnicodeDecodeError:
            continue                                                      # (classified 0: CPython decoded it at once)
        with pytest.raises(UnicodeDecodeError):
            pycells.fill_strings(ptr, lens, seq, asc)
        del seq
        gc.collect()


def test_builders_from_several_python_threads(monkeypatch):
    """two Python threads building strings at once (the GIL changes hands while pages are touched and while workers copy): one of
    them gets the prepared arenas, the other allocates the ordinary way; both get their strings, and ordinary allocations by a third
    thread go on meanwhile"""
    import threading

    monkeypatch.setenv("DYD_PREFAULT_MIN_MB", "0")
    rng = np.random.default_rng(10)
    texts = _texts(200_000, rng, big_every=17)
    ptr, lens, keep, _, raw = _views(texts)
    results, errors, stop = {}, [], threading.Event()

    def build(tag):
        try:
            for _ in range(3):
                a = pycells.strings_from_views(ptr, lens)
                seq, asc = pycells.alloc_strings(ptr, lens)
                b = pycells.fill_strings(ptr, lens, seq, asc)
                results[tag] = a.tolist() == texts and b.tolist() == texts
                del a, b, seq
        except BaseException as e:  # noqa: BLE001
            errors.append(e)

    def churn():
        while not stop.is_set():
            junk = [("k%d" % i) * 4 for i in range(20_000)]
            del junk

    third = threading.Thread(target=churn)
    third.start()
    workers = [threading.Thread(target=build, args=(k,)) for k in range(2)]
    for t in workers:
        t.start()
    for t in workers:
        t.join()
    stop.set()
    third.join()
    assert not errors and results == {0: True, 1: True}
