"""K7 (YOLO label lines, csrc/k7_yolo.hip) through the C ABI against the C oracle (glibc "%.6f"), the Python
restatement and the golden label files written by the reference.  Byte-exact.  Needs a real MI355X."""
import ctypes as C

import numpy as np
import pytest

from conftest import load_golden
from deal_yolo_daya_amd.core import processor as P
from oracle import lib as olib

pytestmark = pytest.mark.gpu

TWO43 = float(2 ** 43)


def _rows(off, text, n):
    return [text[off[i]:off[i + 1]] for i in range(n)]


def _check_against_oracle(native, box4, row_off, sel, w, h, cid):
    off, flag, text = native.yolo_lines(box4, row_off, sel, w, h, cid)
    ooff, oflag, otext = olib.yolo_lines(box4, row_off, sel, w, h, cid)
    n = len(row_off) - 1
    got, want = _rows(off, text, n), _rows(ooff, otext, n)
    host = flag == 2
    assert np.array_equal(flag[~host], oflag[~host])
    assert off[0] == 0 and off[-1] == len(text) and (np.diff(off) >= 0).all()
    for i in range(n):
        if host[i]:
            assert got[i] == b""
            if oflag[i] != 2:                       # left to the host only for a value of 2^43 or more
                vals = [float(t) for line in want[i].split(b"\n") for t in line.split()[1:]]
                assert any(abs(v) >= TWO43 for v in vals if v == v), want[i]
        else:
            assert got[i] == want[i], (i, got[i], want[i])
    return flag


def _random_case(rng, n_rows, max_boxes, with_sel, special=True):
    counts = rng.integers(0, max_boxes + 1, n_rows)
    row_off = np.zeros(n_rows + 1, np.int32)
    np.cumsum(counts, out=row_off[1:])
    nb = int(row_off[-1])
    c = rng.random((nb, 2)) * [1920, 1080]
    d = rng.random((nb, 2)) * 200
    box = np.concatenate([c, c + d], axis=1)
    box = np.where(rng.random((nb, 1)) < 0.5, np.round(box, 0), np.round(box, 2))
    swap = rng.random(nb) < 0.3                                   # corners in any order
    box[swap] = box[swap][:, [2, 3, 0, 1]]
    if special and nb:
        k = rng.integers(0, nb, max(1, nb // 20))
        box[k, rng.integers(0, 4, len(k))] = rng.choice([np.nan, np.inf, -np.inf, 0.0, -0.0, 1e-320, -5.0, 1e13, 3e15], len(k))
        z = rng.integers(0, nb, max(1, nb // 30))
        box[z, 2] = box[z, 0]                                      # zero width: no line
    w = rng.choice([1920.0, 1080.0, 640.0, 1.0, 128.0, 333.0], n_rows)
    h = rng.choice([1080.0, 720.0, 480.0, 1.0, 256.0, 77.0], n_rows)
    if special:
        k = rng.integers(0, n_rows, max(1, n_rows // 25))
        w[k] = rng.choice([0.0, -0.0, np.nan, np.inf, -640.0, 1e-9, 1e300], len(k))
        k = rng.integers(0, n_rows, max(1, n_rows // 25))
        h[k] = rng.choice([0.0, np.nan, -np.inf, 0.5, 1e-12], len(k))
    cid = rng.choice([0, 1, 7, 9, 10, 42, 99, 100, 12345, 2 ** 31 - 1], n_rows).astype(np.int32)
    sel = (rng.random(nb) < 0.7).astype(np.uint8) if with_sel else None
    return box, row_off, sel, w, h, cid


@pytest.fixture(params=[-1, 22, 2, 30], ids=["auto", "paired", "single", "by_box"])
def k7_variant(request, native):
    """22 = two 512-row tiles per ticket, software-pipelined; 2 = one tile per ticket; 30 = tiles of 480 boxes, a lane per
    box; -1 (default) = 22 for tables of one box per row, 30 from 1.02 boxes per row on"""
    native.check(native.lib().dyd_set_option(b"k7_variant", request.param), "opt")
    yield request.param
    native.check(native.lib().dyd_set_option(b"k7_variant", -1), "opt")


@pytest.mark.parametrize("n_rows,max_boxes,with_sel", [(1, 1, False), (255, 3, True), (256, 1, False), (257, 2, True), (512, 1, False),
                                                       (513, 2, True), (1024, 1, False), (1025, 1, True), (5000, 4, True),
                                                       (70001, 1, False), (3000, 40, True)])
def test_k7_random_matches_oracle(native, k7_variant, n_rows, max_boxes, with_sel):
    rng = np.random.default_rng(n_rows * 7 + max_boxes)
    flag = _check_against_oracle(native, *_random_case(rng, n_rows, max_boxes, with_sel))
    if n_rows >= 255:
        assert (flag == 0).any() and (flag == 1).any() and (flag == 2).any()


def test_k7_empty_and_degenerate(native):
    off, flag, text = native.yolo_lines(np.zeros((0, 4)), np.zeros(1, np.int32), None, [], [], [])
    assert off.tolist() == [0] and len(flag) == 0 and text == b""
    off, flag, text = native.yolo_lines(np.zeros((0, 4)), np.zeros(301, np.int32), None, np.ones(300), np.ones(300), np.zeros(300, np.int32))
    assert (off == 0).all() and (flag == 1).all() and text == b""
    # nothing selected
    box = np.array([[0, 0, 2, 2.0]] * 5)
    off, flag, text = native.yolo_lines(box, [0, 5], np.zeros(5, np.uint8), [4.0], [4.0], [3])
    assert flag.tolist() == [1] and text == b""
    off, flag, text = native.yolo_lines(box, [0, 5], np.array([0, 1, 0, 0, 1], np.uint8), [4.0], [4.0], [3])
    assert text == b"3 0.250000 0.250000 0.500000 0.500000\n3 0.250000 0.250000 0.500000 0.500000"
    # negative class id is the host's
    off, flag, text = native.yolo_lines(box, [0, 5], None, [4.0], [4.0], [-1])
    assert flag.tolist() == [2] and text == b""


def _format_cases(native, xs):
    """print x/2 and x through K7 (w = h = 1, box (0, 0, x, x)) and compare with Python's own '%.6f'"""
    xs = np.asarray(xs, np.float64)
    n = len(xs)
    box = np.zeros((n, 4))
    box[:, 2] = xs
    box[:, 3] = xs
    off, flag, text = native.yolo_lines(box, np.arange(n + 1, dtype=np.int32), None, np.ones(n), np.ones(n), np.zeros(n, np.int32))
    rows = _rows(off, text, n)
    bad = []
    for i, x in enumerate(xs.tolist()):
        if not (x > 0):                       # bw <= 0 (or NaN handled elsewhere): no line
            continue
        if flag[i] == 2:
            assert x >= TWO43
            continue
        want = ("0 %.6f %.6f %.6f %.6f" % (x / 2, x / 2, x, x)).encode()
        if rows[i] != want:
            bad.append((x.hex() if hasattr(x, "hex") else x, rows[i], want))
    assert not bad, bad[:5]
    return flag


def test_k7_exact_rounding_ties_and_carries(native, k7_variant):
    xs = []
    for j in range(1, 40):                              # k / 2^j: every exactly representable tie at 6 decimals
        for k in (1, 3, 5, 7, 15625, 15627, 46875, 78125, 999999, 1000001):
            xs.append(k / 2.0 ** j)
    for v in (0.0000005, 0.0000015, 0.0000025, 0.9999995, 0.99999949999, 9.9999995, 99.9999995, 999999.9999995, 1.5e-7, 4.9e-7, 5.1e-7,
              2.0 ** -1074, 2.0 ** -1022, 2.0 ** -200, 2.0 ** -64, 2.0 ** -63, 2.0 ** -65, 2.0 ** -20, 1e-300, 123456.7890125, 0.1, 0.2, 0.3,
              1 / 3, 2 / 3, 2.0 ** 42, 2.0 ** 43 - 0.001, 2.0 ** 43 - 2.0 ** -9, 8796093022207.999, 2.0 ** 43, 2.0 ** 44, 1e15, 1e300):
        xs.extend([v, np.nextafter(v, 0), np.nextafter(v, np.inf)])
    _format_cases(native, xs)


def test_k7_random_bit_patterns(native):
    rng = np.random.default_rng(77)
    n = 400000
    expo = rng.integers(0, 1023 + 46, n).astype(np.uint64)                  # denormals .. 2^46
    expo = np.where(rng.random(n) < 0.7, rng.integers(1023 - 30, 1023 + 44, n).astype(np.uint64), expo)
    bits = (expo << np.uint64(52)) | rng.integers(0, 2 ** 52, n).astype(np.uint64)
    flag = _format_cases(native, bits.view(np.float64))
    assert (flag == 2).any() and (flag == 0).sum() > n // 2


def test_k7_signs_nan_inf(native):
    # negative values come from negative widths; NaN / inf from the coordinates
    box = np.array([[1, 2, 3, 4.0], [1, 2, 3, 4], [np.nan, 2, 3, 4], [1, 2, np.inf, 4], [-np.inf, 2, np.inf, 4], [1e-9, 1, 2e-9, 2]])
    w = np.array([-100.0, 100, 100, 100, 100, -1.0])
    h = np.array([100.0, -np.inf, 100, 100, 100, 1e9])
    off, flag, text = native.yolo_lines(box, np.arange(7, dtype=np.int32), None, w, h, np.arange(6, dtype=np.int32))
    ooff, oflag, otext = olib.yolo_lines(box, np.arange(7, dtype=np.int32), None, w, h, np.arange(6, dtype=np.int32))
    assert text == otext and np.array_equal(off, ooff) and np.array_equal(flag, oflag)
    assert _rows(off, text, 6)[0] == b"0 -0.020000 0.030000 -0.020000 0.020000"
    assert _rows(off, text, 6)[1] == b"1 0.020000 -0.000000 0.020000 -0.000000"
    assert _rows(off, text, 6)[2] == b"2 nan 0.030000 nan 0.020000"
    assert _rows(off, text, 6)[4] == b"4 nan 0.030000 inf 0.020000"
    assert _rows(off, text, 6)[5] == b"5 -0.000000 0.000000 -0.000000 0.000000"


def test_k7_long_rows_bypass_lds(native, k7_variant):
    """tiles whose text exceeds the LDS staging buffer are printed straight to memory"""
    rng = np.random.default_rng(5)
    n_rows = 600
    counts = np.where(np.arange(n_rows) % 97 == 0, 900, rng.integers(0, 3, n_rows))
    row_off = np.zeros(n_rows + 1, np.int32)
    np.cumsum(counts, out=row_off[1:])
    nb = int(row_off[-1])
    c = np.round(rng.random((nb, 2)) * 1000, 1)
    box = np.concatenate([c, c + np.round(rng.random((nb, 2)) * 90 + 1, 1)], axis=1)
    _check_against_oracle(native, box, row_off, None, np.full(n_rows, 1920.0), np.full(n_rows, 1080.0), np.arange(n_rows, dtype=np.int32))


def _boxes(rng, nb, scale=1000.0):
    c = np.round(rng.random((nb, 2)) * scale, 1)
    return np.concatenate([c, c + np.round(rng.random((nb, 2)) * scale * 0.09 + 1, 1)], axis=1)


@pytest.mark.parametrize("shape", ["window_multiple", "empty_runs", "row_over_windows", "text_over_lds", "host_lines", "one_row",
                                   "rows_over_cap"])
def test_k7_box_tiles(native, k7_variant, shape):
    """shapes that sit on the edges of the box-tiled kernel: tiles that end exactly on a window, runs of empty rows (more rows
    than the LDS offset table holds), a row that covers several windows, text beyond the LDS staging area, host lines in the
    middle of a row, everything in one row"""
    rng = np.random.default_rng(len(shape))
    if shape == "window_multiple":
        counts = np.full(480 * 3 // 4, 4)
    elif shape == "empty_runs":
        counts = np.zeros(9000, np.int64)
        counts[[0, 700, 701, 5000, 8999]] = [3, 500, 1, 460, 2]
    elif shape == "row_over_windows":
        counts = np.array([5, 0, 480 * 3 + 17, 0, 0, 2, 480, 1, 479, 1, 448, 3, 31, 1])
    elif shape == "text_over_lds":
        counts = rng.integers(1, 9, 400)
    elif shape == "host_lines":
        counts = rng.integers(0, 12, 1500)
    elif shape == "one_row":
        counts = np.array([2000])
    else:
        counts = np.where(np.arange(4000) % 9 == 0, 1, 0)        # a tile's boxes spread over some 4000 rows
    n_rows = len(counts)
    row_off = np.zeros(n_rows + 1, np.int32)
    np.cumsum(counts, out=row_off[1:])
    nb = int(row_off[-1])
    box = _boxes(rng, nb, 1e12 if shape == "text_over_lds" else 1000.0)
    w, h = np.full(n_rows, 1920.0), np.full(n_rows, 1080.0)
    if shape == "text_over_lds":
        w[:], h[:] = 1.0, -1.0
    if shape == "host_lines":
        k = rng.integers(0, nb, nb // 40)
        box[k, 2] = 1e18                                            # value >= 2^43: the whole row is the host's
        w[rng.integers(0, n_rows, 20)] = 0.0
    cid = rng.choice([0, 5, 17, 99, 100, 4321], n_rows).astype(np.int32)
    sel = (rng.random(nb) < 0.8).astype(np.uint8) if shape in ("host_lines", "empty_runs") else None
    flag = _check_against_oracle(native, box, row_off, sel, w, h, cid)
    if shape == "host_lines":
        assert (flag == 2).sum() > 20 and (flag == 0).sum() > 100


def test_k7_golden_cases_through_the_step(native):
    g = load_golden("yolo_cases.json")
    names = list(g["cases"])
    cases = [g["cases"][n] for n in names]
    stats = {}
    texts, reasons = P.yolo_label_texts([c["json"] for c in cases], [c["label"] for c in cases], [c["class_id"] for c in cases],
                                        [c["width"] for c in cases], [c["height"] for c in cases], None, stats)
    for n, c, t in zip(names, cases, texts):
        assert t == c["text"], n
    assert stats["device_rows"] > 20


def test_k7_dev_capacity_and_measure_mode(native, k7_variant):
    import torch
    L = native.lib()
    n = 5000
    rng = np.random.default_rng(9)
    box, row_off, sel, w, h, cid = _random_case(rng, n, 2, False, special=False)
    want_off, want_flag, want_text = olib.yolo_lines(box, row_off, None, w, h, cid)
    dev = torch.device("cuda:0")
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    d_box, d_off, d_w, d_h, d_cid = t(box.reshape(-1)), t(row_off), t(w), t(h), t(cid)
    d_toff = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    d_flag = torch.zeros(n, dtype=torch.uint8, device=dev)
    total = C.c_int64()
    st = torch.cuda.current_stream().cuda_stream

    def call(text, cap):
        return L.dyd_yolo_lines_dev(d_box.data_ptr(), d_off.data_ptr(), None, d_w.data_ptr(), d_h.data_ptr(), d_cid.data_ptr(), n, int(d_box.shape[0]),
                                    d_toff.data_ptr(), d_flag.data_ptr(), text.data_ptr() if text is not None else None, cap,
                                    C.byref(total), st)
    assert call(None, 0) == 0 and total.value == len(want_text)              # measure only
    assert np.array_equal(d_toff.cpu().numpy(), want_off)
    small = torch.zeros(len(want_text) - 1, dtype=torch.uint8, device=dev)
    assert call(small, small.numel()) == -5 and total.value == len(want_text)    # DYD_ERR_RANGE, size reported
    assert b"too small" in L.dyd_last_error()
    full = torch.full((len(want_text) + 64,), 0x7e, dtype=torch.uint8, device=dev)
    assert call(full, len(want_text)) == 0
    out = full.cpu().numpy()
    assert out[:len(want_text)].tobytes() == want_text and (out[len(want_text):] == 0x7e).all()      # nothing written past the end
    assert np.array_equal(d_flag.cpu().numpy(), want_flag)


def test_k7_full_size_properties(native, k7_variant):
    """20 M single-box rows (the shape of a split sheet): offsets are the running sum of the line lengths,
    the text is lines of five tokens, and sampled rows equal Python's own formatting."""
    import torch
    L = native.lib()
    n = 20_000_000
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(11)
    c = torch.rand((n, 2), generator=g, device=dev, dtype=torch.float64) * torch.tensor([1920.0, 1080.0], device=dev, dtype=torch.float64)
    d = torch.rand((n, 2), generator=g, device=dev, dtype=torch.float64) * 200 + 0.5
    box = torch.round(torch.cat([c, c + d], dim=1) * 100) / 100
    row_off = torch.arange(n + 1, dtype=torch.int32, device=dev)
    w = torch.full((n,), 1920.0, dtype=torch.float64, device=dev)
    h = torch.full((n,), 1080.0, dtype=torch.float64, device=dev)
    cid = (torch.arange(n, device=dev, dtype=torch.int32) % 20)
    toff = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    flag = torch.zeros(n, dtype=torch.uint8, device=dev)
    cap = 48 * n
    text = torch.zeros(cap, dtype=torch.uint8, device=dev)
    total = C.c_int64()
    rc = L.dyd_yolo_lines_dev(box.data_ptr(), row_off.data_ptr(), None, w.data_ptr(), h.data_ptr(), cid.data_ptr(), n, -1, toff.data_ptr(),
                              flag.data_ptr(), text.data_ptr(), cap, C.byref(total), torch.cuda.current_stream().cuda_stream)
    assert rc == 0, L.dyd_last_error()
    assert int(flag.max()) == 0
    lens = toff[1:] - toff[:-1]
    assert int(lens.min()) >= 36 and int(lens.max()) <= 39 and int(toff[-1]) == total.value == int(lens.sum())
    body = text[:total.value]
    assert int((body == 0).sum()) == 0                                             # every byte of the range was written
    assert int((body == ord(" ")).sum()) == 4 * n and int((body == ord(".")).sum()) == 4 * n and int((body == ord("\n")).sum()) == 0
    idx = torch.randint(0, n, (3000,), generator=torch.Generator().manual_seed(3)).tolist() + [0, n - 1, 255, 256, 257]
    hb, ho, hc = box.cpu().numpy(), toff.cpu().numpy(), cid.cpu().numpy()
    ht = body.cpu().numpy()
    for i in idx:
        x1, y1, x2, y2 = hb[i].tolist()
        want = f"{hc[i]} {(x1 + x2) / 2 / 1920.0:.6f} {(y1 + y2) / 2 / 1080.0:.6f} {max(x2 - x1, 0.0) / 1920.0:.6f} {max(y2 - y1, 0.0) / 1080.0:.6f}"
        assert ht[ho[i]:ho[i + 1]].tobytes().decode() == want, i


def test_generate_yolo_datasets_on_the_device(native, tmp_path):
    """the whole step (label files, images, data.yaml, skip records) equals the reference run of the fixture"""
    import test_yolo_host_cpu as host
    got = host.check_yolo_step(tmp_path, None)
    assert sum(got["stats"]["catA"].values()) > 30


def test_k7_signed_values_at_scale(native, k7_variant):
    """coordinates and image sizes of either sign: minus signs change the line lengths row by row, which is what
    the in-tile scan and the look-back carry; 300 k single-box rows against the C oracle, byte for byte"""
    rng = np.random.default_rng(123)
    n = 300_000
    c = np.round(rng.uniform(-600, 600, (n, 2)), 2)
    d = np.round(rng.uniform(0.01, 300, (n, 2)), 2)
    box = np.concatenate([c, c + d], axis=1)
    flip = rng.random(n) < 0.5
    box[flip] = box[flip][:, [2, 3, 0, 1]]
    w = rng.choice([1920.0, -1920.0, 640.0, -333.0, 1e4, -1e-3], n)
    h = rng.choice([1080.0, -1080.0, 77.0, -0.5, 2e5], n)
    cid = rng.integers(0, 150, n).astype(np.int32)
    flag = _check_against_oracle(native, box, np.arange(n + 1, dtype=np.int32), None, w, h, cid)
    assert (flag == 0).sum() > n * 0.9
