"""The str builders alone (pycells.strings_from_views / alloc_strings) with and without prepared arenas (DYD_PREFAULT), one JSON line
per run: 14 M strings of 300 bytes (the split step's records) and 1 M of 2.5 KB (beyond pymalloc: unaffected)."""
import sys, time, os, json
sys.path.insert(0, ".")
import numpy as np
from deal_yolo_daya_amd import pycells
def bench(size, n, tag):
    buf = np.full(size, 97, np.uint8)
    ptr = np.full(n, buf.ctypes.data, np.uint64); lens = np.full(n, size, np.int64)
    for env in ("1", "0", "1", "0"):
        os.environ["DYD_PREFAULT"] = env
        a = time.perf_counter(); out = pycells.strings_from_views(ptr, lens, all_ascii=True); dt = time.perf_counter() - a
        assert out[0] == "a" * size and out[-1] == "a" * size
        b = time.perf_counter(); del out; fr = time.perf_counter() - b
        a2 = time.perf_counter(); seq, _ = pycells.alloc_strings(ptr, lens, all_ascii=True); d2 = time.perf_counter() - a2
        pycells.fill_strings(ptr, lens, seq); assert seq[5] == "a" * size
        b2 = time.perf_counter(); del seq; f2 = time.perf_counter() - b2
        print(json.dumps({"what": tag, "prefault": env, "builder_s": round(dt, 3), "free_s": round(fr, 3), "alloc_strings_s": round(d2, 3), "free2_s": round(f2, 3)}), flush=True)
bench(300, 14_000_000, "300 B x 14M (the split step's records)")
bench(2500, 1_000_000, "2.5 KB x 1M (the replace step's bbox column)")
