#!/usr/bin/env python3
"""Are the native pass's arrays backed by transparent huge pages on this box, and what does freeing them cost?"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def rollup():
    out = {}
    for line in open("/proc/self/smaps_rollup"):
        k = line.split(":")[0]
        if k in ("Rss", "AnonHugePages"):
            out[k] = int(line.split()[1]) // 1024
    return out


def main():
    rows = int(sys.argv[1]) if len(sys.argv) > 1 else 250000
    from deal_yolo_daya_amd import _native, synth, native_json as nj
    _native.lib()
    cells = synth.json_cells(synth.generate(rows, seed=1))
    print("thp:", open("/sys/kernel/mm/transparent_hugepage/enabled").read().strip(), "| before MB", rollup())
    for rep in range(3):
        t0 = time.perf_counter()
        r = nj.replace_iou(cells, 2, 0.98)
        t1 = time.perf_counter()
        m = rollup()
        r.close()
        t2 = time.perf_counter()
        print(f"rep {rep}: pass {t1 - t0:.3f} s, with the handle alive MB {m}, close {t2 - t1:.3f} s, after MB {rollup()}")


if __name__ == "__main__":
    main()
