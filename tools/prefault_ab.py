"""A/B of the prepared arenas (csrc/pyhelpers.c prefault_begin) inside the real steps, on the GPU box: replace_and_filter_frame and
split_frames on a 1 M-row table with DYD_PREFAULT flipped between runs of one process, then a cProfile of split_frames."""
import sys, time, json, os, cProfile, pstats
sys.path.insert(0, ".")
import numpy as np, pandas as pd, torch
from deal_yolo_daya_amd import synth
from deal_yolo_daya_amd.core import processor as P
dev = torch.device("cuda", 0)
parts = []
for ci in range(2):
    t = synth.table_from_device(synth.generate_device(500_000, synth.SEED + 77 + ci, dev))
    parts.append(pd.DataFrame({"source": synth.urls(t), synth.ANN_COL: synth.json_cells(t)}))
df = pd.concat(parts, ignore_index=True); del parts
rules = synth.rules()
for env in ("1", "0", "1", "0"):
    os.environ["DYD_PREFAULT"] = env
    st = {}
    a = time.perf_counter(); kept, excluded, high, other = P.replace_and_filter_frame(df, 2, 0.98, stats=st); dt = time.perf_counter() - a
    print(json.dumps({"what": "replace", "prefault": env, "seconds": round(dt, 3), **{k: round(v, 3) for k, v in st.items() if k.startswith("s_")}}), flush=True)
    b = time.perf_counter(); del kept, excluded, high; fr = time.perf_counter() - b
    s2 = {}
    a = time.perf_counter(); res = P.split_frames(other, rules, stats=s2); dt = time.perf_counter() - a
    print(json.dumps({"what": "split", "prefault": env, "seconds": round(dt, 3), "release_prev_s": round(fr, 3), **{k: (round(v, 3) if isinstance(v, float) else v) for k, v in s2.items() if k.endswith("_s") or k == "category_frames_fine"}}), flush=True)
    b = time.perf_counter(); del res, other; print(json.dumps({"what": "release", "prefault": env, "seconds": round(time.perf_counter() - b, 3)}), flush=True)
os.environ["DYD_PREFAULT"] = "1"
kept, excluded, high, other = P.replace_and_filter_frame(df, 2, 0.98)
del kept, excluded, high
pr = cProfile.Profile(); pr.enable()
res = P.split_frames(other, rules)
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
