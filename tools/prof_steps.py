import cProfile, pstats, io, os, sys, tempfile, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import pandas as pd
from deal_yolo_daya_amd import _native, synth
from deal_yolo_daya_amd.core import processor as P
_native.lib()
df = synth.to_frame(synth.generate(20000, seed=synth.SEED))
with tempfile.TemporaryDirectory() as d:
    Q = lambda n: os.path.join(d, n)
    df.to_csv(Q("in.csv"), index=False, encoding="utf-8-sig")
    pd.DataFrame({"source": synth.reference_urls(20000)}).to_csv(Q("ref.csv"), index=False, encoding="utf-8-sig")
    def chain():
        P.deduplicate_csv_by_source(Q("in.csv"), Q("c1.csv"), verbose=False)
        P.remove_duplicates_between_csv(Q("c1.csv"), Q("ref.csv"), Q("c2.csv"), verbose=False)
        P.process_csv_replace_ptlist(Q("c2.csv"), Q("c3.csv"), Q("c3e.csv"))
        P.filter_by_box_count_and_iou(Q("c3.csv"), Q("c4h.csv"), Q("c4o.csv"), 2, 0.98)
    chain()
    t = time.perf_counter(); chain(); print("chain s", time.perf_counter() - t)
    pr = cProfile.Profile(); pr.enable(); chain(); pr.disable()
    s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(22); print(s.getvalue()[:5000])
