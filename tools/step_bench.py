#!/usr/bin/env python3
"""End-to-end timing of the drop-in STEP FUNCTIONS (CSV in -> CSV out and DataFrame in -> DataFrame
out) against the CPU port of the reference, on the same synthetic table.  Region (2) and (3) of
SURVEY §8d: includes read_csv / flatten / H2D / kernels / D2H / emit / to_csv, so it is host-bound.

    python tools/step_bench.py --rows 20000
"""
import argparse
import json
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=20000)
    ap.add_argument("--skip-cpu", action="store_true")
    args = ap.parse_args()

    import pandas as pd
    from deal_yolo_daya_amd import _native, synth
    from deal_yolo_daya_amd.core import processor as P
    from oracle import steps as osteps

    _native.lib()
    t = synth.generate(args.rows, seed=synth.SEED)
    df = synth.to_frame(t)
    out = {"rows": args.rows, "host_cores": os.cpu_count(), "device": _native.device_name()}

    def clock(fn):
        t0 = time.perf_counter()
        r = fn()
        return time.perf_counter() - t0, r

    with tempfile.TemporaryDirectory() as d:
        Q = lambda n: os.path.join(d, n)  # noqa: E731
        df.to_csv(Q("in.csv"), index=False, encoding="utf-8-sig")
        out["input_csv_MB"] = round(os.path.getsize(Q("in.csv")) / 1e6, 1)

        # ---- product, path level -------------------------------------------------------------------
        for mode in ("1", "0"):
            os.environ["DYD_NATIVE_JSON"] = mode
            tag = "native_json" if mode == "1" else "cpython_json"
            P.process_csv_replace_ptlist(Q("in.csv"), Q("w.csv"), Q("we.csv"))          # warm-up (page cache, HIP init)
            s1, _ = clock(lambda: P.process_csv_replace_ptlist(Q("in.csv"), Q(f"p_{tag}.csv"), Q(f"e_{tag}.csv")))
            s2, _ = clock(lambda: P.filter_by_box_count_and_iou(Q(f"p_{tag}.csv"), Q(f"h_{tag}.csv"), Q(f"o_{tag}.csv"), 2, 0.98))
            out[f"product_path_{tag}"] = {"replace_s": round(s1, 3), "iou_s": round(s2, 3),
                                          "rows_per_s": round(args.rows / (s1 + s2))}
        os.environ["DYD_NATIVE_JSON"] = "1"
        # ---- product, frame level + breakdown ---------------------------------------------------------
        s_read, df_in = clock(lambda: pd.read_csv(Q("in.csv"), encoding="utf-8-sig"))
        stats = {}
        s1, (kept, _) = clock(lambda: P.replace_ptlist_frame(df_in, None, stats))
        s2, (hi, lo) = clock(lambda: P.iou_filter_frame(kept, 2, 0.98))
        s_write, _ = clock(lambda: kept[["source", P.ANNOTATION_COL, P.BBOX_COL, "width", "height"]].to_csv(
            Q("x.csv"), index=False, encoding="utf-8-sig"))
        out["product_frame_native_json"] = {"replace_s": round(s1, 3), "iou_s": round(s2, 3),
                                            "rows_per_s": round(args.rows / (s1 + s2)), "python_cells": stats["python_cells"],
                                            "pandas_read_csv_s": round(s_read, 3), "pandas_to_csv_s": round(s_write, 3)}
        # pieces of the replace step
        cells = df_in[P.ANNOTATION_COL].tolist()
        from deal_yolo_daya_amd import native_json as nj
        a, scan = clock(lambda: nj.scan_polygons(cells))
        b, (_, arg4) = clock(lambda: _native.bbox_minmax(scan.xy, scan.pt_off))
        c, texts = clock(lambda: scan.emit(arg4))
        out["replace_breakdown_s"] = {"native_scan(incl. join+encode)": round(a, 3), "K1 host-pointer call (H2D+kernel+D2H)": round(b, 3),
                                      "K1 kernel_ms": round(_native.last_kernel_ms(), 3), "native_emit(incl. str objects)": round(c, 3)}
        # ---- the four CSV->CSV steps in pipeline order (dedup, ref filter, replace, IoU) --------------------
        ref_df = pd.DataFrame({"source": synth.reference_urls(args.rows)})
        ref_df.to_csv(Q("ref.csv"), index=False, encoding="utf-8-sig")

        def chain(mod, tag, names):
            t0 = time.perf_counter()
            getattr(mod, names[0])(Q("in.csv"), Q(f"c1_{tag}.csv"), **({"verbose": False} if mod is P else {}))
            t1 = time.perf_counter()
            getattr(mod, names[1])(Q(f"c1_{tag}.csv"), Q("ref.csv"), Q(f"c2_{tag}.csv"), **({"verbose": False} if mod is P else {}))
            t2 = time.perf_counter()
            getattr(mod, names[2])(Q(f"c2_{tag}.csv"), Q(f"c3_{tag}.csv"), Q(f"c3e_{tag}.csv"))
            t3 = time.perf_counter()
            getattr(mod, names[3])(Q(f"c3_{tag}.csv"), Q(f"c4h_{tag}.csv"), Q(f"c4o_{tag}.csv"), 2, 0.98)
            t4 = time.perf_counter()
            return {"dedup_s": round(t1 - t0, 3), "ref_filter_s": round(t2 - t1, 3), "replace_s": round(t3 - t2, 3),
                    "iou_s": round(t4 - t3, 3), "total_s": round(t4 - t0, 3), "input_rows_per_s": round(args.rows / (t4 - t0))}

        prod_names = ("deduplicate_csv_by_source", "remove_duplicates_between_csv", "process_csv_replace_ptlist",
                      "filter_by_box_count_and_iou")
        chain(P, "warm", prod_names)
        out["product_chain_csv_to_csv"] = chain(P, "prod", prod_names)
        out["product_chain_io_paths"] = dict(P.LAST_IO_PATH)
        # ---- merge step (f3): the table cut into 4 files and merged back ------------------------------------
        os.makedirs(Q("parts"))
        quarter = (args.rows + 3) // 4
        for k in range(4):
            df.iloc[k * quarter:(k + 1) * quarter].to_csv(Q(f"parts/part{k}.csv"), index=False, encoding="utf-8-sig")
        import contextlib, io as _io
        with contextlib.redirect_stdout(_io.StringIO()):
            P.merge_all_csv_in_folder(Q("parts"), Q("merged_warm.csv"))
            s_m, n_m = clock(lambda: P.merge_all_csv_in_folder(Q("parts"), Q("merged_prod.csv")))
        out["product_merge"] = {"s": round(s_m, 3), "rows_per_s": round(n_m / s_m), "io_paths": dict(P.LAST_IO_PATH["merge"])}
        # ---- label_replace step (between IoU filter and split): CSV -> CSV over the IoU step's "other" table ------------
        mapping = pd.DataFrame({"old": [f"c{i}" for i in range(0, 20, 2)], "new": [f"g{i % 4}" for i in range(10)]})
        real_read_excel, real_to_excel = pd.read_excel, pd.DataFrame.to_excel
        pd.read_excel = lambda *a, **k: mapping.copy()                       # openpyxl is not installed: the Excel layer is stubbed
        pd.DataFrame.to_excel = lambda self, *a, **k: None
        try:
            P.replace_labels_by_mapping(Q("c4o_prod.csv"), "map.xlsx", Q("lr_warm.csv"))
            s_lr, res_lr = clock(lambda: P.replace_labels_by_mapping(Q("c4o_prod.csv"), "map.xlsx", Q("lr_prod.csv"), diff_excel_path=Q("d.xlsx"),
                                                                    unmatched_excel_path=Q("u.xlsx")))
            out["product_label_replace"] = {"s": round(s_lr, 3), "rows": res_lr["summary"]["total_rows"], "rows_per_s": round(res_lr["summary"]["total_rows"] / s_lr),
                                            "replaced_labels": res_lr["summary"]["replaced_labels"], "io_path": P.LAST_IO_PATH["label_replace"]}
            if not args.skip_cpu:
                s_lr, ores = clock(lambda: osteps.label_replace_csv(Q("c4o_prod.csv"), mapping, Q("lr_cpu.csv"), diff_excel_path="d", unmatched_excel_path="u"))
                out["cpu_port_label_replace"] = {"s": round(s_lr, 3), "rows_per_s": round(ores["summary"]["total_rows"] / s_lr)}
                out["label_replace_identical"] = open(Q("lr_prod.csv"), "rb").read() == open(Q("lr_cpu.csv"), "rb").read() \
                    and ores["summary"] == res_lr["summary"] and ores["sample_diff"] == res_lr["sample_diff"]
        finally:
            pd.read_excel, pd.DataFrame.to_excel = real_read_excel, real_to_excel
        # ---- split + YOLO label texts (a5 in memory, f4) ----------------------------------------------------
        rules = pd.DataFrame({"catA": [f"c{i}" for i in range(10)], "catB": [f"c{i}" for i in range(10, 18)] + [None, None]})
        lmap = P.rules_to_label_map(rules)
        s_sp, sp = clock(lambda: P.split_frames(kept, lmap))
        sheet = pd.concat([fr for cat in sp["categories"].values() for fr in cat], ignore_index=True)
        classes = sorted(set(sheet["分类标签"]))
        cid = {c: i for i, c in enumerate(classes)}
        ycells, ylabels = sheet[P.BBOX_COL].tolist(), sheet["分类标签"].tolist()
        ycids, yw, yh = [cid[v] for v in ylabels], sheet["width"].tolist(), sheet["height"].tolist()
        P.yolo_label_texts(ycells[:100], ylabels[:100], ycids[:100], yw[:100], yh[:100])
        ystats = {}
        s_y, (ytexts, _) = clock(lambda: P.yolo_label_texts(ycells, ylabels, ycids, yw, yh, None, ystats))
        out["product_split_frames"] = {"s": round(s_sp, 3), "expanded_rows": len(sheet), "input_rows_per_s": round(len(kept) / s_sp)}
        out["product_yolo_label_texts"] = {"s": round(s_y, 3), "rows": len(sheet), "rows_per_s": round(len(sheet) / s_y), **ystats}
        # ---- CPU port of the reference, path level ------------------------------------------------------
        if not args.skip_cpu:
            s1, _ = clock(lambda: osteps.replace_csv(Q("in.csv"), Q("rp.csv"), Q("re.csv")))
            s2, _ = clock(lambda: osteps.iou_filter_csv(Q("rp.csv"), Q("rh.csv"), Q("ro.csv"), 2, 0.98))
            out["cpu_port_path"] = {"replace_s": round(s1, 3), "iou_s": round(s2, 3), "rows_per_s": round(args.rows / (s1 + s2))}
            same = all(open(Q(a), "rb").read() == open(Q(b), "rb").read()
                       for a, b in (("rp.csv", "p_native_json.csv"), ("rh.csv", "h_native_json.csv"), ("ro.csv", "o_native_json.csv"),
                                    ("rp.csv", "p_cpython_json.csv")))
            out["outputs_byte_identical_to_cpu_port"] = same
            out["cpu_port_chain_csv_to_csv"] = chain(osteps, "cpu", ("dedup_csv", "ref_filter_csv", "replace_csv", "iou_filter_csv"))
            out["chain_outputs_byte_identical"] = all(
                open(Q(f"{n}_prod.csv"), "rb").read() == open(Q(f"{n}_cpu.csv"), "rb").read() for n in ("c1", "c2", "c3", "c3e", "c4h", "c4o"))
            with contextlib.redirect_stdout(_io.StringIO()):
                s_m, n_m = clock(lambda: osteps.merge_folder(Q("parts"), Q("merged_cpu.csv")))
            out["cpu_port_merge"] = {"s": round(s_m, 3), "rows_per_s": round(n_m / s_m)}
            out["merge_output_byte_identical"] = open(Q("merged_prod.csv"), "rb").read() == open(Q("merged_cpu.csv"), "rb").read()
            sub = kept.iloc[: max(1, len(kept) // 10)]                       # the port copies a Series per record: a tenth is enough
            s_sp, osp = clock(lambda: osteps.split_frames(sub, lmap))
            out["cpu_port_split_frames"] = {"s": round(s_sp, 3), "input_rows": len(sub), "input_rows_per_s": round(len(sub) / s_sp)}
            psp = P.split_frames(sub, lmap)
            out["split_frames_identical"] = all(a.equals(b) for c in osp["categories"] for a, b in zip(osp["categories"][c], psp["categories"][c])) \
                and osp["unclassified"].equals(psp["unclassified"]) and osp["split_counts"].equals(psp["split_counts"])
            s_y, want = clock(lambda: [osteps.yolo_row_text(c, l, k, w, h)[0] for c, l, k, w, h in zip(ycells, ylabels, ycids, yw, yh)])
            out["cpu_port_yolo_label_texts"] = {"s": round(s_y, 3), "rows_per_s": round(len(sheet) / s_y)}
            out["yolo_texts_identical"] = want == ytexts
    print(json.dumps(out, ensure_ascii=False))


if __name__ == "__main__":
    main()
