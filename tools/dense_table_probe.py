#!/usr/bin/env python3
"""What do the rows of a device-drawn dense table look like?  (first rows of tables of several sizes, pairs per IoU band)"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))


def main():
    import torch
    from deal_yolo_daya_amd import _native, synth
    from oracle import lib as olib
    L = _native.lib()
    dev = torch.device("cuda:0")
    sp = torch.cuda.current_stream().cuda_stream
    for n in (1000, 250000):
        d = synth.generate_device(n, 7, dev, boxes_per_row=256)
        xy, pt_off, box_off = d["xy"], d["pt_off"], d["box_off"]
        B, P = int(pt_off.shape[0]) - 1, int(xy.shape[0])
        out_box = torch.empty((B, 4), dtype=torch.float64, device=dev); out_arg = torch.empty((B, 4), dtype=torch.int32, device=dev)
        _native.check(L.dyd_bbox_minmax_dev(xy.data_ptr(), pt_off.data_ptr(), B, P, out_box.data_ptr(), out_arg.data_ptr(), sp), "k1")
        torch.cuda.synchronize()
        for r0 in (0, n // 2, n - 16):
            box = out_box[r0 * 256:(r0 + 16) * 256].cpu().numpy()
            off = np.arange(17, dtype=np.int32) * 256
            pt = pt_off[r0 * 256:(r0 + 16) * 256 + 1].cpu().numpy()
            xs = xy[int(pt[0]):int(pt[-1])].cpu().numpy()
            obox, _ = olib.bbox_minmax(xs, (pt - pt[0]).astype(np.int32))
            mx = olib.iou_any_ge(box, off, 2, 0.98, want_max=True)[1]
            w = box[:, 2] - box[:, 0]
            print(json.dumps({"rows": n, "first_row": r0, "k1_matches_oracle": bool(np.array_equal(box, obox)), "mean_w": round(float(w.mean()), 2),
                              "max_w": round(float(w.max()), 2), "distinct_x1_of_4096": int(len(np.unique(box[:, 0]))),
                              "row_max_iou": [round(float(v), 3) for v in mx[:8]], "sample": box[:2].tolist()}), flush=True)
        del d, xy, pt_off, box_off, out_box, out_arg


if __name__ == "__main__":
    main()
