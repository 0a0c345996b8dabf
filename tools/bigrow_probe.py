import os, sys, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from deal_yolo_daya_amd import _native
L = _native.lib(); dev = torch.device("cuda:0"); sp = torch.cuda.current_stream().cuda_stream
g = torch.Generator(device=dev).manual_seed(1)
for big in (0, 1000, 10000, 50000):
    nb = torch.randint(1, 33, (100000,), generator=g, device=dev)
    if big: nb[50000] = big
    box_off = torch.zeros(nb.numel() + 1, dtype=torch.int32, device=dev); box_off[1:] = torch.cumsum(nb, 0).to(torch.int32)
    N, B = nb.numel(), int(box_off[-1].item()); ppb = 8; P = B * ppb
    centre = torch.rand((B, 1, 2), generator=g, device=dev, dtype=torch.float64) * torch.tensor([1920.0, 1080.0], device=dev, dtype=torch.float64)
    xy = (centre + torch.rand((B, ppb, 2), generator=g, device=dev, dtype=torch.float64) * 100 - 50).reshape(P, 2).contiguous()
    pt_off = (torch.arange(B + 1, device=dev, dtype=torch.int64) * ppb).to(torch.int32)
    ob = torch.empty((B, 4), dtype=torch.float64, device=dev); oa = torch.empty((B, 4), dtype=torch.int32, device=dev); oh = torch.empty(N, dtype=torch.uint8, device=dev)
    ts = []
    for it in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); _native.check(L.dyd_bbox_iou_fused_dev(xy.data_ptr(), pt_off.data_ptr(), box_off.data_ptr(), N, B, P, 2, 0.98, ob.data_ptr(), oa.data_ptr(), oh.data_ptr(), sp), "k12"); b.record(); b.synchronize()
        ts.append(a.elapsed_time(b))
    print(json.dumps({"rows": N, "boxes": B, "one_row_of": big, "ms": round(float(np.median(ts[1:])), 3)}), flush=True)
