import os, sys, json, ctypes as C
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from deal_yolo_daya_amd import _native
L = _native.lib(); dev = torch.device("cuda:0"); sp = torch.cuda.current_stream().cuda_stream
g = torch.Generator(device=dev).manual_seed(1)
for big in (0, 600, 5000, 50000, -40, -100, -300):
    nb = torch.randint(1, 33, (100000,), generator=g, device=dev) if big >= 0 else torch.full((1_600_000 // -big,), -big, device=dev)
    if big > 0: nb[50000] = big
    ro = torch.zeros(nb.numel() + 1, dtype=torch.int32, device=dev); ro[1:] = torch.cumsum(nb, 0).to(torch.int32)
    N, B = nb.numel(), int(ro[-1].item())
    c = torch.rand((B, 2), generator=g, device=dev, dtype=torch.float64) * 1000
    box = torch.cat([c, c + torch.rand((B, 2), generator=g, device=dev, dtype=torch.float64) * 100 + 1], 1).contiguous()
    w = torch.full((N,), 1920.0, dtype=torch.float64, device=dev); h = torch.full((N,), 1080.0, dtype=torch.float64, device=dev)
    cid = (torch.arange(N, device=dev, dtype=torch.int32) % 20).contiguous()
    toff = torch.empty(N + 1, dtype=torch.int64, device=dev); flag = torch.empty(N, dtype=torch.uint8, device=dev)
    total = C.c_int64()
    _native.check(L.dyd_yolo_lines_dev(box.data_ptr(), ro.data_ptr(), None, w.data_ptr(), h.data_ptr(), cid.data_ptr(), N, B, toff.data_ptr(), flag.data_ptr(), None, 0, C.byref(total), sp), "m")
    text = torch.empty(total.value, dtype=torch.uint8, device=dev)
    ts = []
    for it in range(4):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); _native.check(L.dyd_yolo_lines_dev(box.data_ptr(), ro.data_ptr(), None, w.data_ptr(), h.data_ptr(), cid.data_ptr(), N, B, toff.data_ptr(), flag.data_ptr(), text.data_ptr(), total.value, C.byref(total), sp), "k7"); b.record(); b.synchronize()
        ts.append(a.elapsed_time(b))
    print(json.dumps({"rows": N, "lines": B, "one_row_of": big, "k7_ms": round(float(np.median(ts[1:])), 3)}), flush=True)
