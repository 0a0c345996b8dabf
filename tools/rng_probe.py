#!/usr/bin/env python3
"""Which step of synth.generate_device goes wrong on big tables?  (rows past the middle of a 64M-box table came out centred on 0)"""
import json

import torch

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(7)
kw = {"generator": g, "device": dev}
out = {}
for B in (1 << 22, 1 << 24, 1 << 26):
    npts = torch.randint(3, 13, (B,), dtype=torch.int64, **kw)
    pt_off = torch.zeros(B + 1, dtype=torch.int64, device=dev)
    pt_off[1:] = torch.cumsum(npts, 0)
    P = int(pt_off[-1])
    box_of_pt = torch.repeat_interleave(torch.arange(B, device=dev), npts)
    centre = torch.rand((B, 2), dtype=torch.float64, **kw) * torch.tensor([1920.0, 1080.0], dtype=torch.float64, device=dev)
    probe = torch.randint(0, P, (4096,), device=dev)
    probe[-1] = P - 1
    want_box = torch.searchsorted(pt_off, probe, right=True) - 1
    res = {"P": P, "repeat_interleave_ok": bool(torch.equal(box_of_pt[probe], want_box))}
    got = centre[box_of_pt]
    res["advanced_index_ok"] = bool(torch.equal(got[probe], centre[want_box]))
    bad = (got[probe] != centre[want_box]).any(1)
    res["first_bad_probe_fraction"] = float(probe[bad].min() / P) if bool(bad.any()) else None
    got2 = torch.index_select(centre, 0, box_of_pt)
    res["index_select_ok"] = bool(torch.equal(got2[probe], centre[want_box]))
    got3 = torch.stack([centre[:, 0][box_of_pt], centre[:, 1][box_of_pt]], dim=1)
    res["per_column_ok"] = bool(torch.equal(got3[probe], centre[want_box]))
    r = torch.rand((P, 2), dtype=torch.float64, **kw)
    res["rand_tail_nonzero"] = bool((r[-1000:] != 0).all())
    s = got + r
    res["add_ok"] = bool(torch.equal(s[probe], got[probe] + r[probe]))
    out[str(B)] = res
    del npts, pt_off, box_of_pt, centre, got, got2, got3, r, s
print(json.dumps(out))
