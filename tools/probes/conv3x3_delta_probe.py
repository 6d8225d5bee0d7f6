#!/usr/bin/env python3
"""Delta-weight probe of the split-bf16 halo conv (development): w = 1 at one (channel, tap) -- which (channel, tap) of x does every
output channel actually show?  Found the vector-indexing pitfall in the halo conversion (channel k read as 4 (k / 4))."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, torch.nn.functional as F
from founddiff_amd.engine import ConvW
from test_gpu_e2e import bare_engine, nhwc, nchw
e = bare_engine("fp32s"); e.f32_split = 1
torch.manual_seed(1)
B, H, W, cin, cout = 1, 64, 64, 64, 64
x = torch.randn(B, cin, H, W).to(torch.bfloat16).float()
for (k, kh, kw) in ((0, 1, 1), (1, 1, 1), (9, 1, 1), (40, 1, 1), (3, 0, 0), (3, 2, 1), (20, 1, 2)):
    w = torch.zeros(cout, cin, 3, 3); w[:, k, kh, kw] = 1.0
    cw = ConvW(w, torch.zeros(cout), e.dev, e.tdt, split=True)
    out = torch.full((B, H, W, cout), float("nan"), device="cuda")
    e.conv(cw, nhwc(x, e.tdt), B, H, W, out); torch.cuda.synchronize()
    o = nchw(out)[0]                         # (cout, H, W)
    xp = F.pad(x[0], (1, 1, 1, 1))
    best = []
    for n in (0, 5, 37):
        cand = []
        for kk in range(cin):
            for dy in range(3):
                for dx in range(3):
                    ref = xp[kk, dy:dy + H, dx:dx + W]
                    cand.append((float((o[n, 8:56, 8:56] - ref[8:56, 8:56]).abs().max()), kk, dy, dx))
        cand.sort()
        best.append((n, cand[0]))
    print("delta (k, kh, kw) =", (k, kh, kw), "-> best match (err, k, kh, kw) per out channel:", best, flush=True)
