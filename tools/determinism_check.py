#!/usr/bin/env python3
"""Run the same forward twice and report which workspace buffers differ (race hunting)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from founddiff_amd import synth, _lib as L

size = int(sys.argv[1]) if len(sys.argv) > 1 else 512
dif, w = bench.build_model(torch.device("cuda"), size, 50, "bf16")
eng = dif._eng()
_, ld = synth.ct_phantom(2, size, seed=10)
x = torch.from_numpy(ld).cuda()
x_in = (x * 2 - 1).contiguous()
img = (x_in + 0.1 * torch.randn_like(x_in)).contiguous()
tb = torch.full((2,), 500.0, device="cuda")
eng.encode_condition(x_in)
eng.forward(img, x_in, tb)
torch.cuda.synchronize()
# per-launch determinism: replay each traced launch twice, compare every buffer after each
L.TRACE = []
eng.forward(img, x_in, tb)
trace, L.TRACE = L.TRACE, None
lib = L.lib()
torch.cuda.synchronize()
snap = {k: v.clone() for k, v in eng.buf.items()}
for rep in range(3):
    for i, (n, a) in enumerate(trace):
        getattr(lib, n)(*a)
    torch.cuda.synchronize()
    bad = [k[0] + str(k[1]) for k, v in eng.buf.items() if not torch.equal(v, snap[k])]
    print("rep", rep, "differing buffers:", bad[:12])
# find first launch whose output differs between two replays
for i, (n, a) in enumerate(trace):
    getattr(lib, n)(*a)
    torch.cuda.synchronize()
    s1 = {k: v.clone() for k, v in eng.buf.items()}
    diffs = set()
    for r in range(4):
        getattr(lib, n)(*a)
        torch.cuda.synchronize()
        for k, v in eng.buf.items():
            if not torch.equal(v, s1[k]):
                diffs.add(k[0] + str(k[1]))
    if diffs:
        d = ""
        if n == "fd_conv2d":
            p = a[0]._obj
            d = f"conv {p.KH}x{p.KW} {p.c0}+{p.c1}->{p.Cout} @{p.OH}x{p.OW} epi{p.epilogue} pro{p.prologue}"
        print("launch", i, n, d, "non-deterministic:", sorted(diffs)[:6])
