#!/usr/bin/env python3
"""One forward at batch B and at batch 2B (the same slices twice): the first probe point of DAEngine.forward at which slice 0's
bits differ.   PREC=fp16 python tools/probes/batch_invariance.py [--size 512] [--batch 4]"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
from founddiff_amd import synth
ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=512)
ap.add_argument("--batch", type=int, default=4)
a = ap.parse_args()
prec = os.environ.get("PREC", "fp16")
dev = torch.device("cuda")
dif, _ = bench.build_model(dev, a.size, 50, prec)
e = dif._eng()
runs = []
for B in (a.batch, 2 * a.batch):
    _, ld = synth.ct_phantom(a.batch, a.size, seed=10)
    x = torch.from_numpy(ld).to(dev)
    x = torch.cat([x] * (B // a.batch), 0)
    x_in = (x * 2 - 1).contiguous()
    g = torch.Generator().manual_seed(4)
    nz = torch.randn(a.batch, 1, a.size, a.size, generator=g).to(dev)
    nz = torch.cat([nz] * (B // a.batch), 0)
    img = (x_in + 0.1 * nz).contiguous()
    tb = torch.full((B,), 500.0, device=dev)
    e.encode_condition(x_in)
    got, order = {}, []
    def rec(tag, t):
        got[tag] = t.detach()[:a.batch].clone()
        order.append(tag)
    e.probe = rec
    got["out"] = e.forward(img, x_in, tb)[:a.batch].clone()
    order.append("out")
    torch.cuda.synchronize()
    e.probe = None
    runs.append((got, order))
(g0, order), (g1, _) = runs
for tag in order:
    if tag not in g1 or g0[tag].shape != g1[tag].shape:
        print(f"{tag}: -"); continue
    d = (g0[tag].float() - g1[tag].float()).abs()
    print(f"{tag}: {'equal' if torch.equal(g0[tag], g1[tag]) else f'DIFFERENT max {float(d.max()):.3e} in {int((d > 0).sum())} of {d.numel()}'}")
