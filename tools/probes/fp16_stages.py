#!/usr/bin/env python3
"""Stage by stage, one forward: the 'bf16' and the 'fp16' engine against the fp32 engine at every probe point of DAEngine.forward
(cumulative L2-relative error).   python tools/probes/fp16_stages.py [--size 256]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from founddiff_amd import synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=256)
ap.add_argument("--low-latency", action="store_true")
a = ap.parse_args()
dev = torch.device("cuda")
_, ld = synth.ct_phantom(1, a.size, seed=10)
x_in = torch.from_numpy(ld).to(dev) * 2 - 1
x_t = (x_in + 0.1 * torch.randn(x_in.shape, generator=torch.Generator().manual_seed(4)).to(dev)).contiguous()
tb = torch.full((1,), 500.0, device=dev)


def run(prec):
    dif, _ = bench.build_model(dev, a.size, 50, prec)
    e = dif._eng()
    e.encode_condition(x_in)
    got, order = {}, []

    def rec(tag, t):
        got[tag] = t.detach().float().clone()
        order.append(tag)
    e.probe = rec
    got["out"] = e.forward(x_t, x_in, tb).float().clone()
    order.append("out")
    torch.cuda.synchronize()
    return got, order


ref, order = run("fp32")
res = {p: run(p)[0] for p in ("bf16", "fp16")}
print("| stage | bf16 | fp16 | max |ref| |")
print("|---|---|---|---|")
for tag in order:
    r = ref[tag].double()
    row = []
    for p in ("bf16", "fp16"):
        g = res[p].get(tag)
        row.append("-" if g is None or g.shape != r.shape else f"{float((g.double() - r).norm() / r.norm().clamp(min=1e-30)):.2e}")
    print(f"| {tag} | {row[0]} | {row[1]} | {float(r.abs().max()):.3g} |")
