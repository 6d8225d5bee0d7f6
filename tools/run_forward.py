#!/usr/bin/env python3
"""Run `--n` eager UNet forwards (512x512, full arch, `--precision` bf16 | fp32s | ...) -- a small target for rocprofv3 --pmc."""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from founddiff_amd import synth
ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=2)
ap.add_argument("--batch", type=int, default=1)
ap.add_argument("--size", type=int, default=512)
ap.add_argument("--precision", default="bf16")
a = ap.parse_args()
dif, w = bench.build_model(torch.device("cuda"), a.size, 50, a.precision)
eng = dif._eng()
_, ld = synth.ct_phantom(a.batch, a.size, seed=10)
x = torch.from_numpy(ld).cuda()
x_in = (x * 2 - 1).contiguous()
img = (x_in + 0.1 * torch.randn_like(x_in)).contiguous()
tb = torch.full((a.batch,), 500.0, device="cuda")
eng.encode_condition(x_in)
for _ in range(a.n):
    eng.forward(img, x_in, tb)
torch.cuda.synchronize()
print("done")
