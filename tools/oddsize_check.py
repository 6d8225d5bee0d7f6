#!/usr/bin/env python3
"""bf16 fast paths vs the fp32 (generic-kernel) engine at sizes that exercise the fallbacks:
one full-architecture UNet forward per size, L2-relative difference of the raw model output."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from founddiff_amd import synth
for (H, W) in ((144, 176), (256, 256), (208, 512), (512, 128), (512, 512), (384, 384)):
    outs = {}
    for prec in ("fp32", "bf16"):
        dif, w = bench.build_model(torch.device("cuda"), size=H, steps=50, precision=prec)
        eng = dif._eng()
        g = torch.Generator().manual_seed(3)
        x_in = (torch.rand(2, 1, H, W, generator=g) * 2 - 1).cuda()
        img = (x_in + 0.1 * torch.randn(2, 1, H, W, generator=g).cuda()).contiguous()
        tb = torch.full((2,), 500.0, device="cuda")
        eng.encode_condition(x_in)
        outs[prec] = eng.forward(img, x_in, tb).clone()
        del dif, eng
    a, b = outs["fp32"], outs["bf16"]
    print(f"{H}x{W}: L2-rel(bf16 vs fp32) = {float((a - b).norm() / a.norm()):.3e}  finite={bool(torch.isfinite(b).all())}")
