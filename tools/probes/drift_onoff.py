#!/usr/bin/env python3
"""Drift of the bf16 production mode against the fp32 parity engine (512x512, 50-step DDIM, 2 slices) with the round-4 dataflow
(z / v recomputed, apply fused with the down-sampling convolutions) and with the round-3 dataflow (all three off)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
from founddiff_amd import synth
dev = torch.device("cuda")
size = int(sys.argv[1]) if len(sys.argv) > 1 else 512
_, ld = synth.ct_phantom(2, size, seed=10)
x = torch.from_numpy(ld).to(dev)
nz = torch.randn(2, 1, size, size, generator=torch.Generator().manual_seed(7)).to(dev)
def l2(a, b): return float((a.double() - b.double()).norm() / b.double().norm())
def psnr(a, b): return float(10 * torch.log10(1.0 / ((a.double() - b.double()) ** 2).mean()))
dif, _ = bench.build_model(dev, size, 50, "fp32")
ref = dif.sample([x], batch_size=2, noise=nz)[-1].float().cpu()
del dif; torch.cuda.empty_cache()
outs = {}
for tag, flags in (("round-4 dataflow", (1, True, True)), ("round-3 dataflow", (0, False, False)), ("z only", (1, False, False))):
    dif, _ = bench.build_model(dev, size, 50, "bf16")
    eng = dif._eng()
    eng.z_recompute, eng.v_recompute, eng.down_fuse = flags
    outs[tag] = dif.sample([x], batch_size=2, noise=nz)[-1].float().cpu()
    print(f"{size}x{size} {tag:18s}: L2 {l2(outs[tag], ref):.3e}  PSNR {psnr(outs[tag], ref):.2f} dB", flush=True)
    del dif; torch.cuda.empty_cache()
print(f"round-4 vs round-3 dataflow: L2 {l2(outs['round-4 dataflow'], outs['round-3 dataflow']):.3e}")
