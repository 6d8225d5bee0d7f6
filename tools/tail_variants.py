#!/usr/bin/env python3
"""Drift (vs the fp32 engine, 512x512, 50-step DDIM) and throughput (batch 16) of precision-tail variants: which of the
outer levels of the LAST step run on the split-bf16 fp32 engine: (outer up levels, of which also on the down side)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from founddiff_amd import synth
dev = torch.device("cuda")
size = int(sys.argv[1]) if len(sys.argv) > 1 else 512
_, ld = synth.ct_phantom(16, size, seed=10)
x = torch.from_numpy(ld).to(dev)
nz = torch.randn(16, 1, size, size, generator=torch.Generator().manual_seed(7)).to(dev)
ref_dif, _ = bench.build_model(dev, size, 50, "fp32")
ref = ref_dif.sample([x[:2]], batch_size=2, noise=nz[:2])[-1].double().cpu()
del ref_dif
torch.cuda.empty_cache()
for outer, down, steps in ((2, 2, 1), (2, 1, 1), (2, 0, 1), (1, 1, 1), (1, 0, 1), (0, 0, 0)):
    dif, _ = bench.build_model(dev, size, 50, "bf16")
    dif.final_fp32_steps, dif.final_outer_levels, dif.final_down_levels = steps, outer, down
    o = dif.sample([x[:2]], batch_size=2, noise=nz[:2])[-1].double().cpu()
    d = o - ref
    l2, psnr = float(d.norm() / ref.norm()), float(10 * torch.log10(1.0 / (d ** 2).mean()))
    dif.sample([x], batch_size=16, noise=nz)
    torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        dif.sample([x], batch_size=16, noise=nz)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    print(f"{size}x{size} tail steps {steps}, outer up levels {outer}, down levels {down}: L2 {l2:.3e}  PSNR {psnr:.1f} dB   {16 / min(ts):.3f} slices/s", flush=True)
    del dif
    torch.cuda.empty_cache()
