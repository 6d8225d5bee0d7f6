#!/usr/bin/env python3
"""BASELINE configs[3] sanity: full 1000-step ancestral p_sample_loop at 512x512 (bf16, synthetic weights)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from founddiff_amd import synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
dif, w = bench.build_model(torch.device("cuda"), steps=1000)
assert not dif.is_ddim_sampling
_, ld = synth.ct_phantom(B, 512, seed=10)
x = torch.from_numpy(ld).cuda()
torch.manual_seed(0)
t0 = time.perf_counter()
out = dif.sample([x], batch_size=B)[-1]
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"ancestral 1000 steps B={B}: {dt:.2f} s ({B / dt:.3f} slices/s), finite={bool(torch.isfinite(out).all())}, "
      f"range=[{float(out.min()):.3f}, {float(out.max()):.3f}]")
