#!/usr/bin/env python3
"""Time DAEngine.encode_condition (DA-CLIP RN50 + heads, once per sample()) and one sample() end to end."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from founddiff_amd import synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dif, w = bench.build_model(torch.device("cuda"))
eng = dif._eng()
_, ld = synth.ct_phantom(B, 512, seed=10)
x = torch.from_numpy(ld).cuda()
x_in = (x * 2 - 1).contiguous()
for _ in range(2):
    eng.encode_condition(x_in)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    eng.encode_condition(x_in)
torch.cuda.synchronize()
print(f"encode_condition B={B}: {(time.perf_counter() - t0) / 5 * 1e3:.2f} ms")
noise = torch.randn(B, 1, 512, 512, device="cuda")
dif.sample([x], batch_size=B, noise=noise)
torch.cuda.synchronize()
t0 = time.perf_counter()
dif.sample([x], batch_size=B, noise=noise)
torch.cuda.synchronize()
print(f"sample() B={B}: {(time.perf_counter() - t0) * 1e3:.1f} ms")
