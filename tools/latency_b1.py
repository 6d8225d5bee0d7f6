#!/usr/bin/env python3
"""ms per 50-step 512x512 slice at batch 1 (bench.py's latency_b1 leg alone); env switches apply (development tool)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
from founddiff_amd import synth  # noqa: E402
dev = torch.device("cuda")
_, ld = synth.ct_phantom(1, 512, seed=10)
x = torch.from_numpy(ld).to(dev)
noise = torch.randn(1, 1, 512, 512, device=dev)
print(json.dumps(bench.latency_leg(dev, x, noise, reps=5)))
