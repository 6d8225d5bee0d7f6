#!/usr/bin/env python3
"""Experiment: two half-batches on two streams (two engines, own workspaces) against one batch on one stream --
do kernels bound by different resources (VALU-issue scan, HBM row-GEMMs, MFMA convs) overlap when they come
from independent launches?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from founddiff_amd import synth
dev = torch.device("cuda")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
nstream = int(sys.argv[2]) if len(sys.argv) > 2 else 2
difs = [bench.build_model(dev, 512, 50, "bf16")[0] for _ in range(nstream)]
engs = [d._eng() for d in difs]
streams = [torch.cuda.Stream() for _ in range(nstream)]
_, ld = synth.ct_phantom(B, 512, seed=10)
x_in = (torch.from_numpy(ld).to(dev) * 2 - 1).contiguous()
img = (x_in + 0.1 * torch.randn_like(x_in)).contiguous()
h = B // nstream
parts = [(img[i * h:(i + 1) * h].contiguous(), x_in[i * h:(i + 1) * h].contiguous(), torch.full((h,), 500.0, device=dev)) for i in range(nstream)]
one = (img, x_in, torch.full((B,), 500.0, device=dev))
engs[0].encode_condition(x_in)
engs[0].forward(*one)
for e, p in zip(engs, parts):
    e.encode_condition(p[1]); e.forward(*p)
torch.cuda.synchronize()
# graphs: one forward per engine on its stream
def cap(e, p, s):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        e.forward(*p)
    return g
engs[0].encode_condition(x_in)
g_one = cap(engs[0], one, streams[0])
for e, p in zip(engs, parts):
    e.encode_condition(p[1])
g_parts = [cap(e, p, s) for e, p, s in zip(engs, parts, streams)]
def run_one(n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    with torch.cuda.stream(streams[0]):
        for _ in range(n): g_one.replay()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
def run_parts(n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        for g, s in zip(g_parts, streams):
            with torch.cuda.stream(s): g.replay()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
for _ in range(2):
    a = run_one(20); b = run_parts(20)
    print(f"B={B}: one stream {a*1e3:.3f} ms/forward ({a*1e3/B:.3f} per slice); {nstream} streams x B={h}: {b*1e3:.3f} ms ({b*1e3/B:.3f} per slice)  ratio {a/b:.3f}")
