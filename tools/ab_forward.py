#!/usr/bin/env python3
"""A/B instrument: time of the UNet forward (one captured graph, batch 8, one stream) and of the production sample()
(batch 16, two streams, 50-step DDIM) with the library FOUNDDIFF_LIB points at.  Run alternately for two builds inside
ONE gpurun call (tools/probes/ab.sh): box-to-box and warm-up drift are larger than most single optimisations.
usage: FOUNDDIFF_LIB=... python tools/ab_forward.py [tag] [--sample]"""
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
from founddiff_amd import synth  # noqa: E402

tag = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("--") else "lib"
# AB_SKIP=fd_gn_finalize,fd_chan_attn_weff: TIMING experiment only -- what would eliminating those launches buy?  The skip is
# switched on AFTER one regular forward / sample() (apply_skip below): the buffers those launches write then hold finite values of
# the right scale (skipped from the start they hold garbage, NaNs spread through the forward, and a NaN forward draws less power
# and clocks higher: that run measured -6 % for 2 % of launch time -- an artefact of the data, not of the launches).
def apply_skip():
    if os.environ.get("AB_SKIP"):
        from founddiff_amd import _lib as _L
        _skip, _call = set(os.environ["AB_SKIP"].split(",")), _L.call
        _L.call = lambda name, *a: None if name in _skip else _call(name, *a)
dev = torch.device("cuda")
dif, _ = bench.build_model(dev)
eng = dif._eng()
B = 8
_, ld = synth.ct_phantom(16, 512, seed=10)
x16 = torch.from_numpy(ld).to(dev)
x_in = (x16[:B] * 2 - 1).contiguous()
img = (x_in + 0.1 * torch.randn_like(x_in)).contiguous()
tb = torch.full((B,), 500.0, device=dev)
eng.encode_condition(x_in)
eng.forward(img, x_in, tb)
torch.cuda.synchronize()
if "--sample" in sys.argv and os.environ.get("AB_SKIP"):
    _n = torch.randn(16, 1, 512, 512, device=dev)
    dif.sample([x16], batch_size=16, noise=_n)          # regular run: every engine's buffers hold real values
    torch.cuda.synchronize()
    dif._drop_graphs()                                  # ... and the loops are captured again, without the skipped launches
apply_skip()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    eng.forward(img, x_in, tb)
for _ in range(5):
    g.replay()
torch.cuda.synchronize()
ts = []
for _ in range(12):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        g.replay()
    e1.record()
    e1.synchronize()
    ts.append(e0.elapsed_time(e1) / 10)
res = {"tag": tag, "forward_b8_ms_median": round(statistics.median(ts), 4), "forward_b8_ms_min": round(min(ts), 4),
       "ms_per_slice": round(statistics.median(ts) / B, 4)}
if "--sample" in sys.argv:
    noise = torch.randn(16, 1, 512, 512, device=dev)
    dif.sample([x16], batch_size=16, noise=noise)
    torch.cuda.synchronize()
    tt = []
    for _ in range(3):
        t0 = time.perf_counter()
        dif.sample([x16], batch_size=16, noise=noise)
        torch.cuda.synchronize()
        tt.append(time.perf_counter() - t0)
    res["sample_b16_slices_per_s"] = round(16 / min(tt), 3)
print(json.dumps(res), flush=True)
