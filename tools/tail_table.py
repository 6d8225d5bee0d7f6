#!/usr/bin/env python3
"""Every launch of the precision tail's forward (the last DDIM step: levels 0-1 on the fp32s engine, the rest on the bf16
engine; ResidualDiffusion._tail_forward) at batch 8, replayed alone between HIP events, grouped by entry point.
usage: python tools/tail_table.py > profiles/rNN_tail_launches.md"""
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
from founddiff_amd import _lib as L, synth  # noqa: E402

dev = torch.device("cuda")
dif, _ = bench.build_model(dev)
eng = dif._eng()
B = 8
_, ld = synth.ct_phantom(B, 512, seed=10)
x_in = (torch.from_numpy(ld).to(dev) * 2 - 1).contiguous()
img = (x_in + 0.1 * torch.randn_like(x_in)).contiguous()
tb = torch.full((B,), 20.0, device=dev)
mo = torch.empty(B, 1, 512, 512, device=dev)
eng.encode_condition(x_in)
K, e32 = dif._tail_engine(eng)
assert K > 0
dif._tail_forward(e32, eng, img, x_in, tb, mo)
L.TRACE = []
dif._tail_forward(e32, eng, img, x_in, tb, mo)
trace, L.TRACE = L.TRACE, None
lib = L.lib()
DT = {0: "f32", 1: "bf16"}
rows, tot = [], 0.0
agg = defaultdict(lambda: [0, 0.0])
for n, args in trace:
    ms = bench._time_launches(lib, [(n, args)], reps=3)
    if n == "fd_conv2d":
        q = args[0]._obj
        desc = (f"{DT.get(q.dtype, q.dtype)}{' split' if q.f32_split else ''} kid {lib.fd_conv_kernel_id(args[0])}: {q.KH}x{q.KW} s{q.stride} "
                f"{q.c0 + q.c1} -> {q.Cout} @ {q.OH}x{q.OW}{' x4 dirs' if q.ndir == 4 else ''} epi {q.epilogue} pro {q.prologue}")
        key = f"fd_conv2d {DT.get(q.dtype, q.dtype)}"
    else:
        dt = args[0] if isinstance(args[0], int) else None
        desc = f"dtype word {dt}" if dt is not None else ""
        key = f"{n} {DT.get(dt & 0xff, '') if dt is not None else ''}"
    rows.append((n, desc, ms))
    agg[key][0] += 1
    agg[key][1] += ms
    tot += ms
print(f"# The precision tail's forward at batch 8, launch by launch (each replayed alone; sum {tot:.2f} ms)\n")
print("| entry point (storage type) | launches | ms |\n|---|---|---|")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"| `{k}` | {v[0]} | {v[1]:.3f} |")
print("\n| # | entry point | what | us |\n|---|---|---|---|")
for i, (n, d, ms) in enumerate(rows):
    print(f"| {i} | `{n}` | {d} | {ms * 1e3:.1f} |")
