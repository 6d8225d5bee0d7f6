#!/usr/bin/env python3
"""Per-launch time of every fd_conv2d launch of one batch-8 forward that the persistent pointwise GEMM takes (kernel id 7)
or would take (id 5 with FD_NO_PWGEMM=1): run once with and once without the switch and compare."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
from founddiff_amd import _lib as L, synth
dev = torch.device("cuda")
dif, _ = bench.build_model(dev)
eng = dif._eng()
B = 8
_, ld = synth.ct_phantom(B, 512, seed=10)
x_in = (torch.from_numpy(ld).to(dev) * 2 - 1).contiguous()
img = (x_in + 0.1 * torch.randn_like(x_in)).contiguous()
tb = torch.full((B,), 500.0, device=dev)
eng.encode_condition(x_in)
eng.forward(img, x_in, tb)
L.TRACE = []
eng.forward(img, x_in, tb)
trace, L.TRACE = L.TRACE, None
lib = L.lib()
tot = 0.0
for n, a in trace:
    if n != "fd_conv2d":
        continue
    q = a[0]._obj
    kid = lib.fd_conv_kernel_id(a[0])
    if kid not in (5, 7):
        continue
    ms = bench._time_launches(lib, [(n, a)], reps=5)
    tot += ms
    print(f"kid={kid} K={q.c0 + q.c1:5d} N={q.Cout:5d} HW={q.OH}x{q.OW} epi={q.epilogue} wbs={int(q.w_batch_stride > 0)}: {ms * 1e3:7.1f} us")
print(f"sum {tot * 1e3:.1f} us")
