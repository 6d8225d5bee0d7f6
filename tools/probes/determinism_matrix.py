"""Run-to-run repeatability of sample() at the bench shape, by configuration (streams, precision tail)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from founddiff_amd import synth
dev = torch.device("cuda")
B = int(os.environ.get("PB", "16"))
_, ld = synth.ct_phantom(B, 512, seed=10)
x = torch.from_numpy(ld).to(dev)
noise = torch.stack([torch.randn(1, 512, 512, generator=torch.Generator().manual_seed(1000 + i)) for i in range(B)]).to(dev)
for prec, tail in (("bf16", 1), ("bf16", 0), ("fp32s", 0), ("fp16", 1), ("fp16", 0)):
    dif, _ = bench.build_model(dev, precision=prec)
    if prec in ("bf16", "fp16"):
        dif.final_fp32_steps = tail
    outs = [dif.sample([x], batch_size=B, noise=noise)[-1].clone() for _ in range(3)]
    torch.cuda.synchronize()
    msg = []
    for i in (1, 2):
        d = (outs[i] - outs[0]).abs()
        msg.append("equal" if torch.equal(outs[i], outs[0]) else f"max diff {float(d.max()):.2e} ({int((d > 0).sum())} px)")
    print(f"precision {prec} tail {tail} streams {dif.streams} batch {B}: {msg}", flush=True)
    del dif
    torch.cuda.empty_cache()
