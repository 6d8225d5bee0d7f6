"""precision='fp16' at the bench shape: run-to-run repeatability per stream count, and one stream of batch B against the same slices
run as sub-batches (is a slice's result independent of the batch it is computed in?)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from founddiff_amd import synth
dev = torch.device("cuda")
B = int(os.environ.get("PB", "8"))
prec = os.environ.get("PREC", "fp16")
_, ld = synth.ct_phantom(B, 512, seed=10)
x = torch.from_numpy(ld).to(dev)
noise = torch.stack([torch.randn(1, 512, 512, generator=torch.Generator().manual_seed(1000 + i)) for i in range(B)]).to(dev)
for tail in (0, 1):
    dif, _ = bench.build_model(dev, precision=prec)
    dif.final_fp32_steps = tail
    res = {}
    for streams in (2, 1):
        dif.streams = streams
        outs = [dif.sample([x], batch_size=B, noise=noise)[-1].clone() for _ in range(3)]
        torch.cuda.synchronize()
        msg = ["equal" if torch.equal(o, outs[0]) else f"max diff {float((o - outs[0]).abs().max()):.2e} ({int(((o - outs[0]).abs() > 0).sum())} px)" for o in outs[1:]]
        print(f"{prec} tail {tail} streams {streams} batch {B}: run 2, 3 vs run 1: {msg}", flush=True)
        res[streams] = outs[0]
    d = (res[2] - res[1]).abs()
    print(f"   two streams vs one: {'equal' if torch.equal(res[2], res[1]) else f'max diff {float(d.max()):.2e} ({int((d > 0).sum())} px)'}", flush=True)
    dif.streams = 1
    h = B // 2
    halves = torch.cat([dif.sample([x[i:i + h]], batch_size=h, noise=noise[i:i + h].contiguous())[-1] for i in (0, h)], 0)
    d = (halves - res[1]).abs()
    print(f"   one stream, two halves after each other vs the whole batch: {'equal' if torch.equal(halves, res[1]) else f'max diff {float(d.max()):.2e} ({int((d > 0).sum())} px)'}", flush=True)
    del dif
    torch.cuda.empty_cache()
