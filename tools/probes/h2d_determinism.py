"""Is sample() bitwise repeatable when its input batch arrives by a pinned-host -> device copy on the launch stream?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from founddiff_amd import synth
dev = torch.device("cuda")
dif, _ = bench.build_model(dev)
B = 16
_, ld = synth.ct_phantom(B, 512, seed=10)
x = torch.from_numpy(ld).to(dev)
noise = torch.stack([torch.randn(1, 512, 512, generator=torch.Generator().manual_seed(1000 + i)) for i in range(B)]).to(dev)
outs = []
for i in range(3):
    outs.append(dif.sample([x], batch_size=B, noise=noise)[-1].clone())
xh = x.cpu().pin_memory()
xd = torch.empty_like(x)
for i in range(3):
    xd.copy_(xh, non_blocking=True)
    outs.append(dif.sample([xd], batch_size=B, noise=noise)[-1].clone())
torch.cuda.synchronize()
print("x == xd:", torch.equal(x, xd))
for i in range(1, 6):
    d = (outs[i] - outs[0]).abs()
    print(i, "equal" if torch.equal(outs[i], outs[0]) else f"max diff {float(d.max()):.3e} in {int((d > 0).sum())} px, slices {sorted(set((d.flatten(1).max(1).values > 0).nonzero().flatten().tolist()))}")
