"""Which stage of one UNet forward is not repeatable when two engines run concurrently on two HIP streams?
Reference: engine 0 alone.  Then both engines at once (same inputs), NREP times; every probe-point tensor is compared
with the reference bit for bit; prints the first differing stage per run."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from founddiff_amd import synth
from founddiff_amd.engine import DAEngine
dev = torch.device("cuda")
prec = os.environ.get("PREC", "bf16")
B = int(os.environ.get("PB", "8"))
NREP = int(os.environ.get("NREP", "4"))
dif, w = bench.build_model(dev, precision=prec)
sd = {k[len("model.unet0."):]: v for k, v in w.items() if k.startswith("model.unet0.")}
engs = [DAEngine(sd, "", dev, prec) for _ in range(2)]
_, ld = synth.ct_phantom(B, 512, seed=10)
x_in = (torch.from_numpy(ld).to(dev) * 2 - 1).contiguous()
img = (x_in + 0.1 * torch.randn_like(x_in)).contiguous()
tb = torch.full((B,), 500.0, device=dev)
for e in engs:
    e.encode_condition(x_in)
    e.forward(img, x_in, tb)
torch.cuda.synchronize()
ref = {}
def rec(tag, t):
    ref[tag] = t.clone()
engs[0].probe = rec
engs[0].forward(img, x_in, tb)
torch.cuda.synchronize()
order = list(ref)
print(len(order), "probe points")
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
for rep in range(NREP):
    got = [{}, {}]
    for k in (0, 1):
        engs[k].probe = (lambda kk: (lambda tag, t: got[kk].__setitem__(tag, t.clone())))(k)
    # interleave the two launch sequences stage by stage is not possible from one host thread: launch them back to back;
    # the GPU overlaps them because the host runs ahead of both streams
    for k in (0, 1):
        streams[k].wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(streams[k]):
            engs[k].forward(img, x_in, tb)
    torch.cuda.synchronize()
    for k in (0, 1):
        bad = [t for t in order if not torch.equal(got[k][t], ref[t])]
        if bad:
            t0 = bad[0]
            d = (got[k][t0].float() - ref[t0].float()).abs()
            print(f"rep {rep} engine {k}: first differing stage {t0!r} (max diff {float(d.max()):.3e}, {int((d > 0).sum())} elements, "
                  f"{len(bad)} of {len(order)} stages differ); next: {bad[1:4]}")
            if d.dim() == 4:
                nz = (d > 0).nonzero()
                for bb in sorted(set(nz[:, 0].tolist())):
                    q = nz[nz[:, 0] == bb]
                    chans = sorted(set(q[:, 3].tolist()))
                    print(f"    slice {bb}: rows {int(q[:, 1].min())}..{int(q[:, 1].max())} cols {int(q[:, 2].min())}..{int(q[:, 2].max())} "
                          f"pixels {len(set(map(tuple, q[:, 1:3].tolist())))} channels {chans[0]}..{chans[-1]} ({len(chans)})")
                    g_, r_ = got[k][t0][bb].float(), ref[t0][bb].float()
                    yy, xx_ = int(q[0, 1]), int(q[0, 2])
                    print(f"      e.g. pixel ({yy},{xx_}): got {g_[yy, xx_, :4].tolist()} ref {r_[yy, xx_, :4].tolist()}")
        else:
            print(f"rep {rep} engine {k}: all stages equal")
