"""Which co-running kernel makes a packed-fp32 build of the level-0 scan non-repeatable?  (FOUNDDIFF_LIB = a build WITH
v_pk_*_f32: python /tmp/build_variant.py pk --packed.)  Victim: the level-0 fd_selective_scan_xproj launch of one engine, on
stream A.  Aggressor: one launch kind of a second engine's forward, looped on stream B meanwhile.  The victim's y is compared
bit for bit with its solo result; prints mismatching repetitions per aggressor kind."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from founddiff_amd import _lib as L, synth
from founddiff_amd.engine import DAEngine
dev = torch.device("cuda")
B = 8
NREP = int(os.environ.get("NREP", "12"))
dif, w = bench.build_model(dev)
sd = {k[len("model.unet0."):]: v for k, v in w.items() if k.startswith("model.unet0.")}
engs = [DAEngine(sd, "", dev, "bf16") for _ in range(2)]
_, ld = synth.ct_phantom(B, 512, seed=10)
x_in = (torch.from_numpy(ld).to(dev) * 2 - 1).contiguous()
img = (x_in + 0.1 * torch.randn_like(x_in)).contiguous()
tb = torch.full((B,), 500.0, device=dev)
traces = []
for e in engs:
    e.encode_condition(x_in)
    e.forward(img, x_in, tb)
    L.TRACE = []
    e.forward(img, x_in, tb)
    traces.append(L.TRACE)
    L.TRACE = None
torch.cuda.synchronize()
lib = L.lib()
print(lib.fd_dev_options().decode())
victim = [(n, a) for n, a in traces[0] if n == "fd_selective_scan_xproj"][-1]        # u3m: the last level-0 scan: its inputs are what the buffers hold
ybuf = engs[0].buf[("scan_y", (B, 512, 512, 128), torch.bfloat16)]
sA, sB = torch.cuda.Stream(), torch.cuda.Stream()


def run_victim():
    with torch.cuda.stream(sA):
        a = list(victim[1])
        a[-1] = sA.cuda_stream
        getattr(lib, victim[0])(*a)


run_victim()
torch.cuda.synchronize()
ref = ybuf.clone()
kinds = {}
for n, a in traces[1]:
    key = n
    if n == "fd_conv2d":
        q = a[0]._obj
        key = f"fd_conv2d kid {lib.fd_conv_kernel_id(a[0])} {q.KH}x{q.KW} {q.c0 + q.c1}->{q.Cout} @{q.OH}"
    elif n.startswith("fd_selective_scan"):
        o = 1 if n.endswith("xproj") else 0
        key = f"{n} D={a[12 + o]} N={a[13 + o]} @{a[10 + o]}"
    elif n in ("fd_pw_dw3x3", "fd_pw_dw3x3_gram", "fd_pw_dw3x3_proj"):
        key = f"{n} @{a[-3]}"
    kinds.setdefault(key, (n, a))
print(len(kinds), "aggressor kinds")
for key, (n, a) in kinds.items():
    a = list(a)
    a[-1] = sB.cuda_stream
    bad = 0
    worst = 0.0
    for rep in range(NREP):
        torch.cuda.synchronize()
        with torch.cuda.stream(sB):
            for _ in range(6):
                getattr(lib, n)(*a)
        run_victim()
        with torch.cuda.stream(sB):
            for _ in range(6):
                getattr(lib, n)(*a)
        torch.cuda.synchronize()
        if not torch.equal(ybuf, ref):
            bad += 1
            worst = max(worst, float((ybuf.float() - ref.float()).abs().max()))
    if bad:
        print(f"{bad:2d} / {NREP} differ (max {worst:.2e})  with aggressor {key}", flush=True)
print("done")
