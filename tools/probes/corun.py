#!/usr/bin/env python3
"""Probe (round 4): do two DIFFERENT kernels of the level-0 Mamba block overlap when they run on two streams, and does
capping their workgroups per CU (FD_PAD_<FAMILY> / FD_ROWS_PER_CU) help?  For each pair: t_A and t_B alone (N launches
each on its own stream, one after the other) against both streams launching concurrently.  gain = (t_A + t_B) / t_AB.
The production sampler runs two half-batches on two streams; rocprof shows its big kernels barely overlap (durations
x1.04-1.10 with two streams): every kernel is tuned to fill a CU's LDS / registers by itself.
usage: [FD_PAD_SCAN=20 FD_PAD_PWDW=28 FD_ROWS_PER_CU=2 FD_PAD_CONV3=..] python tools/probes/corun.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from founddiff_amd import _lib as L  # noqa: E402
from founddiff_amd.engine import DAEngine, ConvW, _T  # noqa: E402

B, H, W, C, D, N, R = 8, 512, 512, 64, 128, 4, 4
dev = torch.device("cuda")


class Bare(DAEngine):
    def __init__(self):
        self.mode = "bf16"
        self.dt, self.tdt = _T["bf16"]
        self.dev = dev
        self.buf = {}


e = Bare()
torch.manual_seed(0)
bf = torch.bfloat16
xc = (torch.randn(B, H, W, D, device=dev) * 0.5).to(bf)
xz = (torch.randn(B, H, W, 2 * D, device=dev) * 0.5).to(bf)
x = torch.randn(B, H, W, C, device=dev).to(bf)
x1 = torch.empty(B, H, W, C, device=dev, dtype=bf)
y = torch.empty(B, H, W, D, device=dev, dtype=bf)
CD, Lq = R + 2 * N, (H // 2) * (W // 2)
xw = (torch.randn(4, CD, D, device=dev) * D ** -0.5).to(bf)
xdbl = torch.empty(4, B, Lq, CD, device=dev)
dtw = (torch.rand(4, D, R, device=dev) * 2 - 1) * R ** -0.5
dtb = torch.randn(4, D, device=dev) * 0.5 - 3
A = -torch.exp(torch.log(torch.arange(1, N + 1, device=dev).float())[None].repeat(4 * D, 1))
Ds = torch.ones(4 * D, device=dev)
ws = torch.empty(L.lib().fd_scan_ws_floats(B, H, W, D, N), device=dev)
mod = torch.randn(B, 384, device=dev) * 0.5
loc = torch.randn(B, D, device=dev) * 0.5
g64, b64 = torch.randn(64, device=dev), torch.randn(64, device=dev)
g128, b128 = torch.randn(D, device=dev), torch.randn(D, device=dev)
wpw = (torch.randn(256, 64, device=dev) / 8).to(bf)
wm = DAEngine._dw_masked((torch.randn(128, 9, device=dev) / 3).t().contiguous())
bdw = torch.randn(128, device=dev)
w_out = ConvW(torch.randn(C, D) / D ** 0.5, None, dev, bf)
w3 = ConvW(torch.randn(64, 64, 3, 3) / 24, torch.randn(64), dev, bf)
hraw = torch.empty(B, H, W, 64, device=dev, dtype=bf)
part = torch.empty(B, L.lib().fd_conv_mtiles(H, W), 64, 2, device=dev)
xc2 = torch.empty(B, H, W, D, device=dev, dtype=bf)
xz2 = torch.empty(B, H, W, 2 * D, device=dev, dtype=bf)


def cur():
    return torch.cuda.current_stream().cuda_stream


def k_scan():
    L.call("fd_selective_scan_xproj", L.FD_BF16, xc.data_ptr(), xw.data_ptr(), xdbl.data_ptr(), dtw.data_ptr(), dtb.data_ptr(),
           A.data_ptr(), Ds.data_ptr(), y.data_ptr(), ws.data_ptr(), B, H, W, D, N, R, cur())


def k_outproj():
    e.conv(w_out, y, B, H, W, x1, epi=L.EPI_GATE_RES, res=x, gate=mod, gate_ld=384, prologue=L.PRO_LN_GATE, ln_gamma=g128,
           ln_beta=b128, ln_eps=1e-5, ln_shift=loc, ln_ld=D, ln_z=xz, ln_ldz=2 * D, ln_offz=D)


def k_pwdw():
    L.call("fd_pw_dw3x3", L.FD_BF16, x.data_ptr(), 64, 0, 64, g64.data_ptr(), b64.data_ptr(), 1e-5, mod.data_ptr(), mod.data_ptr() + 256,
           384, wpw.data_ptr(), 128, wm.data_ptr(), bdw.data_ptr(), 1, xc2.data_ptr(), 128, 0, 128, xz2.data_ptr(), 256, 128, B, H, W, cur())


def k_conv3():
    e.conv(w3, x, B, H, W, hraw, c0=64, stats=part)


KERN = {"scan": k_scan, "outproj": k_outproj, "pwdw": k_pwdw, "conv3": k_conv3}
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
NIT = 10


def run(fa, fb):
    """wall time (ms) of NIT launches of fa on s1 and of fb on s2 (fb may be None)"""
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(NIT):
        with torch.cuda.stream(s1):
            fa()
        if fb is not None:
            with torch.cuda.stream(s2):
                fb()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3 / NIT


for f in KERN.values():
    for _ in range(2):
        f()
torch.cuda.synchronize()
alone = {k: min(run(f, None) for _ in range(3)) for k, f in KERN.items()}
print("knobs:", {k: v for k, v in os.environ.items() if k.startswith("FD_PAD") or k.startswith("FD_ROWS")})
print("alone (ms):", {k: round(v, 3) for k, v in alone.items()})
names = list(KERN)
for i, a in enumerate(names):
    for b in names[i:]:
        t = min(run(KERN[a], KERN[b]) for _ in range(3))
        print(f"{a:8s} + {b:8s}: serial {alone[a] + alone[b]:.3f} ms, concurrent {t:.3f} ms, gain x{(alone[a] + alone[b]) / t:.3f}", flush=True)
