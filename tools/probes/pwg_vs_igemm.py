"""kid 7 (persistent pointwise GEMM) against kid 5 (generic 256x256 tile) on the GATE_RES layer of the 64x64 level, per build of the
library: how many elements differ, where, by how much; with the epilogue's inputs simplified one at a time."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from founddiff_amd import _lib as L
from founddiff_amd.engine import ConvW, DAEngine, _T
for mode in ("bf16", "fp16"):
    class Bare(DAEngine):
        def __init__(self):
            self.mode = mode; self.dt, self.tdt = _T[mode]; self.hip = L.F16 if mode == "fp16" else L.BF16
            self.dev = torch.device("cuda"); self.buf = {}; self.f32_split = 0
    e = Bare()
    torch.manual_seed(5)
    B, H, W, c0, cout = 8, 64, 64, 1024, 512
    hd = e.tdt
    for variant in ("full", "gate=1", "res=0", "bias=0", "gate=1,res=0,bias=0"):
        a = torch.randn(B, H, W, c0).to(hd)
        w = (torch.randn(cout, c0) / c0 ** 0.5)
        bias = torch.zeros(cout) if "bias=0" in variant else torch.randn(cout)
        res = (torch.zeros(B, H, W, cout) if "res=0" in variant else torch.randn(B, H, W, cout)).to(hd).cuda()
        gate = (torch.ones(B, cout) if "gate=1" in variant else torch.randn(B, cout)).cuda()
        cw = ConvW(w, bias, e.dev, e.tdt)
        ad = a.cuda()
        kw = dict(c0=c0, epi=L.EPI_GATE_RES, res=res, gate=gate, gate_ld=cout)
        o7 = torch.zeros(B, H, W, cout, device="cuda", dtype=hd)
        assert e.conv(cw, ad, B, H, W, o7, probe="kid", **kw) == 7
        e.conv(cw, ad, B, H, W, o7, **kw)
        h0 = c0 // 2
        kw2 = dict(kw, c0=h0, in1=ad[..., h0:].contiguous(), c1=c0 - h0)
        a0 = ad[..., :h0].contiguous()
        o5 = torch.zeros_like(o7)
        assert e.conv(cw, a0, B, H, W, o5, probe="kid", **kw2) == 5
        e.conv(cw, a0, B, H, W, o5, **kw2)
        torch.cuda.synchronize()
        d = (o7.float() - o5.float()).abs()
        nz = d.nonzero()
        msg = "equal" if len(nz) == 0 else f"{len(nz)} of {d.numel()} differ, max {float(d.max()):.3e}; channels mod 8: {sorted(set((nz[:, 3] % 8).tolist()))}, pixels mod 16: {sorted(set(((nz[:, 1] * W + nz[:, 2]) % 16).tolist()))[:16]}"
        print(f"{mode} {variant}: {msg}", flush=True)
