import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from founddiff_amd import _lib as L
from founddiff_amd.engine import DAEngine, ConvW, _T
class Bare(DAEngine):
    def __init__(self, mode):
        self.mode = mode; self.dt, self.tdt = _T[mode]; self.dev = torch.device("cuda"); self.buf = {}
e = Bare("bf16")
torch.manual_seed(0)
B, H, W = 2, 256, 256
hw = H * W
x = (torch.randn(B, hw, 64) * 1.5).cuda().to(torch.bfloat16)
w = (torch.randn(256, 64) / 8)
mod = (torch.randn(B, 384) * 0.5).cuda()
g, b_ = torch.randn(64).cuda(), torch.randn(64).cuda()
cw = ConvW(w, None, e.dev, e.tdt)
outs = []
for r in range(4):
    out = torch.zeros(B, H, W, 256, device="cuda", dtype=torch.bfloat16)
    e.conv(cw, x, B, H, W, out, epi=L.EPI_SILU_SPLIT, split=128, prologue=L.PRO_LN_MOD, ln_gamma=g, ln_beta=b_,
           ln_eps=1e-5, ln_shift=C.c_void_p(mod.data_ptr()), ln_scale=C.c_void_p(mod.data_ptr() + 256), ln_ld=384)
    torch.cuda.synchronize()
    outs.append(out.reshape(B, hw, 256).float())
for r in range(1, 4):
    d = (outs[r] != outs[0])
    print("run", r, "ndiff", int(d.sum()))
    if d.any():
        idx = d.nonzero()
        print(" batches", idx[:, 0].unique().tolist(), "pix%32", (idx[:, 1] % 32).unique().tolist()[:40],
              "chan%32", (idx[:, 2] % 32).unique().tolist()[:40])
        print(" pix range", int(idx[:, 1].min()), int(idx[:, 1].max()), "n pix", idx[:, 1].unique().numel())
        i = idx[0]
        print(" sample", i.tolist(), float(outs[0][tuple(i)]), float(outs[r][tuple(i)]))
