import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.nn.functional as F
from founddiff_amd import _lib as L
from founddiff_amd.engine import DAEngine, ConvW, _T


class Bare(DAEngine):
    def __init__(self, mode):
        self.mode = mode
        self.dt, self.tdt = _T[mode]
        self.dev = torch.device("cuda")
        self.buf = {}
        self.f32_split = int(mode == "fp32s")


torch.manual_seed(0)
for (cin, cout, k, hw) in [(64, 64, 3, (24, 20)), (32, 64, 1, (16, 16)), (8, 64, 1, (16, 16))]:
    H, W = hw
    for kind in ("ones", "rand"):
        x = torch.ones(1, cin, H, W) if kind == "ones" else torch.randn(1, cin, H, W)
        w = torch.ones(cout, cin, k, k) / (cin * k * k) if kind == "ones" else torch.randn(cout, cin, k, k) / (cin * k * k) ** 0.5
        ref = F.conv2d(x, w, None, padding=k // 2)
        for mode in ("fp32", "fp32s"):
            e = Bare(mode)
            cw = ConvW(w, None, e.dev, e.tdt)
            out = torch.empty(1, H, W, cout, device="cuda")
            e.conv(cw, x.permute(0, 2, 3, 1).contiguous().cuda(), 1, H, W, out, pad=k // 2)
            torch.cuda.synchronize()
            o = out.cpu().permute(0, 3, 1, 2)
            print(cin, cout, k, kind, mode, "err", float((o - ref).norm() / ref.norm()), "out[0,0,5,5]", float(o[0, 0, 5, 5]), "ref", float(ref[0, 0, 5, 5]))
