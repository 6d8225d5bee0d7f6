import torch, torch.nn.functional as F, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from founddiff_amd import _lib as L
torch.manual_seed(4)
B, Cc, H, W = 2, 48, 9, 14
x = torch.randn(B, Cc, H, W).bfloat16().float()
w, b = torch.randn(Cc, 1, 3, 3) / 3, torch.randn(Cc)
ref = F.conv2d(x, w, b, padding=1, groups=Cc)
xd = x.permute(0, 2, 3, 1).contiguous().cuda().bfloat16()
out = torch.empty(B, H, W, Cc, device="cuda", dtype=torch.bfloat16)
wd, bd = w.reshape(Cc, 9).t().contiguous().cuda(), b.cuda()
L.call("fd_dwconv3x3", 1, xd.data_ptr(), Cc, 0, wd.data_ptr(), bd.data_ptr(), 0, out.data_ptr(), Cc, 0, B, H, W, Cc, None)
torch.cuda.synchronize()
o = out.float().cpu().permute(0, 3, 1, 2)
err = (o - ref).abs()
print("by batch", err.amax((1, 2, 3)))
print("by chan", err.amax((0, 2, 3)))
print("by row", err.amax((0, 1, 3)))
print("by col", err.amax((0, 1, 2)))
