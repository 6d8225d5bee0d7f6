"""PSNR / SSIM / RMSE of Trainer.test on the device (reference: src/util.py:188-236)."""
import ctypes as C

import torch

from . import _lib as L


def compute_metrics(pred, target):
    """pred, target: (B,1,H,W) or (B,H,W) fp32 device tensors in [0,1] -> (B,3) = PSNR, SSIM, RMSE."""
    pred = pred.contiguous().float()
    target = target.contiguous().float()
    if pred.shape != target.shape:
        raise TypeError(f"Expected tensors of equal shapes, but got {pred.shape} and {target.shape}")
    H, W = pred.shape[-2:]
    B = pred.numel() // (H * W)
    nblk = L.lib().fd_metrics_nblk(H, W)
    ws = torch.empty(B, nblk, 2, device=pred.device)
    out = torch.empty(B, 3, device=pred.device)
    stream = C.c_void_p(torch.cuda.current_stream(pred.device).cuda_stream)
    L.call("fd_metrics", pred.data_ptr(), target.data_ptr(), B, H, W, ws.data_ptr(), out.data_ptr(), stream)
    return out


def compute_psnr(input, target, max_val=1.0):
    return compute_metrics(input, target)[:, 0].mean()


def compute_ssim(img1, img2):
    return compute_metrics(img1, img2)[:, 1].mean()


def compute_rmse(input, target):
    return compute_metrics(input, target)[:, 2].mean()
