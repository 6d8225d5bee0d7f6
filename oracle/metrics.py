"""ORACLE (test infrastructure): PSNR / SSIM / RMSE as Trainer.test computes them.

Restates /root/reference/src/util.py:188-236.  compute_psnr / compute_rmse are torch-only in the
reference; compute_ssim depends on kornia (get_gaussian_kernel2d((11,11),(1.5,1.5)) + filter2d with
its default 'reflect' border), which is absent from this image, so the SSIM here follows kornia's
published definition and is cross-checked against scipy.ndimage in tests/test_oracle_golden.py:
PARITY UNPINNED for SSIM against the reference itself.
"""
import torch
import torch.nn.functional as F


def psnr(a, b, max_val=1.0):
    mse = F.mse_loss(a, b, reduction="mean")
    return 10 * torch.log10(torch.tensor(max_val * max_val) / mse)


def rmse(a, b):
    return torch.sqrt(F.mse_loss(a, b))


def gaussian_window(size=11, sigma=1.5):
    x = torch.arange(size, dtype=torch.float32) - size // 2
    g = torch.exp(-x ** 2 / (2 * sigma ** 2))
    g = g / g.sum()
    return g[:, None] * g[None, :]


def _filt(x, k):
    r = k.shape[-1] // 2
    return F.conv2d(F.pad(x, (r, r, r, r), mode="reflect"), k[None, None])


def ssim(a, b, max_val=1.0):
    """a, b (B,1,H,W).  Mean over everything of clamp(ssim_map, 0, 1)."""
    k = gaussian_window()
    C1, C2 = (0.01 * max_val) ** 2, (0.03 * max_val) ** 2
    mu1, mu2 = _filt(a, k), _filt(b, k)
    s1 = _filt(a * a, k) - mu1 ** 2
    s2 = _filt(b * b, k) - mu2 ** 2
    s12 = _filt(a * b, k) - mu1 * mu2
    m = ((2 * mu1 * mu2 + C1) * (2 * s12 + C2)) / ((mu1 ** 2 + mu2 ** 2 + C1) * (s1 + s2 + C2))
    return torch.clamp(m, 0, 1).mean()
