"""Drop-in for the reference's one native extension, `selective_scan_cuda_core` (VMamba
kernels/selective_scan; imported at /root/reference/src/emamba2.py:20-27, called at 154):

    out, x, *rest = selective_scan_cuda_core.fwd(u, delta, A, B, C, D, delta_bias, delta_softplus, nrows)

Forward only (this is a sampling engine): `bwd` raises.  To run the reference's own `emamba2.py` on MI355X with
nothing else changed, put this module under that name before `src.emamba2` is imported:

    import sys, founddiff_amd.selective_scan_cuda_core as m
    sys.modules["selective_scan_cuda_core"] = m

Arguments as the extension takes them (src/emamba2.py:124-149): fp32 CUDA/HIP tensors (fp16 / bf16 are up-cast, as the
reference's custom_fwd(cast_inputs=float32) does), last dim contiguous,
u / delta (b, KD, L), A (KD, N), B / C (b, K, N, L), D / delta_bias (KD) or None.  Returns `(out, x)`: out
(b, KD, L); x = the state after the last position, (b, KD, N) -- the extension returns its backward pass's chunk
states there, which a forward-only library has no use for (the reference's forward ignores `x` except to save
it for backward, src/emamba2.py:156).  Errors follow the reference's asserts (nrows in 1..4, KD divisible by
K * nrows) and surface as RuntimeError.
"""
import ctypes as C

import torch

from . import _lib as L


def _chk(name, t, ndim):
    if t.dtype in (torch.float16, torch.bfloat16):
        t = t.float()          # the reference's autograd wrapper casts its inputs to fp32 (custom_fwd, src/emamba2.py:127)
    if t.dtype != torch.float32:
        raise RuntimeError(f"selective_scan_cuda_core.fwd: {name} must be float32 / float16 / bfloat16 (got {t.dtype})")
    if t.dim() != ndim:
        raise RuntimeError(f"selective_scan_cuda_core.fwd: {name} must be {ndim}-dimensional (got {tuple(t.shape)})")
    if not t.is_cuda:
        raise RuntimeError(f"selective_scan_cuda_core.fwd: {name} must live on the GPU (there is no CPU path)")
    return t if t.is_contiguous() else t.contiguous()


def fwd(u, delta, A, B, C_, D=None, delta_bias=None, delta_softplus=False, nrows=1):
    u, delta = _chk("u", u, 3), _chk("delta", delta, 3)
    A = _chk("A", A, 2)
    if B.dim() == 3:            # (b, N, L): a single group, as the autograd wrapper un-squeezes it (emamba2.py:144-149)
        B = B.unsqueeze(1)
    if C_.dim() == 3:
        C_ = C_.unsqueeze(1)
    B, C_ = _chk("B", B, 4), _chk("C", C_, 4)
    b, KD, Ln = u.shape
    K, N = B.shape[1], A.shape[1]
    if delta.shape != u.shape or A.shape[0] != KD or tuple(B.shape) != (b, K, N, Ln) or C_.shape != B.shape:
        raise RuntimeError(f"selective_scan_cuda_core.fwd: inconsistent shapes u{tuple(u.shape)} delta{tuple(delta.shape)} "
                           f"A{tuple(A.shape)} B{tuple(B.shape)} C{tuple(C_.shape)}")
    if D is not None:
        D = _chk("D", D, 1)
    if delta_bias is not None:
        delta_bias = _chk("delta_bias", delta_bias, 1)
    for name, t in (("delta", delta), ("A", A), ("B", B), ("C", C_), ("D", D), ("delta_bias", delta_bias)):
        if t is not None and t.device != u.device:
            raise RuntimeError(f"selective_scan_cuda_core.fwd: {name} lives on {t.device}, u on {u.device}")
    p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)
    with torch.cuda.device(u.device):          # the launch goes to u's GPU, whatever the current device is
        out = torch.empty_like(u)
        x = torch.empty(b, KD, N, device=u.device, dtype=torch.float32)
        stream = C.c_void_p(torch.cuda.current_stream(u.device).cuda_stream)
        L.call("fd_selective_scan_fwd_f32", p(u), p(delta), p(A), p(B), p(C_), p(D), p(delta_bias),
               int(bool(delta_softplus)), int(nrows), b, KD, K, N, Ln, p(out), p(x), stream)
    return out, x


def bwd(*a, **k):
    raise NotImplementedError("founddiff_amd is a sampling engine: selective_scan_cuda_core.bwd is not built")
