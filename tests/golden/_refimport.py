"""Import the read-only reference (/root/reference) in THIS container to generate
golden vectors.  Never shipped to / used on the GPU box: only make_golden.py
imports this file, and the fixtures it writes are plain data.

The reference cannot be imported as-is: it depends on
packages that are absent (torchvision, ema_pytorch, open_clip, lpips, timm,
clip, kornia, ...), downloads CLIP weights from the network, reads
`Dose-CLIP.pth` from the CWD, and calls a third-party CUDA selective-scan
extension.  We stub the absent packages with inert modules, patch the network
loads to build seeded random models, and inject a pure-torch sequential
selective scan with the published `selective_scan_ref` semantics.
"""
import importlib.abc
import importlib.machinery
import os
import sys
import types

import torch
import torch.nn as nn

REF = "/root/reference"

_STUB_ROOTS = {
    "torchvision", "ema_pytorch", "wandb", "Augmentor", "cv2", "ipdb", "open_clip",
    "lpips", "timm", "clip", "kornia", "pywt", "skimage", "lmdb",
    "selective_scan_vmamba_pt202",
}


class _Anything:
    """Inert placeholder: callable, subscriptable, attribute access returns itself."""

    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return _Anything()

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return _Anything()


class _StubModule(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return _Anything


class _StubFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path, target=None):
        if fullname.split(".")[0] in _STUB_ROOTS:
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        m = _StubModule(spec.name)
        m.__path__ = []
        return m

    def exec_module(self, module):
        pass


def scan_ref(u, delta, A, B, C, D=None, delta_bias=None, delta_softplus=False, nrows=1):
    """Sequential selective scan with the public `selective_scan_ref` semantics
    (the call site is src/emamba2.py:154; the CUDA op itself is not in the tree).
    u, delta: (b, KD, L); A: (KD, N); B, C: (b, K, N, L); D, delta_bias: (KD)."""
    b, KD, L = u.shape
    K = B.shape[1]
    N = A.shape[1]
    Dg = KD // K
    delta = delta + delta_bias[None, :, None] if delta_bias is not None else delta
    if delta_softplus:
        delta = torch.nn.functional.softplus(delta)
    Bx = B.repeat_interleave(Dg, dim=1)  # (b, KD, N, L)
    Cx = C.repeat_interleave(Dg, dim=1)
    h = torch.zeros(b, KD, N, dtype=torch.float32)
    ys = []
    for t in range(L):
        dA = torch.exp(delta[:, :, t, None] * A[None])
        dBu = delta[:, :, t, None] * Bx[:, :, :, t] * u[:, :, t, None]
        h = dA * h + dBu
        ys.append((h * Cx[:, :, :, t]).sum(-1))
    y = torch.stack(ys, dim=-1)
    if D is not None:
        y = y + D[None, :, None] * u
    return y, h


def install(clip_cfg=None, workdir="/tmp/fd_golden"):
    """Make `import src.DADiff` work.  clip_cfg: kwargs tuple for src.DACLIP.CLIP
    (defaults to a shrunken RN: layers (1,1,1,1), width 8)."""
    os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
    sys.dont_write_bytecode = True
    os.makedirs(workdir, exist_ok=True)
    os.chdir(workdir)
    sys.meta_path.insert(0, _StubFinder())
    if REF not in sys.path:
        sys.path.insert(0, REF)
    # Q12: an installed HF `datasets` package shadows the reference's namespace dir
    ds = types.ModuleType("datasets")
    ds.__path__ = [os.path.join(REF, "datasets")]
    sys.modules["datasets"] = ds

    import timm.models.layers as tl  # stub
    tl.DropPath = nn.Identity
    tl.trunc_normal_ = lambda t, *a, **k: t
    import timm.models.registry as tr
    tr.register_model = lambda f: f
    import lpips

    class _LP(nn.Module):
        def __init__(self, *a, **k):
            super().__init__()
    lpips.LPIPS = _LP
    import clip as clipmod

    def _tokenize(texts, *a, **k):
        out = torch.zeros(len(texts), 77, dtype=torch.long)
        for i in range(len(texts)):
            out[i, :24] = torch.arange(1, 25)
            out[i, 24] = 500  # EOT = argmax; must stay < the shrunken vocab
        return out
    clipmod.tokenize = _tokenize
    import selective_scan_vmamba_pt202 as ssv

    class _Core:
        @staticmethod
        def fwd(u, delta, A, B, C, D, delta_bias, delta_softplus, nrows):
            y, h = scan_ref(u, delta, A, B, C, D, delta_bias, delta_softplus, nrows)
            return y, h
    ssv.selective_scan_cuda_core = _Core

    if clip_cfg is None:
        clip_cfg = dict(embed_dim=1024, image_resolution=224, vision_layers=(1, 1, 1, 1),
                        vision_width=8, vision_patch_size=None, context_length=77,
                        vocab_size=512, transformer_width=32, transformer_heads=2,
                        transformer_layers=1)

    import src.DACLIP as DACLIP
    import src.model_clipiqa as MC

    def _load(name="RN50", device="cpu", **k):
        g = torch.random.get_rng_state()
        torch.manual_seed(1234)
        m = DACLIP.CLIP(**clip_cfg).float().eval()
        torch.random.set_rng_state(g)
        return m

    DACLIP.load = _load
    MC.load = _load
    tw = clip_cfg["transformer_width"]

    def _lffu(*a, **k):
        p = os.path.join(workdir, "ctx.pth")
        torch.manual_seed(4321)
        torch.save(torch.randn(2, 16, tw) * 0.02, p)
        return p
    DACLIP.load_file_from_url = _lffu
    # Dose-CLIP.pth read from CWD by Unet.__init__ (src/DADiff.py:595)
    torch.manual_seed(99)
    iqa = DACLIP.CLIPIQA(model_type="clipiqa+")
    torch.save(iqa.state_dict(), os.path.join(workdir, "Dose-CLIP.pth"))
    import src.DADiff as DADiff
    DADiff.load = _load
    return DADiff
