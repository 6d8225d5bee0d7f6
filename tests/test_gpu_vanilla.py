"""GPU parity of the vanilla DDPM path (src/denoising_diffusion_pytorch.py mirror) vs goldens
captured from the reference: modules, one U-Net forward, 10-step DDIM for the three objectives and
ancestral steps (config 1 of BASELINE.json: 64x64-class patch, Unet(dim=32, dim_mults=(1,2)))."""
import pytest
import torch

from conftest import rel_err

pytestmark = pytest.mark.gpu


def nhwc(x, tdt):
    return x.permute(0, 2, 3, 1).contiguous().to("cuda", tdt)


def nchw(x):
    return x.float().cpu().permute(0, 3, 1, 2).contiguous()


def _bare(mode):
    from founddiff_amd.denoising_diffusion_pytorch import VanillaEngine
    from founddiff_amd.engine import _T

    class B_(VanillaEngine):
        def __init__(self):
            self.mode = mode
            self.dt, self.tdt = _T[mode]
            self.dev = torch.device("cuda")
            self.buf = {}
            self._films = []
    return B_()


@pytest.mark.parametrize("mode,tol", [("fp32", 1e-4), ("bf16", 3e-2)])
def test_vanilla_modules(golden, mode, tol):
    from founddiff_amd.engine import _Sub
    g = golden("modules_vanilla")
    e = _bare(mode)
    # ResnetBlock with FiLM, two-source input (48 = 32 + 16)
    r = e._vres(_Sub(g.weights("vrb."), "vrb."))
    x, t = g["vrb.x"], g["vrb.t"]
    B, Cin, H, W = x.shape
    r["film_off"] = 0
    e.film_total = r["film_w"].shape[0]
    e.film_all = torch.empty(B, e.film_total, device="cuda")
    e.linear(t.cuda(), r.pop("film_w").cuda(), r.pop("film_b").cuda(), e.film_all, pre_silu=True)
    xd = nhwc(x, e.tdt)
    out = e.vres_block(r, xd[..., :32].contiguous(), 32, xd[..., 32:].contiguous(), 16, B, H, W, "t")
    torch.cuda.synchronize()
    assert rel_err(nchw(out), g["vrb.out"]) < tol
    # LinearAttention and Attention (with PreNorm + Residual)
    a = e._lin(_Sub(g.weights("vlin."), "vlin.fn."))
    x = g["vlin.x"]
    out = e.lin_attn(a, nhwc(x, e.tdt), x.shape[0], x.shape[2], x.shape[3], "l")
    torch.cuda.synchronize()
    assert rel_err(nchw(out), g["vlin.out"]) < tol
    w = g.weights("vatt.")
    s = _Sub(w, "vatt.fn.")
    att = dict(g=e._f(s["norm.g"].reshape(-1)), qkv=e._convw(s["fn.to_qkv.weight"]),
               out=e._convw(s["fn.to_out.weight"], s["fn.to_out.bias"]))
    x = g["vatt.x"]
    out = e.full_attn(att, nhwc(x, e.tdt), x.shape[0], x.shape[2], x.shape[3], "a")
    torch.cuda.synchronize()
    assert rel_err(nchw(out), g["vatt.out"]) < tol


def _model(golden, obj, S, sched="cosine", precision="fp32"):
    from founddiff_amd.denoising_diffusion_pytorch import GaussianDiffusion, Unet
    g = golden("e2e_vanilla_tiny")
    net = Unet(32, dim_mults=(1, 2), channels=1, precision=precision)
    dif = GaussianDiffusion(net, image_size=64, timesteps=1000, sampling_timesteps=S, objective=obj, beta_schedule=sched)
    missing, unexpected = dif.load_state_dict(g.weights("model."), strict=False)
    assert not [k for k in missing if k.startswith("model.")] and not unexpected
    return g, dif.to("cuda")


def test_vanilla_schedule_buffers(golden):
    g = golden("schedule")
    for sched in ("linear", "cosine"):
        _, dif = _model(golden, "pred_noise", 10, sched)
        for k, v in dif.named_buffers(recurse=False):
            assert torch.equal(v.cpu(), g[f"vanilla.{sched}.{k}"]), (sched, k)


def test_vanilla_unet_and_ddim_fp32(golden):
    g, dif = _model(golden, "pred_noise", 10)
    out = dif.model(g["unet.x"].cuda(), g["unet.t"].cuda())
    assert rel_err(out.cpu(), g["unet.out"]) < 1e-3
    for obj in ("pred_noise", "pred_x0", "pred_v"):
        g, dif = _model(golden, obj, 10)
        res = dif.sample(batch_size=2, noise=g[f"ddim.{obj}.xT"].cuda())
        e_ = rel_err(res[0].cpu(), g[f"ddim.{obj}.out"])
        print(obj, "ddim rel_err", e_)
        assert e_ < 1e-3, (obj, e_)


def test_vanilla_ancestral_fp32(golden):
    g, dif = _model(golden, "pred_noise", 1000, "linear")
    img = g["ddim.pred_v.xT"].cuda()
    for i, t in enumerate(range(999, 993, -1)):
        img, _ = dif.p_sample(img, t, noise=g["anc.noise"][i].cuda())
        assert rel_err(img.cpu(), g["anc.imgs"][i]) < 1e-3, t


def test_vanilla_bf16_runs(golden):
    g, dif = _model(golden, "pred_noise", 10, precision="bf16")
    res = dif.sample(batch_size=2, noise=g["ddim.pred_noise.xT"].cuda())
    assert torch.isfinite(res[0]).all()
    # ... and on the library's binary16 build (precision='fp16': the same kernels, three more significand bits in every stored
    # tensor): the one-UNet forward against the reference, closer than the bf16 engine's
    outs = {}
    for prec in ("bf16", "fp16"):
        g, d = _model(golden, "pred_noise", 10, precision=prec)
        outs[prec] = rel_err(d.model(g["unet.x"].cuda(), g["unet.t"].cuda()).cpu(), g["unet.out"])
    print(f"vanilla UNet forward vs the reference: bf16 {outs['bf16']:.2e}, fp16 {outs['fp16']:.2e} max-rel")
    assert outs["fp16"] < 0.4 * outs["bf16"] and outs["fp16"] < 5e-3, outs


@pytest.mark.parametrize("mode,tol", [("fp32", 2e-5), ("bf16", 1.5e-2)])
@pytest.mark.parametrize("n", [50, 1000, 1024, 4096])
def test_attention_mfma(mode, tol, n):
    """fd_attention (MFMA flash attention, dim_head 32, K/V tiles in LDS; src/denoising_diffusion_pytorch.py:257-279)
    against softmax(q k^T * 32^-0.5) v in fp32: token counts that are not multiples of the 64-key tile / the
    128-query workgroup, and the 4096 tokens of the 512x512 bottleneck."""
    from founddiff_amd import _lib as L
    from founddiff_amd.engine import _T
    dt, tdt = _T[mode]
    torch.manual_seed(n)
    B, heads, hidden = 2, 4, 128
    qkv = torch.randn(B, n, 3 * hidden) * 1.5
    if mode == "bf16":
        qkv = qkv.to(torch.bfloat16).float()
    q, k, v = [t.reshape(B, n, heads, 32).permute(0, 2, 1, 3) for t in qkv.chunk(3, dim=-1)]
    attn = torch.softmax((q * 32 ** -0.5) @ k.transpose(-1, -2), dim=-1)
    ref = (attn @ v).permute(0, 2, 1, 3).reshape(B, n, hidden)
    qd = qkv.to("cuda", tdt).contiguous()
    out = torch.empty(B, n, hidden, device="cuda", dtype=tdt)
    L.call("fd_attention", dt, qd.data_ptr(), out.data_ptr(), B, n, hidden, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert rel_err(out.float().cpu(), ref) < tol
