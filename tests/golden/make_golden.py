#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE itself (CPU, this container only).

    python tests/golden/make_golden.py [schedule|modules|modules_odd|e2e_da|e2e_variants|e2e_vanilla|full64|all]

Each fixture is data only: seeded inputs, the reference's outputs, and the {key: shape} spec +
seed from which founddiff_amd.synth regenerates the exact weights that were loaded into the
reference (weights themselves are not stored).  /root/reference never travels to the GPU box.
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

import _refimport  # noqa: E402
from founddiff_amd import synth  # noqa: E402

torch.set_grad_enabled(False)
SEED_W = 0

TINY_CLIP = dict(embed_dim=1024, image_resolution=224, vision_layers=(2, 1, 1, 1),
                 vision_width=16, vision_patch_size=None, context_length=77,
                 vocab_size=512, transformer_width=32, transformer_heads=2,
                 transformer_layers=1)
FULL_CLIP = dict(embed_dim=1024, image_resolution=224, vision_layers=(3, 4, 6, 3),
                 vision_width=64, vision_patch_size=None, context_length=77,
                 vocab_size=49408, transformer_width=512, transformer_heads=8,
                 transformer_layers=12)


def load_synth(module, seed=SEED_W, prefix=""):
    sd = module.state_dict()
    spec = synth.spec_of(sd)
    new = synth.synth_state_dict({prefix + k: v for k, v in spec.items()}, seed)
    module.load_state_dict({k: new[prefix + k] for k in sd}, strict=True)
    return {prefix + k: v for k, v in spec.items()}


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def save(name, spec=None, seed=SEED_W, **arrs):
    """One .npz per fixture, written so that the FILE is reproducible byte for byte: keys in sorted order, the spec as
    sorted-key JSON, zip members with a fixed timestamp (np.savez stamps them with the wall clock)."""
    import io
    import zipfile
    out = {k: (v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v))
           for k, v in arrs.items()}
    if spec is not None:
        out["spec_json"] = np.frombuffer(json.dumps(spec, sort_keys=True).encode(), dtype=np.uint8)
        out["weight_seed"] = np.asarray(seed)
    path = os.path.join(os.environ.get("FD_GOLDEN_OUT", HERE), name + ".npz")     # tests regenerate into a temp dir
    with zipfile.ZipFile(path, "w", compression=zipfile.ZIP_DEFLATED, compresslevel=6) as zf:
        for k in sorted(out):
            buf = io.BytesIO()
            a = out[k]
            np.lib.format.write_array(buf, np.ascontiguousarray(a) if a.ndim else a, allow_pickle=False)
            zi = zipfile.ZipInfo(k + ".npy", date_time=(1980, 1, 1, 0, 0, 0))
            zi.compress_type = zipfile.ZIP_DEFLATED
            zi.external_attr = 0o644 << 16
            zf.writestr(zi, buf.getvalue(), compresslevel=6)
    print(f"wrote {path}  ({os.path.getsize(path) / 1024:.1f} KiB)")


# ---------------------------------------------------------------------------------------
def g_schedule(D):
    class _M(torch.nn.Module):
        channels, out_dim, self_condition, random_or_learned_sinusoidal_cond = 1, 1, False, False
    arrs = {}
    for S in (2, 10, 25, 50):
        d = D.ResidualDiffusion(_M(), image_size=64, timesteps=1000, sampling_timesteps=S,
                                objective="pred_res", loss_type="l2", condition=True, sum_scale=0.01)
        if S == 2:
            for k in ("alphas", "alphas_cumsum", "one_minus_alphas_cumsum", "betas2", "betas",
                      "betas2_cumsum", "betas_cumsum", "posterior_mean_coef1", "posterior_mean_coef2",
                      "posterior_mean_coef3", "posterior_variance", "posterior_log_variance_clipped"):
                arrs["ctor." + k] = getattr(d, k).clone()
            d.init()
            for k in list(arrs):
                kk = k.split(".", 1)[1]
                arrs["init." + kk] = getattr(d, kk).clone()
        times = torch.linspace(-1, 999, steps=S + 1)
        times = list(reversed(times.int().tolist()))
        arrs[f"pairs.{S}"] = np.asarray(list(zip(times[:-1], times[1:])), dtype=np.int64)
    from src import denoising_diffusion_pytorch as V
    for bs in ("linear", "cosine"):
        g = V.GaussianDiffusion(_M(), image_size=64, timesteps=1000, sampling_timesteps=10,
                                beta_schedule=bs)
        for k, v in g.state_dict().items():
            if not k.startswith("model."):
                arrs[f"vanilla.{bs}.{k}"] = v.clone()
    save("schedule", **arrs)


def g_modules(D):
    from src import emamba2 as E
    from src import DACLIP
    arrs, spec = {}, {}

    # time MLP (dim 32 -> time_dim 128)
    tm = torch.nn.Sequential(D.SinusoidalPosEmb(32), torch.nn.Linear(32, 128), torch.nn.GELU(),
                             torch.nn.Linear(128, 128))
    spec.update(load_synth(tm, prefix="time_mlp."))
    t_in = torch.tensor([998.5, 3.25, 0.0, 512.0])
    arrs["time_mlp.in"], arrs["time_mlp.out"] = t_in, tm(t_in)

    # Block / ResnetBlock (with and without res_conv)
    for name, cin, cout in (("rb_same", 32, 32), ("rb_proj", 48, 32)):
        rb = D.ResnetBlock(cin, cout, time_emb_dim=128, groups=8).eval()
        spec.update(load_synth(rb, prefix=name + "."))
        x = rnd(2, cin, 12, 10, seed=11)
        arrs[name + ".in"], arrs[name + ".out"] = x, rb(x)
        arrs[name + ".block_out"] = rb.block1(x)

    # down / up samplers
    dn = D.Downsample(32, 64)
    spec.update(load_synth(dn, prefix="down."))
    x = rnd(2, 32, 12, 10, seed=12)
    arrs["down.in"], arrs["down.out"] = x, dn(x)
    up = D.Upsample(64, 32)
    spec.update(load_synth(up, prefix="up."))
    x = rnd(2, 64, 6, 5, seed=13)
    arrs["up.in"], arrs["up.out"] = x, up(x)

    # EfficientScan / EfficientMerge (even and odd sizes)
    for tag, (h, w) in (("even", (6, 8)), ("odd", (5, 7))):
        x = rnd(2, 3, h, w, seed=14)
        xs = E.EfficientScan.apply(x, 2)
        arrs[f"escan_{tag}.in"], arrs[f"escan_{tag}.out"] = x, xs
        arrs[f"emerge_{tag}.out"] = E.EfficientMerge.apply(xs, h, w, 2)

    # SS2D, several (C, N) combos incl. N=32
    for tag, C, N, hw in (("c32n4", 32, 4, (8, 12)), ("c64n32", 64, 32, (6, 6)), ("c32n16", 32, 16, (10, 8))):
        m = E.SS2D(d_model=C, d_state=N, expand=2.0, dropout=0).eval()
        spec.update(load_synth(m, prefix=f"ss2d_{tag}."))
        x = rnd(2, hw[0], hw[1], C, seed=15)
        c = rnd(2, 1, 256, seed=16)
        arrs[f"ss2d_{tag}.x"], arrs[f"ss2d_{tag}.c"] = x, c
        arrs[f"ss2d_{tag}.out"] = m(x, c)
        # the inner cross_selective_scan on its own
        xi = rnd(2, 2 * C, hw[0], hw[1], seed=17, scale=0.5)
        arrs[f"ss2d_{tag}.core_in"] = xi
        arrs[f"ss2d_{tag}.core_out"] = m.forward_corev2(xi, channel_first=True)

    # TransposedAttention
    ta = D.TransposedAttention(64, 2).eval()
    spec.update(load_synth(ta, prefix="tattn."))
    x = rnd(2, 64, 8, 6, seed=18)
    arrs["tattn.in"], arrs["tattn.out"] = x, ta(x)

    # Mamba_block
    for tag, C, N, hw in (("c32", 32, 4, (8, 8)), ("c64", 64, 8, (4, 6))):
        mb = D.Mamba_block(hidden_size=C, d_state=N, expand=2.0, dropout=0, cross=False,
                           time_emb_dim=128).eval()
        spec.update(load_synth(mb, prefix=f"mamba_{tag}."))
        x = rnd(2, C, hw[0], hw[1], seed=19)
        c = rnd(2, 1, 256, seed=20)
        t = rnd(2, 128, seed=21)
        arrs[f"mamba_{tag}.x"], arrs[f"mamba_{tag}.c"], arrs[f"mamba_{tag}.t"] = x, c, t
        arrs[f"mamba_{tag}.out"] = mb(x, c, t)

    # CLIPIQA (shrunken RN) live outputs
    iqa = DACLIP.CLIPIQA(model_type="clipiqa+").eval()
    spec.update(load_synth(iqa, prefix="iqa."))
    x = rnd(2, 1, 64, 64, seed=22).clamp(-1, 1).repeat(1, 3, 1, 1)
    _, dose, ctx = iqa(x)
    arrs["iqa.in"], arrs["iqa.dose"], arrs["iqa.ctx"] = x[:, :1], dose, ctx
    save("modules", spec=spec, **arrs)


def g_modules_odd(D):
    """SS2D and Mamba_block at ODD image sizes: the reference pads the image to even sizes before the 4-way gather
    and crops after the merge (src/emamba2.py:191-199, 253-260)."""
    from src import emamba2 as E
    arrs, spec = {}, {}
    for tag, C, N, hw in (("c32n4", 32, 4, (7, 9)), ("c64n32", 64, 32, (5, 6)), ("c32n16", 32, 16, (10, 7))):
        m = E.SS2D(d_model=C, d_state=N, expand=2.0, dropout=0).eval()
        spec.update(load_synth(m, prefix=f"ss2d_{tag}."))
        x = rnd(2, hw[0], hw[1], C, seed=41)
        c = rnd(2, 1, 256, seed=42)
        arrs[f"ss2d_{tag}.x"], arrs[f"ss2d_{tag}.c"], arrs[f"ss2d_{tag}.out"] = x, c, m(x, c)
    for tag, C, N, hw in (("c32", 32, 4, (9, 7)), ("c64", 64, 8, (5, 5))):
        mb = D.Mamba_block(hidden_size=C, d_state=N, expand=2.0, dropout=0, cross=False, time_emb_dim=128).eval()
        spec.update(load_synth(mb, prefix=f"mamba_{tag}."))
        x, c, t = rnd(2, C, hw[0], hw[1], seed=43), rnd(2, 1, 256, seed=44), rnd(2, 128, seed=45)
        arrs[f"mamba_{tag}.x"], arrs[f"mamba_{tag}.c"], arrs[f"mamba_{tag}.t"] = x, c, t
        arrs[f"mamba_{tag}.out"] = mb(x, c, t)
    save("modules_odd", spec=spec, **arrs)


def g_modules_vanilla(D):
    from src import denoising_diffusion_pytorch as V
    arrs, spec = {}, {}
    rb = V.ResnetBlock(48, 32, time_emb_dim=128, groups=8).eval()
    spec.update(load_synth(rb, prefix="vrb."))
    x, t = rnd(2, 48, 8, 6, seed=31), rnd(2, 128, seed=32)
    arrs["vrb.x"], arrs["vrb.t"], arrs["vrb.out"] = x, t, rb(x, t)
    la = V.Residual(V.PreNorm(32, V.LinearAttention(32))).eval()
    spec.update(load_synth(la, prefix="vlin."))
    x = rnd(2, 32, 8, 6, seed=33)
    arrs["vlin.x"], arrs["vlin.out"] = x, la(x)
    at = V.Residual(V.PreNorm(64, V.Attention(64))).eval()
    spec.update(load_synth(at, prefix="vatt."))
    x = rnd(2, 64, 6, 6, seed=34)
    arrs["vatt.x"], arrs["vatt.out"] = x, at(x)
    save("modules_vanilla", spec=spec, **arrs)


def _da_model(D, dim, mults, S, size):
    net = D.UnetRes(dim=dim, dim_mults=mults, num_unet=1, condition=True, objective="pred_res",
                    test_res_or_noise="res")
    dif = D.ResidualDiffusion(net, image_size=size, timesteps=1000, sampling_timesteps=S,
                              objective="pred_res", loss_type="l2", condition=True, sum_scale=0.01,
                              test_res_or_noise="res").eval()
    full = dif.state_dict()
    spec = {k: v for k, v in synth.spec_of(full).items() if k.startswith("model.")}
    new = synth.synth_state_dict(spec, SEED_W)
    dif.load_state_dict(new, strict=False)   # only the 12 schedule buffers are left as built
    dif.init()
    # the fixture only needs the live keys (dead: the unused second CLIP, Q7) -- plus the names / shapes of the 12
    # schedule buffers, so that the fixture's spec is the reference's whole ResidualDiffusion.state_dict() layout
    # (tests/test_host_cpu.py::test_module_state_dict_layout); synth skips them (they are re-derived by init())
    live = {k: v for k, v in spec.items() if ".unet0.clip_model." not in k}
    live.update({k: v for k, v in synth.spec_of(full).items() if not k.startswith("model.") and not k.startswith("perceploss.")})
    return dif, live


def g_e2e_da(D):
    """config 1 on the DA path: UnetRes(32,(1,2)), 64x64, 10-step DDIM + 20 ancestral steps."""
    from founddiff_amd.synth import ct_phantom
    dif, spec = _da_model(D, 32, (1, 2), 10, 64)
    _, ld = ct_phantom(2, 64, seed=10)
    x_in = torch.from_numpy(ld)
    arrs = {"x_input": x_in}
    # DDIM: the reference draws randn(shape) exactly once (DADiff.py:1294)
    torch.manual_seed(10)
    noise0 = torch.randn(x_in.shape)
    torch.manual_seed(10)
    outs = dif.sample([x_in.clone()], batch_size=2, last=False)
    arrs["ddim.noise0"] = noise0
    arrs["ddim.imgs"] = torch.stack(outs, 0)        # [x_T01, step0..step9] in [0,1]
    torch.manual_seed(10)
    outs_last = dif.sample([x_in.clone()], batch_size=2, last=True)
    arrs["ddim.out"] = outs_last[-1]
    # one model_predictions call (API surface)
    xi = x_in * 2 - 1
    xt = xi + 0.1 * noise0
    tt = torch.full((2,), 979, dtype=torch.long)
    p = dif.model_predictions(xi, xt, tt)
    arrs["mp.pred_res"], arrs["mp.pred_noise"], arrs["mp.x_start"] = p.pred_res, p.pred_noise, p.pred_x_start
    # raw UNet output for the same call
    arrs["unet.out"] = dif.model(torch.cat((xt, xi), 1), [dif.alphas_cumsum[tt] * 1000, dif.betas_cumsum[tt] * 1000])[0]
    # ancestral: 20 p_sample steps t=999..980, noise replayed from a seeded stream
    g = torch.Generator().manual_seed(77)
    noises = torch.randn(20, *x_in.shape, generator=g)
    img = xt.clone()
    import unittest.mock as um
    anc = []
    for i, t in enumerate(range(999, 979, -1)):
        with um.patch.object(torch, "randn_like", lambda x, i=i: noises[i]):
            img, x_start = dif.p_sample(xi, img, t)
        anc.append(img.clone())
    arrs["anc.noise"] = noises
    arrs["anc.imgs"] = torch.stack(anc, 0)
    # t = 0 step (no noise, coef override path)
    with um.patch.object(torch, "randn_like", lambda x: noises[0]):
        img0, xs0 = dif.p_sample(xi, xt, 0)
    arrs["anc.t0_img"], arrs["anc.t0_xstart"] = img0, xs0
    save("e2e_da_tiny", spec=spec, **arrs)


VARIANTS = {   # objectives / dual-UNet configurations the reference supports (src/DADiff.py:817-836, 1168-1207)
    "pred_noise": dict(num_unet=1, objective="pred_noise", test="noise"),
    "res_noise": dict(num_unet=2, objective="pred_res_noise", test="res_noise"),
    "rn_noise": dict(num_unet=2, objective="pred_res_noise", test="noise"),
    "rn_res": dict(num_unet=2, objective="pred_res_noise", test="res"),
    "x0_noise": dict(num_unet=2, objective="pred_x0_noise", test="res_noise"),
    "incond": dict(num_unet=1, objective="pred_res", test="res", input_condition=True),   # third input plane (1157-1158)
    "incond_mask": dict(num_unet=1, objective="pred_res", test="res", input_condition=True, mask=True),   # 1372-1373
}


def g_e2e_da_variants(D):
    """The other objectives / the dual-UNet model on the tiny DA configuration: 4-step DDIM (every step),
    one model_predictions call and 3 ancestral steps per variant; weights from founddiff_amd.synth."""
    import unittest.mock as um
    from founddiff_amd.synth import ct_phantom
    _, ld = ct_phantom(1, 64, seed=10)
    x_in = torch.from_numpy(ld)
    g = torch.Generator().manual_seed(78)
    noises = torch.randn(3, *x_in.shape, generator=g)
    torch.manual_seed(10)
    noise0 = torch.randn(x_in.shape)
    x_cond2 = torch.from_numpy(ct_phantom(1, 64, seed=11)[1])        # second condition plane (input_condition)
    arrs = {"x_input": x_in, "x_cond2": x_cond2, "noise0": noise0, "anc.noise": noises}
    spec_all = {}
    for name, v in VARIANTS.items():
        ic, mask = v.get("input_condition", False), v.get("mask", False)
        net = D.UnetRes(dim=32, dim_mults=(1, 2), num_unet=v["num_unet"], condition=True, objective=v["objective"],
                        test_res_or_noise=v["test"], input_condition=ic)
        dif = D.ResidualDiffusion(net, image_size=64, timesteps=1000, sampling_timesteps=4, objective=v["objective"],
                                  loss_type="l2", condition=True, sum_scale=0.01, test_res_or_noise=v["test"],
                                  input_condition=ic, input_condition_mask=mask).eval()
        spec = {k: s for k, s in synth.spec_of(dif.state_dict()).items() if k.startswith("model.")}
        dif.load_state_dict(synth.synth_state_dict(spec, SEED_W), strict=False)
        dif.init()
        # the 3-plane init_conv of the input_condition variants gets its own key namespace in the fixture
        ns = (lambda k: k.replace("model.", "model_ic.", 1)) if ic else (lambda k: k)
        spec_all.update({ns(k): s for k, s in spec.items() if ".clip_model." not in k or ".dose_encoder." in k})
        if ic:
            new = synth.synth_state_dict({ns(k): s for k, s in spec.items()}, SEED_W)
            dif.load_state_dict({k: new[ns(k)] for k in spec}, strict=False)
            dif.init()
        torch.manual_seed(10)
        xs_in = [x_in.clone(), x_cond2.clone()] if ic else [x_in.clone()]
        outs = dif.sample(xs_in, batch_size=1, last=False)
        arrs[name + ".ddim.imgs"] = torch.stack(outs, 0)
        xi = x_in * 2 - 1
        xt = xi + 0.1 * noise0
        tt = torch.full((1,), 979, dtype=torch.long)
        xc2 = (x_cond2 if mask else x_cond2 * 2 - 1) if ic else 0
        pr = dif.model_predictions(xi, xt, tt, xc2)
        arrs[name + ".mp.pred_res"], arrs[name + ".mp.pred_noise"], arrs[name + ".mp.x_start"] = \
            pr.pred_res, pr.pred_noise, pr.pred_x_start
        img, anc = xt.clone(), []
        for i, t in enumerate(range(999, 996, -1)):
            with um.patch.object(torch, "randn_like", lambda x, i=i: noises[i]):
                img, _ = dif.p_sample(xi, img, t, xc2)
            anc.append(img.clone())
        arrs[name + ".anc.imgs"] = torch.stack(anc, 0)
        print(name, "done")
    save("e2e_da_variants", spec=spec_all, **arrs)


def g_e2e_vanilla(D):
    """BASELINE configs[0] geometry on the vanilla path: Unet(dim=32, dim_mults=(1,2)), 64x64, 10-step DDIM for the
    three objectives + 6 ancestral steps."""
    from src import denoising_diffusion_pytorch as V
    net = V.Unet(32, dim_mults=(1, 2), channels=1)
    for obj, S in (("pred_noise", 10), ("pred_x0", 10), ("pred_v", 10)):
        dif = V.GaussianDiffusion(net, image_size=64, timesteps=1000, sampling_timesteps=S,
                                  objective=obj, beta_schedule="cosine").eval()
        sd = dif.state_dict()
        spec = {k: v for k, v in synth.spec_of(sd).items() if k.startswith("model.")}
        dif.load_state_dict(synth.synth_state_dict(spec, SEED_W), strict=False)  # schedule buffers kept
        if obj == "pred_noise":
            arrs = {}
            x = rnd(2, 1, 64, 64, seed=41)
            arrs["unet.x"], arrs["unet.t"] = x, torch.tensor([999, 17])
            arrs["unet.out"] = dif.model(x, torch.tensor([999, 17]))
        torch.manual_seed(5)
        xT = torch.randn(2, 1, 64, 64)
        torch.manual_seed(5)
        out = dif.sample(batch_size=2)
        arrs[f"ddim.{obj}.xT"], arrs[f"ddim.{obj}.out"] = xT, out[0]
    # ancestral, 6 steps
    dif = V.GaussianDiffusion(net, image_size=64, timesteps=1000, objective="pred_noise",
                              beta_schedule="linear").eval()
    dif.load_state_dict(synth.synth_state_dict(spec, SEED_W), strict=False)
    g = torch.Generator().manual_seed(78)
    noises = torch.randn(6, 2, 1, 64, 64, generator=g)
    img = xT.clone()
    import unittest.mock as um
    anc = []
    for i, t in enumerate(range(999, 993, -1)):
        with um.patch.object(torch, "randn_like", lambda x, i=i: noises[i]):
            img, _ = dif.p_sample(img, t)
        anc.append(img.clone())
    arrs["anc.noise"], arrs["anc.imgs"] = noises, torch.stack(anc, 0)
    save("e2e_vanilla_tiny", spec={k: v for k, v in spec.items()}, **arrs)


def g_full64(D):
    """Shipped architecture (dim 64, mults 1-2-4-8, real RN50-sized DA-CLIP) at 64x64: one forward
    and a 2-step DDIM sample."""
    from founddiff_amd.synth import ct_phantom
    dif, spec = _da_model(D, 64, (1, 2, 4, 8), 2, 64)
    _, ld = ct_phantom(1, 64, seed=10)
    x_in = torch.from_numpy(ld)
    torch.manual_seed(10)
    noise0 = torch.randn(x_in.shape)
    xi = x_in * 2 - 1
    xt = xi + 0.1 * noise0
    tt = torch.full((1,), 999, dtype=torch.long)
    arrs = {"x_input": x_in, "noise0": noise0}
    arrs["unet.out"] = dif.model(torch.cat((xt, xi), 1), [dif.alphas_cumsum[tt] * 1000, dif.betas_cumsum[tt] * 1000])[0]
    torch.manual_seed(10)
    arrs["ddim2.out"] = dif.sample([x_in.clone()], batch_size=1, last=True)[-1]
    save("full_arch_64", spec=spec, **arrs)


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    if what == "full64":
        D = _refimport.install(FULL_CLIP)
        g_full64(D)
    else:
        D = _refimport.install(TINY_CLIP)
        if what in ("schedule", "all"):
            g_schedule(D)
        if what in ("modules", "all"):
            g_modules(D)
            g_modules_vanilla(D)
        if what in ("modules_odd", "all"):
            g_modules_odd(D)
        if what in ("e2e_da", "all"):
            g_e2e_da(D)
        if what in ("e2e_variants", "all"):
            g_e2e_da_variants(D)
        if what in ("e2e_vanilla", "all"):
            g_e2e_vanilla(D)
