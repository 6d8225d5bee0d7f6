"""Pin the CPU oracle (oracle/) against golden vectors captured from the reference itself
(tests/golden/make_golden.py).  CPU only."""
import os

import pytest
import torch

from conftest import rel_err
from oracle import nets, sampler, schedule

TOL = 2e-5   # fp32 re-association noise between two CPU implementations


def test_residual_schedule(golden):
    g = golden("schedule")
    for tag, after in (("ctor", False), ("init", True)):
        sch = schedule.residual_schedule(1000, after_init=after)
        for k in schedule.RES_KEYS:
            ref = g[f"{tag}.{k}"]
            assert torch.equal(sch[k], ref), (tag, k, (sch[k] - ref).abs().max())


def test_ddim_pairs(golden):
    g = golden("schedule")
    for S in (2, 10, 25, 50):
        assert schedule.ddim_time_pairs(1000, S) == [tuple(p) for p in g[f"pairs.{S}"].tolist()]
    assert schedule.ddim_time_pairs(1000, 50)[0] == (999, 979)


def test_gaussian_schedule(golden):
    g = golden("schedule")
    for bs in ("linear", "cosine"):
        sch = schedule.gaussian_schedule(1000, bs)
        for k, v in sch.items():
            assert torch.equal(v, g[f"vanilla.{bs}.{k}"]), (bs, k)


def test_time_mlp(golden):
    g = golden("modules")
    sd = nets.SD(g.weights("time_mlp."), "time_mlp.")
    assert rel_err(nets.time_mlp(sd, g["time_mlp.in"], 32), g["time_mlp.out"]) < TOL


@pytest.mark.parametrize("name", ["rb_same", "rb_proj"])
def test_resnet_block(golden, name):
    g = golden("modules")
    sd = nets.SD(g.weights(name + "."), name + ".")
    assert rel_err(nets.block(sd.sub("block1."), g[name + ".in"]), g[name + ".block_out"]) < TOL
    assert rel_err(nets.da_resnet_block(sd, g[name + ".in"]), g[name + ".out"]) < TOL


def test_efficient_scan_merge(golden):
    g = golden("modules")
    for tag in ("even", "odd"):
        x = g[f"escan_{tag}.in"]
        xs = nets.efficient_scan(x)
        assert torch.equal(xs, g[f"escan_{tag}.out"])
        y = nets.efficient_merge(xs, x.shape[2], x.shape[3])
        assert torch.equal(y, g[f"emerge_{tag}.out"])
        assert torch.equal(y.view_as(x), x)      # merge inverts scan


def test_scan_c_vs_torch():
    torch.manual_seed(0)
    b, K, Dg, N, L = 2, 4, 6, 5, 37
    u, dl = torch.randn(b, K * Dg, L), torch.randn(b, K * Dg, L) * 0.5
    A = -torch.exp(torch.randn(K * Dg, N) * 0.3)
    B, C = torch.randn(b, K, N, L), torch.randn(b, K, N, L)
    D, bias = torch.randn(K * Dg), torch.randn(K * Dg) * 0.1
    y1 = nets.selective_scan(u, dl, A, B, C, D, bias, True)
    y2 = nets.selective_scan_torch(u, dl, A, B, C, D, bias, True)
    assert rel_err(y1, y2) < 1e-5


@pytest.mark.parametrize("tag", ["c32n4", "c64n32", "c32n16"])
def test_ss2d(golden, tag):
    g = golden("modules")
    p = f"ss2d_{tag}."
    sd = nets.SD(g.weights(p), p)
    assert rel_err(nets.cross_selective_scan(sd, g[p + "core_in"]), g[p + "core_out"]) < TOL
    assert rel_err(nets.ss2d(sd, g[p + "x"], g[p + "c"]), g[p + "out"]) < TOL


def test_transposed_attention(golden):
    g = golden("modules")
    sd = nets.SD(g.weights("tattn."), "tattn.")
    assert rel_err(nets.transposed_attention(sd, g["tattn.in"]), g["tattn.out"]) < TOL


@pytest.mark.parametrize("tag", ["c32", "c64"])
def test_mamba_block(golden, tag):
    g = golden("modules")
    p = f"mamba_{tag}."
    sd = nets.SD(g.weights(p), p)
    out = nets.mamba_block(sd, g[p + "x"], g[p + "c"], g[p + "t"])
    assert rel_err(out, g[p + "out"]) < TOL


@pytest.mark.parametrize("tag", ["c32n4", "c64n32", "c32n16"])
def test_ss2d_odd_sizes(golden, tag):
    """odd H / W: the reference's pad-to-even before the gather and crop after the merge (src/emamba2.py:191-199,
    253-260), SS2D outputs captured from the reference at 7x9, 5x6, 10x7."""
    g = golden("modules_odd")
    p = f"ss2d_{tag}."
    sd = nets.SD(g.weights(p), p)
    assert rel_err(nets.ss2d(sd, g[p + "x"], g[p + "c"]), g[p + "out"]) < TOL


@pytest.mark.parametrize("tag", ["c32", "c64"])
def test_mamba_block_odd_sizes(golden, tag):
    g = golden("modules_odd")
    p = f"mamba_{tag}."
    sd = nets.SD(g.weights(p), p)
    assert rel_err(nets.mamba_block(sd, g[p + "x"], g[p + "c"], g[p + "t"]), g[p + "out"]) < TOL


def test_samplers_conv(golden):
    g = golden("modules")
    import torch.nn.functional as F
    w = g.weights("down.")
    assert rel_err(F.conv2d(g["down.in"], w["down.weight"], w["down.bias"], stride=2, padding=1), g["down.out"]) < TOL
    w = g.weights("up.")
    up = F.interpolate(g["up.in"], scale_factor=2, mode="nearest")
    assert rel_err(F.conv2d(up, w["up.1.weight"], w["up.1.bias"], padding=1), g["up.out"]) < TOL


def test_dose_encoder(golden):
    g = golden("modules")
    sd = nets.SD(g.weights("iqa."), "iqa.")
    dose, ctx = nets.dose_encoder(sd, g["iqa.in"].repeat(1, 3, 1, 1))
    assert rel_err(dose, g["iqa.dose"]) < TOL
    assert rel_err(ctx, g["iqa.ctx"]) < TOL


def test_vanilla_modules(golden):
    g = golden("modules_vanilla")
    sd = nets.SD(g.weights("vrb."), "vrb.")
    assert rel_err(nets.v_resnet_block(sd, g["vrb.x"], g["vrb.t"]), g["vrb.out"]) < TOL
    sd = nets.SD(g.weights("vlin."), "vlin.fn.")
    assert rel_err(nets.v_linear_attention(sd, g["vlin.x"]), g["vlin.out"]) < TOL
    sd = nets.SD(g.weights("vatt."), "vatt.fn.")
    assert rel_err(nets.v_attention(sd, g["vatt.x"]), g["vatt.out"]) < TOL


def test_e2e_da_tiny(golden):
    g = golden("e2e_da_tiny")
    w = g.weights()
    orc = sampler.ResidualOracle(w, prefix="model.unet0.", sampling_timesteps=10)
    x_in = g["x_input"]
    xi = x_in * 2 - 1
    xt = xi + 0.1 * g["ddim.noise0"]
    tt = torch.full((2,), 979, dtype=torch.long)
    assert rel_err(orc.unet(xt, xi, tt), g["unet.out"]) < 1e-4
    pr, pn, xs = orc.model_predictions(xi, xt, tt)
    assert rel_err(pr, g["mp.pred_res"]) < 1e-4
    assert rel_err(pn, g["mp.pred_noise"]) < 1e-4
    assert rel_err(xs, g["mp.x_start"]) < 1e-4
    trace = {}
    out = orc.sample(x_in, g["ddim.noise0"], trace=trace)
    imgs = g["ddim.imgs"]
    assert rel_err(out[0], imgs[0]) < 1e-6
    for i, im in enumerate(trace["img"]):
        assert rel_err((im + 1) * 0.5, imgs[i + 1]) < 2e-4, i
    assert rel_err(out[1], g["ddim.out"]) < 2e-4
    # ancestral steps
    img = xt.clone()
    for i, t in enumerate(range(999, 979, -1)):
        img, _ = orc.p_sample(xi, img, t, g["anc.noise"][i])
        assert rel_err(img, g["anc.imgs"][i]) < 2e-4, t
    img0, xs0 = orc.p_sample(xi, xt, 0, None)
    assert rel_err(img0, g["anc.t0_img"]) < 1e-4
    assert rel_err(xs0, g["anc.t0_xstart"]) < 1e-4


VARIANTS = {   # name -> (num_unet, objective, test_res_or_noise); mirrors tests/golden/make_golden.py
    "pred_noise": (1, "pred_noise", "noise"), "res_noise": (2, "pred_res_noise", "res_noise"),
    "rn_noise": (2, "pred_res_noise", "noise"), "rn_res": (2, "pred_res_noise", "res"),
    "x0_noise": (2, "pred_x0_noise", "res_noise"), "incond": (1, "pred_res", "res"), "incond_mask": (1, "pred_res", "res"),
}


@pytest.mark.parametrize("name", sorted(VARIANTS))
def test_e2e_da_variants(golden, name):
    """The other objectives and the dual-UNet dispatch against the reference's own outputs."""
    g = golden("e2e_da_variants")
    nu, obj, tst = VARIANTS[name]
    ic, mask = name.startswith("incond"), name == "incond_mask"
    orc = sampler.ResidualOracle(g.weights("model_ic." if ic else "model."), sampling_timesteps=4, objective=obj,
                                 test_res_or_noise=tst, num_unet=nu, input_condition=ic, input_condition_mask=mask,
                                 prefix="model_ic.unet0." if ic else "model.unet0.")
    x_in = g["x_input"]
    if ic:
        orc.x_cond2 = g["x_cond2"] if mask else g["x_cond2"] * 2 - 1
    xi = x_in * 2 - 1
    xt = xi + 0.1 * g["noise0"]
    tt = torch.full((1,), 979, dtype=torch.long)
    pr, pn, xs = orc.model_predictions(xi, xt, tt)
    # x_start from a noise prediction divides by one_minus_alphas_cumsum[t] (6.5e-3 at t=979): a 1e-6
    # difference in the UNet output is amplified ~150x, hence the looser gate for those two variants
    tol, tol_seq = (1e-3, 1e-3) if name in ("pred_noise", "rn_noise") else (1e-4, 2e-4)
    assert rel_err(pr, g[name + ".mp.pred_res"]) < tol
    assert rel_err(pn, g[name + ".mp.pred_noise"]) < 1e-4
    assert rel_err(xs, g[name + ".mp.x_start"]) < tol
    trace = {}
    out = orc.sample(x_in, g["noise0"], trace=trace, x_cond2_01=g["x_cond2"] if ic else None)
    imgs = g[name + ".ddim.imgs"]
    assert rel_err(out[0], imgs[0]) < 1e-6
    for i, im in enumerate(trace["img"]):
        assert rel_err((im + 1) * 0.5, imgs[i + 1]) < tol_seq, i
    img = xt.clone()
    for i, t in enumerate(range(999, 996, -1)):
        img, _ = orc.p_sample(xi, img, t, g["anc.noise"][i])
        assert rel_err(img, g[name + ".anc.imgs"][i]) < tol_seq, t


def test_e2e_vanilla_tiny(golden):
    g = golden("e2e_vanilla_tiny")
    w = g.weights()
    sd = nets.SD(w, "model.")
    assert rel_err(nets.vanilla_unet(sd, g["unet.x"], g["unet.t"]), g["unet.out"]) < 1e-4
    for obj in ("pred_noise", "pred_x0", "pred_v"):
        orc = sampler.GaussianOracle(w, sampling_timesteps=10, objective=obj)
        out = orc.sample(g[f"ddim.{obj}.xT"])
        assert rel_err(out[0], g[f"ddim.{obj}.out"]) < 5e-4, obj
    orc = sampler.GaussianOracle(w, beta_schedule="linear")
    img = g["ddim.pred_v.xT"]
    for i, t in enumerate(range(999, 993, -1)):
        img, _ = orc.p_sample(img, t, g["anc.noise"][i])
        assert rel_err(img, g["anc.imgs"][i]) < 2e-4, t


def test_full_arch_64(golden):
    """Shipped architecture (dim 64, mults 1-2-4-8, RN50-sized DA-CLIP) at 64x64."""
    g = golden("full_arch_64")
    w = g.weights("model.")
    orc = sampler.ResidualOracle(w, prefix="model.unet0.", sampling_timesteps=2)
    xi = g["x_input"] * 2 - 1
    xt = xi + 0.1 * g["noise0"]
    tt = torch.full((1,), 999, dtype=torch.long)
    assert rel_err(orc.unet(xt, xi, tt), g["unet.out"]) < 1e-4
    out = orc.sample(g["x_input"], g["noise0"])
    assert rel_err(out[-1], g["ddim2.out"]) < 2e-4


def test_metrics_oracle_vs_scipy():
    """SSIM restatement (kornia is absent) cross-checked against scipy's reflect-mode correlation."""
    import numpy as np
    from scipy import ndimage
    from founddiff_amd import synth
    from oracle import metrics as om
    nd, ld = synth.ct_phantom(1, 48, seed=1)
    a, b = torch.from_numpy(ld), torch.from_numpy(nd)
    k = om.gaussian_window().numpy().astype(np.float64)
    f = lambda x: ndimage.correlate(x.astype(np.float64), k, mode="mirror")
    x, y = ld[0, 0], nd[0, 0]
    mu1, mu2 = f(x), f(y)
    s1, s2, s12 = f(x * x) - mu1 ** 2, f(y * y) - mu2 ** 2, f(x * y) - mu1 * mu2
    m = ((2 * mu1 * mu2 + 1e-4) * (2 * s12 + 9e-4)) / ((mu1 ** 2 + mu2 ** 2 + 1e-4) * (s1 + s2 + 9e-4))
    assert abs(float(om.ssim(a, b)) - float(np.clip(m, 0, 1).mean())) < 1e-5
    assert abs(float(om.psnr(a, b)) - 10 * np.log10(1.0 / np.mean((x - y) ** 2))) < 1e-4


@pytest.mark.skipif(not os.path.isdir("/root/reference/src"), reason="the reference only exists in the build container")
def test_fixtures_regenerate_from_the_reference(tmp_path):
    """Where /root/reference is present, three fixtures are regenerated from it (tests/golden/make_golden.py: schedule,
    modules_odd, e2e_da) into a temp dir and compared with the committed files BYTE FOR BYTE (sorted keys, sorted-key
    spec JSON, fixed zip timestamps: make_golden.save) and array by array: the goldens are reproducible outputs of the
    reference, not hand-kept data (spec_json drift; VERDICT r3: key order of e2e_da_tiny's spec)."""
    import subprocess
    import sys
    import numpy as np
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    env = dict(os.environ, FD_GOLDEN_OUT=str(tmp_path), PYTHONDONTWRITEBYTECODE="1")
    for what, fname in (("schedule", "schedule"), ("modules_odd", "modules_odd"), ("e2e_da", "e2e_da_tiny")):
        r = subprocess.run([sys.executable, os.path.join(here, "make_golden.py"), what], env=env, capture_output=True, text=True,
                           timeout=900)
        assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
        new, old = np.load(tmp_path / f"{fname}.npz"), np.load(os.path.join(here, f"{fname}.npz"))
        assert sorted(new.files) == sorted(old.files)
        for k in old.files:
            assert np.array_equal(new[k], old[k]), (what, k)
        assert open(tmp_path / f"{fname}.npz", "rb").read() == open(os.path.join(here, f"{fname}.npz"), "rb").read(), \
            f"{fname}.npz: arrays equal but the file differs (key order / spec JSON / zip metadata)"
