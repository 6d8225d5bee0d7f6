"""Round-5 GPU tests (through the drop-in Python API / the C ABI):

* the whole sampling loop on the `fp32s` engine (fp32 storage, split-bf16 contractions: 3 bf16 MFMAs per product)
  held to the SAME 1e-3 gate the exact-f32 parity engine meets -- against the reference's goldens (config 1 at every
  DDIM step, the shipped architecture at 64x64) and against the CPU oracle at the bench size
  (src/DADiff.py:1276-1365 is the loop whose output the north star's tolerance applies to);
* the production mode (bf16 kernels + precision tail) against the CPU ORACLE over a loop, not only against this
  library's own fp32 engine: BASELINE configs[1] geometry, 10-step DDIM.
"""
import os

import pytest
import torch
import torch.nn.functional as F

from conftest import GOLDEN, rel_err
from test_gpu_e2e import TINY_CLIP, _tiny_model, l2rel, psnr

pytestmark = pytest.mark.gpu


def test_ddim_tiny_fp32s(golden):
    """config 1 (64x64, 10-step DDIM, DA path) with precision='fp32s': every step's image within 1e-3 of the reference."""
    g, dif = _tiny_model(golden, "fp32s")
    assert dif._eng().mode == "fp32s" and dif._eng().f32_split == 1
    imgs = dif.sample([g["x_input"].cuda()], batch_size=2, last=False, noise=g["ddim.noise0"].cuda())
    ref = g["ddim.imgs"]
    assert len(imgs) == ref.shape[0]
    for i, im in enumerate(imgs):
        assert rel_err(im.cpu(), ref[i]) < 1e-3, i
    out = dif.sample([g["x_input"].cuda()], batch_size=2, last=True, noise=g["ddim.noise0"].cuda())
    assert rel_err(out[-1].cpu(), g["ddim.out"]) < 1e-3
    assert torch.equal(out[-1], imgs[-1])          # whole-loop graph == per-step graphs


def test_full_arch_64_fp32s(golden):
    """The shipped architecture at 64x64, one forward + 2-step DDIM, precision='fp32s' vs the reference: 1e-3."""
    if not os.path.exists(os.path.join(GOLDEN, "full_arch_64.npz")):
        pytest.skip("full_arch_64.npz not generated")
    from founddiff_amd.DADiff import ResidualDiffusion, UnetRes
    g = golden("full_arch_64")
    net = UnetRes(dim=64, dim_mults=(1, 2, 4, 8), num_unet=1, condition=True, objective="pred_res",
                  test_res_or_noise="res", precision="fp32s")
    dif = ResidualDiffusion(net, image_size=64, timesteps=1000, sampling_timesteps=2, objective="pred_res",
                            loss_type="l2", condition=True, sum_scale=0.01, test_res_or_noise="res")
    missing, unexpected = dif.load_state_dict(g.weights("model."), strict=False)
    assert not [k for k in missing if k.startswith("model.")] and not unexpected
    dif = dif.to("cuda")
    dif.init()
    xi = (g["x_input"] * 2 - 1).cuda()
    xt = xi + 0.1 * g["noise0"].cuda()
    tt = torch.full((1,), 999, dtype=torch.long, device="cuda")
    out = dif.model(torch.cat((xt, xi), 1), [dif.alphas_cumsum[tt] * 1000, dif.betas_cumsum[tt] * 1000])[0]
    assert rel_err(out.cpu(), g["unet.out"]) < 1e-3
    res = dif.sample([g["x_input"].cuda()], batch_size=1, last=True, noise=g["noise0"].cuda())
    assert rel_err(res[-1].cpu(), g["ddim2.out"]) < 1e-3


def test_vs_oracle_512_one_forward_fp32s():
    """BASELINE configs[2] geometry (512x512, full architecture), ONE model_predictions call on the fp32s engine against
    the CPU oracle: the 1e-3 parity gate at the bench size (same inputs as test_gpu_e2e.test_vs_oracle_512_one_forward)."""
    from founddiff_amd import arch, synth
    from founddiff_amd.DADiff import ResidualDiffusion, UnetRes, load_weights
    from oracle import sampler
    spec = arch.da_unet_spec(64, (1, 2, 4, 8), prefix="model.unet0.")
    w = synth.synth_state_dict(spec, seed=0)
    _, ld = synth.ct_phantom(1, 512, seed=10)
    x_in = torch.from_numpy(ld) * 2 - 1
    x_t = x_in + 0.1 * torch.randn(1, 1, 512, 512, generator=torch.Generator().manual_seed(4))
    tt = torch.full((1,), 500, dtype=torch.long)
    torch.set_num_threads(min(32, torch.get_num_threads()))
    ref = sampler.ResidualOracle(w, prefix="model.unet0.", sampling_timesteps=50).model_predictions(x_in, x_t, tt)
    net = UnetRes(dim=64, dim_mults=(1, 2, 4, 8), num_unet=1, condition=True, objective="pred_res",
                  test_res_or_noise="res", precision="fp32s")
    dif = ResidualDiffusion(net, image_size=512, timesteps=1000, sampling_timesteps=50, objective="pred_res",
                            loss_type="l2", condition=True, sum_scale=0.01, test_res_or_noise="res")
    load_weights(dif, w)
    dif = dif.to("cuda")
    dif.init()
    p = dif.model_predictions(x_in.cuda(), x_t.cuda(), tt.cuda())
    assert rel_err(p.pred_res.cpu(), ref[0]) < 1e-3
    assert rel_err(p.pred_x_start.cpu(), ref[2]) < 1e-3


def test_config2_256_production_vs_oracle_50step():
    """BASELINE configs[1] (256x256, full architecture + DA-CLIP, 50-step DDIM): the PRODUCTION mode (bf16 kernels, levels 0-1
    of the last step on the fp32s engine) and the fp32s whole-loop mode against oracle.sampler.ResidualOracle.sample on the
    same x_T -- the CPU restatement of the reference loop (src/DADiff.py:1276-1365), not this library's own fp32 engine.
    Production: L2 <= 1e-2 and >= 45 dB (the drift gate of SURVEY section 7); fp32s: max-rel <= 1e-3 (the north star's
    tolerance, over the whole loop).  Measured in round 5: production 7.3e-3-class (engine-vs-engine figure of
    test_config2_256_bf16_50step_drift), and at 10 steps 1.37e-2 / 47.3 dB (fewer, larger steps weigh the bf16 steps more);
    fp32s 6e-5.  ~50 CPU forwards at 256x256: about a minute on 32 host threads."""
    from conftest import oracle_ddim_loop
    from founddiff_amd.DADiff import ResidualDiffusion, UnetRes, load_weights
    S = 50
    w, x_in, noise, ref = oracle_ddim_loop(256, S, noise_seed=7)          # (shared with tests/test_gpu_fp16.py)
    outs = {}
    for prec in ("bf16", "fp32s"):
        net = UnetRes(dim=64, dim_mults=(1, 2, 4, 8), num_unet=1, condition=True, objective="pred_res",
                      test_res_or_noise="res", precision=prec)
        dif = ResidualDiffusion(net, image_size=256, timesteps=1000, sampling_timesteps=S, objective="pred_res",
                                loss_type="l2", condition=True, sum_scale=0.01, test_res_or_noise="res")
        load_weights(dif, w)
        dif = dif.to("cuda")
        dif.init()
        if prec == "bf16":
            assert dif.final_fp32_steps == 1 and dif.final_outer_levels == 2      # the benchmarked configuration
        outs[prec] = dif.sample([x_in.cuda()], batch_size=1, noise=noise.cuda())[-1].float().cpu()
        del dif, net
        torch.cuda.empty_cache()
    e, db = l2rel(outs["bf16"], ref), psnr(outs["bf16"], ref)
    print(f"production vs oracle, 256x256 / {S} steps: L2 {e:.3e}, {db:.1f} dB; fp32s max-rel {rel_err(outs['fp32s'], ref):.2e}")
    assert e < 1e-2 and db > 45.0, (e, db)
    assert rel_err(outs["fp32s"], ref) < 1e-3


def test_ln_rows_with_common_offset():
    """LayerNorm rows whose common offset dwarfs their spread: ln_rows_kernel (every mode; centred two-pass
    statistics, applies (x - mean) * rstd) on fp32 rows x = 50 + 0.1 randn, and the one-pass packed statistics of the fused bf16
    kernels (fd_ln_mod_chunk through fd_pw_dw3x3's LayerNorm prologue is covered by its own tests; here the row kernel in bf16)
    on rows x = 8 + 0.25 randn -- the largest |mean| / std a bf16 row can carry with its spread still resolved."""
    import ctypes as C
    from founddiff_amd import _lib as L
    torch.manual_seed(3)
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for dt, tdt, Cc, off, sd, tol in ((L.FD_F32, torch.float32, 256, 50.0, 0.1, 5e-4), (L.FD_F32, torch.float32, 64, -300.0, 0.05, 2e-3),
                                      (L.FD_BF16, torch.bfloat16, 128, 8.0, 0.25, 2e-2)):
        B, hw = 2, 512
        x = (off + sd * torch.randn(B, hw, Cc)).to(tdt)
        g, b = torch.randn(Cc), torch.randn(Cc)
        shift, scale = torch.randn(B, Cc) * 0.2, torch.randn(B, Cc) * 0.2
        ref = F.layer_norm(x.float().double(), (Cc,), g.double(), b.double(), 1e-5) * (1 + scale.double()[:, None]) + shift.double()[:, None]
        xd, out = x.cuda(), torch.empty(B, hw, Cc, device="cuda", dtype=tdt)
        gd, bd, shd, scd = g.cuda(), b.cuda(), shift.cuda().contiguous(), scale.cuda().contiguous()
        L.call("fd_ln_modulate", dt, C.c_void_p(xd.data_ptr()), C.c_void_p(gd.data_ptr()), C.c_void_p(bd.data_ptr()), 1e-5,
               C.c_void_p(shd.data_ptr()), C.c_void_p(scd.data_ptr()), Cc, C.c_void_p(out.data_ptr()), B, hw, Cc, s)
        torch.cuda.synchronize()
        err = float((out.float().cpu().double() - ref).abs().max() / ref.abs().max())
        print(f"ln rows offset {off} sd {sd} {tdt}: max-rel {err:.2e}")
        assert err < tol, (off, sd, tdt, err)


@pytest.mark.parametrize("cfg", [dict(cin=128, cout=64, hw=(32, 32)),       # <64, 16, up>: u2s of the shipped architecture
                                 dict(cin=128, cout=64, hw=(40, 48)),       # <64, 8, up>: source rows not a multiple of 16
                                 dict(cin=256, cout=128, hw=(32, 48)),      # <128, 8, up>: u1s
                                 dict(cin=192, cout=200, hw=(32, 48)),      # three slabs, a ragged channel tile
                                 dict(cin=512, cout=256, hw=(64, 64))])     # u0s at its real size: 8 slabs x 4 classes per workgroup
def test_conv3x3_upsample_as_four_2x2(cfg):
    """nn.Upsample(scale_factor=2, nearest) -> Conv2d 3x3 (src/DADiff.py:121-127) as four 2x2 convolutions on the source grid
    (conv3x3_halo_kernel<..., UP>, kernel id 14, fd_conv_params.weight_up2x): against torch on the up-sampled image with the
    fp32 weights the sub-pixel matrix was summed from, against the 9-tap halo kernel on the same operands, borders included,
    and bit for bit repeatable (the weight ring's slot rotation and counted waits are manual)."""
    import ctypes as C
    from founddiff_amd import _lib as L
    from founddiff_amd.engine import ConvW
    from test_gpu_e2e import bare_engine, nhwc, nchw
    e = bare_engine("bf16")
    torch.manual_seed(31)
    B, (H, W), cin, cout = 2, cfg["hw"], cfg["cin"], cfg["cout"]
    x = torch.randn(B, cin, H, W).to(torch.bfloat16).float()
    w = torch.randn(cout, cin, 3, 3) / (3 * cin ** 0.5)
    bias = torch.randn(cout)
    ref = F.conv2d(F.interpolate(x, scale_factor=2, mode="nearest"), w, bias, padding=1)
    cw9 = ConvW(w, bias, e.dev, e.tdt)
    cw4 = ConvW(w, bias, e.dev, e.tdt, up2x=True)
    assert cw9.w_up is None and tuple(cw4.w_up.shape) == (cout, 16 * cin)
    xa = nhwc(x, e.tdt)
    o9 = torch.empty(B, 2 * H, 2 * W, cout, device="cuda", dtype=e.tdt)
    o4 = torch.full_like(o9, float("nan"))
    assert e.conv(cw9, xa, B, H, W, o9, upsample=True, probe="kid") == 11
    assert e.conv(cw4, xa, B, H, W, o4, upsample=True, probe="kid") == 14
    e.conv(cw9, xa, B, H, W, o9, upsample=True)
    e.conv(cw4, xa, B, H, W, o4, upsample=True)
    torch.cuda.synchronize()
    e9, e4 = rel_err(nchw(o9), ref), rel_err(nchw(o4), ref)
    print(f"up-sampling conv {cin} -> {cout} @ {H}x{W} -> {2 * H}x{2 * W}: 9-tap max-rel {e9:.2e}, 4 x 2x2 {e4:.2e}")
    assert torch.isfinite(o4.float()).all()
    assert e4 < 1.2e-2 and e4 < 1.5 * e9 + 1e-3                  # the tolerance of test_conv3x3_halo; no worse than the 9-tap form
    first = o4.clone()
    for _ in range(8):
        o4.zero_()
        e.conv(cw4, xa, B, H, W, o4, upsample=True)
        torch.cuda.synchronize()
        assert torch.equal(o4, first)
    # the class-parallel grid of the one-slice kernel set (`upsample` = 2: a workgroup per (tile, parity class)): the same bits
    e.low_latency = True
    o1 = torch.full_like(o4, float("nan"))
    e.conv(cw4, xa, B, H, W, o1, upsample=True)
    torch.cuda.synchronize()
    e.low_latency = False
    assert torch.equal(o1, first)


@pytest.mark.parametrize("cfg", [dict(c0=64, c1=0, cout=64, hw=(64, 64), up=False),          # <64, 8, SPL> (the split form launches 8-row tiles at every width: no SPL TH = 16 instantiation exists)
                                 dict(c0=64, c1=64, cout=64, hw=(72, 64), up=False),         # two sources, <64, 8, SPL>
                                 dict(c0=128, c1=64, cout=128, hw=(64, 64), up=False),       # 192 -> 128: <128, 8, SPL>, three slabs
                                 dict(c0=128, c1=0, cout=64, hw=(64, 64), up=True),          # through the up-sampling index map
                                 dict(c0=64, c1=0, cout=200, hw=(64, 64), up=False)])        # a ragged channel tile
def test_conv3x3_split_bf16_halo(cfg):
    """The fp32s engine's 3x3 convolutions on the halo-tiled kernel (conv3x3_halo_kernel<..., SPL>, kernel id 15: fp32 storage,
    x_hi.w_hi + x_hi.w_lo + x_lo.w_hi on the bf16 MFMA, fp32 out): against torch in fp64 at the split contraction's own
    accuracy, against the generic split implicit GEMM (the same weights without the pre-split copies), with the GroupNorm
    partial sums, and bit for bit repeatable."""
    from founddiff_amd import _lib as L
    from founddiff_amd.engine import ConvW
    from test_gpu_e2e import bare_engine, nhwc, nchw
    e = bare_engine("fp32s")
    e.f32_split = 1
    torch.manual_seed(41)
    B, (OH, OW), up = 2, cfg["hw"], cfg["up"]
    H, W = (OH // 2, OW // 2) if up else (OH, OW)
    cin, cout = cfg["c0"] + cfg["c1"], cfg["cout"]
    x = torch.randn(B, cin, H, W)
    w = torch.randn(cout, cin, 3, 3) / (3 * cin ** 0.5)
    bias = torch.randn(cout)
    xin = F.interpolate(x, scale_factor=2, mode="nearest") if up else x
    ref = F.conv2d(xin.double(), w.double(), bias.double(), padding=1)
    cws = ConvW(w, bias, e.dev, e.tdt, split=True)
    cwg = ConvW(w, bias, e.dev, e.tdt)
    assert cws.w_hi is not None and cwg.w_hi is None
    xa = nhwc(x[:, :cfg["c0"]], e.tdt)
    kw = dict(c0=cfg["c0"], upsample=up)
    if cfg["c1"]:
        kw.update(in1=nhwc(x[:, cfg["c0"]:], e.tdt), c1=cfg["c1"])
    mt = L.lib().fd_conv_mtiles(OH, OW)
    outs, parts = {}, {}
    for tag, cw, kid in (("halo", cws, 15), ("generic", cwg, None)):
        out = torch.full((B, OH, OW, cout), float("nan"), device="cuda")
        part = torch.full((B, mt, cout, 2), 7.0, device="cuda")
        got = e.conv(cw, xa, B, H, W, out, probe="kid", stats=part, **kw)
        assert (got == 15) if kid else (got != 15), (tag, got)
        e.conv(cw, xa, B, H, W, out, stats=part, **kw)
        torch.cuda.synchronize()
        outs[tag], parts[tag] = out, part
    eh, eg = rel_err(nchw(outs["halo"]).double(), ref), rel_err(nchw(outs["generic"]).double(), ref)
    print(f"split-bf16 3x3 {cin} -> {cout} @ {OH}x{OW} up={int(up)}: halo-tiled max-rel {eh:.2e}, generic {eg:.2e}")
    assert torch.isfinite(outs["halo"]).all()
    assert eh < 2e-5 and eh < 3 * eg + 1e-6
    s = parts["halo"].sum(1).cpu().double()
    assert rel_err(s[..., 0], ref.sum((2, 3))) < 1e-4
    assert rel_err(s[..., 1], (ref ** 2).sum((2, 3))) < 1e-4
    first = outs["halo"].clone()
    for _ in range(6):
        outs["halo"].zero_()
        e.conv(cws, xa, B, H, W, outs["halo"], stats=parts["halo"], **kw)
        torch.cuda.synchronize()
        assert torch.equal(outs["halo"], first)


@pytest.mark.parametrize("cfg", [dict(cin=128, cout=64, hw=(64, 64)), dict(cin=256, cout=128, hw=(64, 96)),
                                 dict(cin=128, cout=200, hw=(64, 64)), dict(cin=64, cout=64, hw=(128, 128))])
def test_conv3x3_upsample_four_2x2_split_bf16(cfg):
    """The fp32s engine's Upsample convolutions (src/DADiff.py:121-127) as four 2x2 SPLIT-bf16 convolutions on the source grid
    (conv3x3_halo_kernel<..., UP, SPL>, kernel id 16: the sub-pixel matrix of the fp32 weights pre-split into bf16 halves,
    fd_conv_params.weight_up2x_split_hi / _lo): against torch in fp64 on the up-sampled image at the split contraction's accuracy,
    against the 9-tap split form through the up-sampling index map, with the class-parallel grid (`upsample` = 2) bit for bit
    the same, and repeatable."""
    from founddiff_amd.engine import ConvW
    from test_gpu_e2e import bare_engine, nhwc, nchw
    e = bare_engine("fp32s")
    e.f32_split = 1
    torch.manual_seed(43)
    B, (OH, OW) = 2, cfg["hw"]
    H, W, cin, cout = OH // 2, OW // 2, cfg["cin"], cfg["cout"]
    x = torch.randn(B, cin, H, W)
    w = torch.randn(cout, cin, 3, 3) / (3 * cin ** 0.5)
    bias = torch.randn(cout)
    ref = F.conv2d(F.interpolate(x, scale_factor=2, mode="nearest").double(), w.double(), bias.double(), padding=1)
    cwu = ConvW(w, bias, e.dev, e.tdt, split=True, up2x=True)
    cw9 = ConvW(w, bias, e.dev, e.tdt, split=True)
    assert cwu.w_up_hi is not None and cwu.w_up_hi.shape == (cout, 16 * cin) and cw9.w_up_hi is None
    xa = nhwc(x, e.tdt)
    outs = {}
    for tag, cw, kid in (("four2x2", cwu, 16), ("ninetap", cw9, 15)):
        out = torch.full((B, OH, OW, cout), float("nan"), device="cuda")
        assert e.conv(cw, xa, B, H, W, out, probe="kid", c0=cin, upsample=True) == kid, tag
        e.conv(cw, xa, B, H, W, out, c0=cin, upsample=True)
        torch.cuda.synchronize()
        outs[tag] = out
    e4, e9 = rel_err(nchw(outs["four2x2"]).double(), ref), rel_err(nchw(outs["ninetap"]).double(), ref)
    print(f"split-bf16 up-sampling 3x3 {cin} -> {cout} to {OH}x{OW}: four 2x2 max-rel {e4:.2e}, nine taps {e9:.2e}")
    assert torch.isfinite(outs["four2x2"]).all()
    assert e4 < 2e-5 and e4 < 3 * e9 + 1e-6
    first = outs["four2x2"].clone()
    e.low_latency = True                       # one workgroup per (tile, parity class): the same bits
    try:
        par = torch.full_like(first, float("nan"))
        e.conv(cwu, xa, B, H, W, par, c0=cin, upsample=True)
        torch.cuda.synchronize()
        assert torch.equal(par, first)
    finally:
        e.low_latency = False
    for _ in range(4):
        outs["four2x2"].zero_()
        e.conv(cwu, xa, B, H, W, outs["four2x2"], c0=cin, upsample=True)
        torch.cuda.synchronize()
        assert torch.equal(outs["four2x2"], first)
