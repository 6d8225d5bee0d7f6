"""GPU tests of precision='fp16' (round 6): the 16-bit engine -- same kernels, same dataflow, same bytes as 'bf16' -- on the library's
binary16 build (csrc/fd_common.h: FD_HALF_F16; founddiff_amd/lib/libfounddiff_hip_f16.so), with the last step of a loop on the
fp32-storage engine of the default build.

* against the REFERENCE's goldens: config 1 at every DDIM step, the shipped architecture at 64x64 (one forward + 2-step DDIM), the
  Mamba blocks incl. the odd-size scans;
* against the CPU oracle over whole loops: BASELINE configs[1] (256x256) and configs[2] (512x512), 50-step DDIM -- L2 below the
  1e-3 the north star names (measured 6.9e-4 / 74.9 dB at 256x256; the bf16 production mode: 9.2e-3 / 52.4 dB);
* the two-stream sample() at the benchmarked shape is bitwise repeatable; odd sizes; the one-slice kernel set;
* a checkpoint whose activations leave binary16's range is REPORTED (FoundDiffHipError), not returned as NaNs.

The kernel-level tests of the binary16 build are tests/test_gpu_kernels.py's, which run once per build.
"""
import os

import pytest
import torch

from conftest import GOLDEN, rel_err
from test_gpu_e2e import TINY_CLIP, _tiny_model, bare_engine, l2rel, nchw, nhwc, psnr

pytestmark = pytest.mark.gpu


def test_engine_binds_the_binary16_build(golden):
    from founddiff_amd import _lib as L
    g, dif = _tiny_model(golden, "fp16")
    e = dif._eng()
    assert e.mode == "fp16" and e.tdt == torch.float16 and e.hip is L.F16
    assert L.F16.lib().fd_half_format() == 1 and L.BF16.lib().fd_half_format() == 0
    assert L.F16.lib().fd_version() == L.BF16.lib().fd_version()
    assert dif.final_fp32_steps == 1 and dif.final_outer_levels == 0        # the mode's default: the whole last step on the tail engine
    dif.sample([g["x_input"].cuda()], batch_size=2, noise=g["ddim.noise0"].cuda())
    e32 = dif.model.unet0._engine[("fp32s", 0)]
    assert e32.mode == "fp32s" and e32.hip is L.BF16                          # the tail engine: fp32 storage, split-bf16, default build
    # precision='auto' on a model whose activations stay in range IS the fp16 mode
    out16 = dif.sample([g["x_input"].cuda()], batch_size=2, noise=g["ddim.noise0"].cuda())[-1]
    g2, auto = _tiny_model(golden, "auto")
    assert auto.model.unet0.kernel_precision == "fp16" and auto.final_outer_levels == 0
    assert torch.equal(auto.sample([g["x_input"].cuda()], batch_size=2, noise=g["ddim.noise0"].cuda())[-1], out16)
    assert auto.model.unet0.kernel_precision == "fp16"


def test_ddim_tiny_fp16_vs_reference(golden):
    """config 1 (64x64, 10-step DDIM, DA path) with precision='fp16' against the REFERENCE's own images: every step within 5e-3
    max-rel (bf16 production: 1e-2 L2 gate only), the returned image within 2e-3; whole-loop graph == per-step graphs."""
    g, dif = _tiny_model(golden, "fp16")
    imgs = dif.sample([g["x_input"].cuda()], batch_size=2, last=False, noise=g["ddim.noise0"].cuda())
    ref = g["ddim.imgs"]
    assert len(imgs) == ref.shape[0]
    errs = [rel_err(im.cpu(), ref[i]) for i, im in enumerate(imgs)]
    print("fp16, config 1, max-rel per step:", " ".join(f"{e:.1e}" for e in errs))
    assert max(errs) < 5e-3, errs
    out = dif.sample([g["x_input"].cuda()], batch_size=2, last=True, noise=g["ddim.noise0"].cuda())
    assert torch.equal(out[-1], imgs[-1])
    e, db = l2rel(out[-1].cpu(), g["ddim.out"]), psnr(out[-1].cpu(), g["ddim.out"])
    print(f"fp16, config 1, returned image: L2 {e:.2e}, {db:.1f} dB")
    assert e < 1e-3 and db > 65.0, (e, db)


def test_full_arch_64_fp16(golden):
    """The shipped architecture at 64x64 against the reference: one raw forward at t = 999 (no tail: the 16-bit engine alone) and
    the 2-step DDIM.  The bf16 engine's gates here are 3e-2 / 2e-2."""
    if not os.path.exists(os.path.join(GOLDEN, "full_arch_64.npz")):
        pytest.skip("full_arch_64.npz not generated")
    from founddiff_amd.DADiff import ResidualDiffusion, UnetRes
    g = golden("full_arch_64")
    net = UnetRes(dim=64, dim_mults=(1, 2, 4, 8), num_unet=1, condition=True, objective="pred_res",
                  test_res_or_noise="res", precision="fp16")
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
    e1 = l2rel(out.cpu(), g["unet.out"])
    res = dif.sample([g["x_input"].cuda()], batch_size=1, last=True, noise=g["noise0"].cuda())
    e2 = rel_err(res[-1].cpu(), g["ddim2.out"])
    print(f"fp16, full architecture 64x64: raw forward L2 {e1:.2e}, 2-step DDIM max-rel {e2:.2e}")
    assert e1 < 6e-3 and e2 < 3e-3, (e1, e2)


@pytest.mark.parametrize("tag", ["c32", "c64", "odd:c32", "odd:c64"])
def test_mamba_block_fp16(golden, tag):
    """Mamba_block on the binary16 build against the reference's outputs (test_gpu_e2e.test_mamba_block's cases: fused 64-channel
    kernels, the generic path, the odd-size scans); gate 4e-3 max-rel where bf16's measured errors are 1.2e-2 .. 2.2e-2."""
    from founddiff_amd import _lib as L
    from founddiff_amd.engine import _Sub
    g = golden("modules_odd" if tag.startswith("odd:") else "modules")
    tag = tag.split(":")[-1]
    p = f"mamba_{tag}."
    e = bare_engine("fp16")
    m = e._pack_mamba(_Sub(g.weights(p), p))
    m["mod_off"], m["loc_off"] = 0, 0
    e.mod_total, e.loc_total = 6 * m["C"], m["D"]
    x, c, t = g[p + "x"], g[p + "c"], g[p + "t"]
    B, Cc, H, W = x.shape
    e.mod_all = torch.empty(B, 6 * Cc, device="cuda")
    e.linear(t.cuda(), m.pop("adaln_w").cuda(), m.pop("adaln_b").cuda(), e.mod_all, pre_silu=True)
    e.local_all = torch.empty(B, m["D"], device="cuda")
    e.linear(c.reshape(B, 256).cuda(), m.pop("local_w").cuda(), None, e.local_all, L.ACT_SILU)
    out = e.mamba_block(m, nhwc(x, e.tdt), B, H, W, "t")
    torch.cuda.synchronize()
    err = rel_err(nchw(out), g[p + "out"])
    print(f"mamba_block[{tag}] fp16: max-rel {err:.2e}")
    assert err < 4e-3, err


def _fp16_model(w, size, S, **kw):
    from founddiff_amd.DADiff import ResidualDiffusion, UnetRes, load_weights
    net = UnetRes(dim=64, dim_mults=(1, 2, 4, 8), num_unet=1, condition=True, objective="pred_res", test_res_or_noise="res",
                  precision="fp16")
    dif = ResidualDiffusion(net, image_size=size, timesteps=1000, sampling_timesteps=S, objective="pred_res", loss_type="l2",
                            condition=True, sum_scale=0.01, test_res_or_noise="res", **kw)
    load_weights(dif, w)
    dif = dif.to("cuda")
    dif.init()
    return dif


@pytest.mark.parametrize("size", [256, 512])
def test_fp16_50step_vs_oracle(size):
    """BASELINE configs[1] / configs[2] geometry, 50-step DDIM, precision='fp16' in its default configuration against
    oracle.sampler.ResidualOracle.sample on the same x_T: L2 <= 1e-3 -- the tolerance of the north star, over the loop -- and
    >= 70 dB; without the tail (the 16-bit engine alone for all 50 steps) <= 2.5e-3."""
    from conftest import oracle_ddim_loop
    w, x_in, noise, ref = oracle_ddim_loop(size, 50, noise_seed=7 if size == 256 else 1000)     # (the loops the bf16 / fp32s tests use)
    dif = _fp16_model(w, size, 50)
    assert dif.final_fp32_steps == 1
    out = dif.sample([x_in.cuda()], batch_size=1, noise=noise.cuda())[-1].float().cpu()
    e, db, mr = l2rel(out, ref), psnr(out, ref), rel_err(out, ref)
    dif.final_fp32_steps = 0
    pure = dif.sample([x_in.cuda()], batch_size=1, noise=noise.cuda())[-1].float().cpu()
    ep, dbp = l2rel(pure, ref), psnr(pure, ref)
    print(f"fp16 vs oracle, {size}x{size} / 50 steps: L2 {e:.3e}, {db:.1f} dB, max-rel {mr:.2e}; without the tail L2 {ep:.3e}, {dbp:.1f} dB")
    assert e < 1e-3 and db > 70.0, (e, db)
    assert ep < 2.5e-3 and dbp > 62.0, (ep, dbp)


def test_fp16_two_stream_repeatable_at_bench_size():
    """sample() at 512x512, batch 8 on two HIP streams, precision='fp16': bitwise repeatable and equal to the one-stream run (the
    binary16 build is compiled without packed-fp32 instructions like the default one: founddiff_amd/build.py)."""
    import bench
    from founddiff_amd import synth
    dev = torch.device("cuda")
    dif, _ = bench.build_model(dev, steps=50, precision="fp16")
    B = 8
    _, ld = synth.ct_phantom(B, 512, seed=10)
    x = torch.from_numpy(ld).to(dev)
    noise = torch.stack([torch.randn(1, 512, 512, generator=torch.Generator().manual_seed(1000 + i)) for i in range(B)]).to(dev)
    dif.streams = 2
    outs = [dif.sample([x], batch_size=B, noise=noise)[-1].clone() for _ in range(3)]
    dif.streams = 1
    one = dif.sample([x], batch_size=B, noise=noise)[-1].clone()
    torch.cuda.synchronize()
    for i, o in enumerate(outs):
        assert torch.equal(o, one), (i, float((o - one).abs().max()))


@pytest.mark.parametrize("cfg", [("tiny", 38, 46), ("full", 72, 88)])
def test_fp16_odd_sizes_and_one_slice_kernel_set(cfg):
    """Sizes that are not multiples of 16 (pad-to-even scans, ragged tiles: test_gpu_e2e.test_odd_sizes_fp32's geometry) on the
    binary16 build, default and low-latency kernel set, one model_predictions call against the CPU oracle: 8e-3 L2 of the raw
    residual (the bf16 engine's gate against the fp32 ENGINE on this geometry is 3e-2)."""
    from founddiff_amd import arch, synth
    from founddiff_amd.DADiff import ResidualDiffusion, UnetRes, load_weights
    from oracle import sampler
    kind, H, W = cfg
    dim, mults, clip = (32, (1, 2), TINY_CLIP) if kind == "tiny" else (64, (1, 2, 4, 8), None)
    spec = arch.da_unet_spec(dim, mults, prefix="model.unet0.", **({"clip": clip} if clip else {}))
    w = synth.synth_state_dict(spec, seed=0)
    g = torch.Generator().manual_seed(5)
    x_in = torch.rand(1, 1, H, W, generator=g) * 2 - 1
    x_t = x_in + 0.1 * torch.randn(1, 1, H, W, generator=g)
    tt = torch.full((1,), 700, dtype=torch.long)
    ref = sampler.ResidualOracle(w, prefix="model.unet0.", sampling_timesteps=2).model_predictions(x_in, x_t, tt)
    net = UnetRes(dim=dim, dim_mults=mults, num_unet=1, condition=True, objective="pred_res", test_res_or_noise="res",
                  precision="fp16", clip_cfg=clip)
    dif = ResidualDiffusion(net, image_size=H, timesteps=1000, sampling_timesteps=2, objective="pred_res", loss_type="l2",
                            condition=True, sum_scale=0.01, test_res_or_noise="res")
    load_weights(dif, w)
    dif = dif.to("cuda")
    dif.init()
    errs = []
    for ll in (False, True):
        net.unet0.low_latency = ll
        p = dif.model_predictions(x_in.cuda(), x_t.cuda(), tt.cuda())
        errs.append(l2rel(p.pred_res.cpu(), ref[0]))
    print(f"fp16 {cfg}: raw residual L2 {errs[0]:.2e} (default kernel set), {errs[1]:.2e} (one-slice set)")
    assert max(errs) < 8e-3, errs


def test_fp16_out_of_range_checkpoint_is_reported(golden):
    """(and precision='auto' falls back to the bfloat16 kernels.)  binary16 stops at 65504.  A checkpoint whose activations leave that range must not come back as an image of NaNs: sample()
    checks its result in the fp16 mode and raises, naming the mode that has the range (bf16).  Here: the tiny model with its first
    residual block's convolution scaled by 1e6 (the 3x3 output under the GroupNorm is stored in 16 bits)."""
    from founddiff_amd import _lib as L
    from founddiff_amd.DADiff import ResidualDiffusion, UnetRes
    g = golden("e2e_da_tiny")
    w = {k: v.clone() for k, v in g.weights("model.").items()}
    w["model.unet0.downs.0.0.block1.proj.bias"] += 3e5        # (the weight itself is standardised: scale the bias, which is not)
    outs = {}
    for prec in ("fp16", "bf16", "auto"):
        net = UnetRes(dim=32, dim_mults=(1, 2), num_unet=1, condition=True, objective="pred_res", test_res_or_noise="res",
                      precision=prec, clip_cfg=TINY_CLIP)
        dif = ResidualDiffusion(net, image_size=64, timesteps=1000, sampling_timesteps=4, objective="pred_res", loss_type="l2",
                                condition=True, sum_scale=0.01, test_res_or_noise="res")
        dif.load_state_dict(w, strict=False)
        dif = dif.to("cuda")
        dif.init()
        if prec == "fp16":
            with pytest.raises(L.FoundDiffHipError, match="binary16"):
                dif.sample([g["x_input"].cuda()], batch_size=2, noise=g["ddim.noise0"].cuda())
        elif prec == "auto":
            # 'auto' = fp16 until a sample() leaves the range, bf16 from then on: warns once, returns the bf16 engine's image
            assert net.unet0.kernel_precision == "fp16"
            with pytest.warns(UserWarning, match="binary16"):
                outs[prec] = dif.sample([g["x_input"].cuda()], batch_size=2, noise=g["ddim.noise0"].cuda())[-1]
            assert net.unet0.kernel_precision == "bf16" and torch.equal(outs["auto"], outs["bf16"])
            again = dif.sample([g["x_input"].cuda()], batch_size=2, noise=g["ddim.noise0"].cuda())[-1]
            assert torch.equal(again, outs["bf16"])
        else:
            outs[prec] = dif.sample([g["x_input"].cuda()], batch_size=2, noise=g["ddim.noise0"].cuda())[-1]
            assert torch.isfinite(outs[prec]).all()
