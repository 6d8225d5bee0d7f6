"""GPU tests of the evaluation harness either side of the sampling path: device metrics vs the
CPU oracle, and Trainer.load / test / sample on a reference-format checkpoint."""
import os

import numpy as np
import pytest
import torch

from conftest import rel_err

pytestmark = pytest.mark.gpu
TINY_CLIP = dict(layers=(2, 1, 1, 1), width=16, embed_dim=1024)


def test_metrics_vs_oracle():
    from founddiff_amd import metrics, synth
    from oracle import metrics as om
    nd, ld = synth.ct_phantom(3, 72, seed=4)          # 72 is not a multiple of the 16-pixel tile
    a, b = torch.from_numpy(ld), torch.from_numpy(nd)
    got = metrics.compute_metrics(a.cuda(), b.cuda()).cpu()
    for i in range(3):
        assert abs(float(got[i, 0]) - float(om.psnr(a[i:i + 1], b[i:i + 1]))) < 1e-3
        assert abs(float(got[i, 1]) - float(om.ssim(a[i:i + 1], b[i:i + 1]))) < 1e-5
        assert abs(float(got[i, 2]) - float(om.rmse(a[i:i + 1], b[i:i + 1]))) < 1e-6
    assert abs(float(metrics.compute_ssim(a.cuda(), a.cuda())) - 1.0) < 1e-6


def test_trainer_load_test_sample(tmp_path):
    from types import SimpleNamespace
    from founddiff_amd import arch, synth
    from founddiff_amd.DADiff import ResidualDiffusion, Trainer, UnetRes
    from founddiff_amd.data import SyntheticCTDataset
    from oracle import metrics as om, sampler

    def make():
        net = UnetRes(dim=32, dim_mults=(1, 2), num_unet=1, condition=True, objective="pred_res",
                      test_res_or_noise="res", precision="fp32", clip_cfg=TINY_CLIP)
        return ResidualDiffusion(net, image_size=64, timesteps=1000, sampling_timesteps=2, objective="pred_res",
                                 loss_type="l2", condition=True, sum_scale=0.01, test_res_or_noise="res")
    spec = arch.da_unet_spec(32, (1, 2), prefix="model.unet0.", clip=TINY_CLIP)
    w = synth.synth_state_dict(spec, seed=5)
    # a checkpoint in the reference's format, incl. dead weight and an EMA copy that must win
    ck = tmp_path / "ck" / "sample"
    ck.mkdir(parents=True)
    stale = {k: torch.zeros_like(v) for k, v in w.items()}
    ema = {"ema_model." + k: v for k, v in w.items()}
    ema.update({"online_model." + k: v for k, v in stale.items()})
    ema["initted"], ema["step"] = torch.tensor(True), torch.tensor(7)
    stale["perceploss.net.lin0.model.1.weight"] = torch.zeros(1, 64, 1, 1)
    torch.save({"step": 123, "model": stale, "opt0": {}, "ema": ema, "scaler": None}, ck / "model-400.pt")
    ds = SyntheticCTDataset(3, 64, seed=2)
    tr = Trainer(SimpleNamespace(is_train=False), make(), None, num_samples=1, condition=True, num_unet=1,
                 checkpoint_folder=str(tmp_path / "ck"), is_train=False, dataset=ds)
    assert tr.accelerator.is_local_main_process
    tr.load(400)
    assert tr.step == 123
    tr.load(999)                                  # missing file: skipped silently
    torch.manual_seed(3)
    psnr, ssim, rmse = tr.test(last=True)
    # same run through the CPU oracle: reproduce the global-generator noise draws
    torch.manual_seed(3)
    orc = sampler.ResidualOracle(w, prefix="model.unet0.", sampling_timesteps=2)
    ps = []
    for i in range(3):
        y, x = ds[i]
        noise = torch.randn((1, 1, 64, 64), device="cuda").cpu()
        ref = orc.sample(x[None], noise)[-1]
        out = np.load(os.path.join(tr.results_folder, ds.load_name(i)[:-4] + ".npy"))
        assert out.shape == (64, 64)
        assert rel_err(torch.from_numpy(out), ref[0, 0]) < 1e-3
        ps.append(float(om.psnr(ref, y[None])))
    assert abs(psnr - float(np.mean(ps))) < 0.05
    # per-anatomy / per-dose means of the log tail (src/DADiff.py:1918-1952), with groups sized for 3 items
    assert set(tr.test_group_means) == {"ab", "lung", "head"}
    assert tr.test_image_names == [f"synthetic-quarter-{i:04d}.png" for i in range(3)]      # name rule, 1904-1907
    tr.test_groups = (("ab", 1), ("head", 0))
    g = tr._log_groups()
    assert abs(g["ab"]["mean"][0] - psnr) < 1e-4 and abs(g["ab"]["dose"][1][0] - tr.test_running_psnr[1]) < 1e-6
    # same seed -> same run (the loop re-initialises the schedule and the running lists); a batched pass runs too
    one = list(tr.test_running_psnr)
    torch.manual_seed(3)
    tr.test(last=True, batch_size=1)
    assert np.allclose(tr.test_running_psnr, one, atol=1e-4)
    assert len(tr.test(last=True, batch_size=2)) == 3 and len(tr.test_running_psnr) == 3
    # crop_patch (src/DADiff.py:1872-1886): only the SAVED array is cropped by the dataset's pad size, the metrics use
    # the uncropped prediction
    ds.get_pad_size = lambda i: (4, 6)
    tr.crop_patch = True
    torch.manual_seed(3)
    tr.test(last=True, batch_size=1)
    assert np.allclose(tr.test_running_psnr, one, atol=1e-4)
    assert np.load(os.path.join(tr.results_folder, ds.load_name(2)[:-4] + ".npy")).shape == (60, 58)
    tr.crop_patch = False
    # sample=True: inputs + outputs, no metrics, nothing saved (src/DADiff.py:1863-1866)
    assert tr.test(sample=True) is None and tr.test_running_psnr == []
    # preview grid: PNG in the HU window, like torchvision's save_image (1792-1812); FID: one PNG per image
    from PIL import Image
    assert tr.sample(1) == 1
    im = np.asarray(Image.open(os.path.join(tr.results_folder, "sample-1.png")))
    assert im.shape == (4 * 66 + 2, 66 + 2, 3)                       # [NDCT, LDCT, x_T, output] x 1 sample, nrow 1
    tr.total_n_samples = 50000
    assert tr.sample(10, FID=True) == 11 and os.path.exists(os.path.join(tr.results_folder, "sample-10.png"))
    # a checkpoint of another geometry must not load silently
    bad = {k: v for k, v in w.items() if "mid_attn" not in k}
    torch.save({"step": 1, "model": bad, "opt0": {}, "ema": None, "scaler": None}, ck / "model-7.pt")
    with pytest.raises(RuntimeError, match="live keys missing"):
        tr.load(7)


def test_dose_clip_file_and_ema_layout(tmp_path):
    """f3: (i) `Dose-CLIP.pth` (the CLIPIQA state_dict the reference loads strictly at construction,
    src/DADiff.py:595-596) through Unet.load_dose_clip; (ii) the `ema` entry in ema-pytorch 0.0.10's layout
    (install.yaml:186): online_model.* / ema_model.* / initted / step -- the ema_model copy wins."""
    from founddiff_amd import arch, synth
    from founddiff_amd.DADiff import ResidualDiffusion, Unet, UnetRes, _EMAView
    spec = arch.da_unet_spec(32, (1, 2), prefix="", clip=TINY_CLIP)
    w = synth.synth_state_dict(spec, seed=9)
    iqa = {k[len("dose_encoder."):]: v for k, v in w.items() if k.startswith("dose_encoder.")}
    iqa["prompt_learner.ctx"] = torch.zeros(2, 16, 512)              # dead weight of the real file: ignored
    torch.save(iqa, tmp_path / "Dose-CLIP.pth")
    u = Unet(32, dim_mults=(1, 2), precision="fp32", clip_cfg=TINY_CLIP)
    u.load_dose_clip(str(tmp_path / "Dose-CLIP.pth"))
    for k, v in u.state_dict().items():
        if k.startswith("dose_encoder."):
            assert torch.equal(v, w[k]), k
    broken = dict(iqa)
    broken.pop("head1.0.weight")
    with pytest.raises(RuntimeError, match="head1.0.weight"):
        u.load_dose_clip(broken)
    # ema-pytorch 0.0.10 state_dict
    net = UnetRes(dim=32, dim_mults=(1, 2), num_unet=1, condition=True, objective="pred_res", test_res_or_noise="res",
                  precision="fp32", clip_cfg=TINY_CLIP)
    dif = ResidualDiffusion(net, image_size=64, timesteps=1000, sampling_timesteps=2, objective="pred_res",
                            loss_type="l2", condition=True, sum_scale=0.01, test_res_or_noise="res")
    full = {"model.unet0." + k: v for k, v in w.items()}
    ema = {"ema_model." + k: v for k, v in full.items()}
    ema.update({"online_model." + k: torch.zeros_like(v) for k, v in full.items()})
    ema["initted"], ema["step"] = torch.Tensor([True]), torch.tensor([400000])
    _EMAView(dif).load_state_dict(ema)
    assert torch.equal(dif.state_dict()["model.unet0.init_conv.weight"], full["model.unet0.init_conv.weight"])
    with pytest.raises(RuntimeError):
        _EMAView(dif).load_state_dict({k: v for k, v in ema.items() if "final_conv" not in k})
