"""CPU-side checks of the host layer: parameter layout vs the reference's, product schedule vs
golden, C-ABI library loads and exports every declared symbol (no compute without a GPU)."""
import os
import re

import pytest
import torch

from conftest import ROOT


def _norm(spec):
    return {k: (tuple(v[0]), v[1]) if (len(v) == 2 and isinstance(v[1], str)) else (tuple(v), "float32")
            for k, v in spec.items()}


@pytest.mark.parametrize("fixture,kw", [
    ("e2e_da_tiny", dict(dim=32, dim_mults=(1, 2), clip=dict(layers=(2, 1, 1, 1), width=16, embed_dim=1024))),
    ("full_arch_64", dict(dim=64, dim_mults=(1, 2, 4, 8))),
])
def test_arch_spec_matches_reference_layout(golden, fixture, kw):
    from founddiff_amd import arch
    g = golden(fixture)
    ref = _norm(g.spec)
    mine = _norm(arch.da_unet_spec(prefix="model.unet0.", **kw))
    assert not [k for k in mine if k not in ref]
    assert not [k for k in mine if ref[k] != mine[k]]
    sched = {"alphas", "alphas_cumsum", "one_minus_alphas_cumsum", "betas2", "betas", "betas2_cumsum",
             "betas_cumsum", "posterior_mean_coef1", "posterior_mean_coef2", "posterior_mean_coef3",
             "posterior_variance", "posterior_log_variance_clipped"}
    extra = [k for k in ref if k not in mine and not arch.is_dead_key(k) and k not in sched]
    assert not extra, extra[:10]


def test_module_state_dict_layout(golden):
    from founddiff_amd.DADiff import ResidualDiffusion, UnetRes
    g = golden("e2e_da_tiny")
    net = UnetRes(dim=32, dim_mults=(1, 2), num_unet=1, condition=True, objective="pred_res",
                  test_res_or_noise="res", clip_cfg=dict(layers=(2, 1, 1, 1), width=16, embed_dim=1024))
    dif = ResidualDiffusion(net, image_size=64, timesteps=1000, sampling_timesteps=10, objective="pred_res",
                            loss_type="l2", condition=True, sum_scale=0.01)
    sd = dif.state_dict()
    ref = _norm(g.spec)
    for k, v in sd.items():
        assert k in ref and tuple(v.shape) == ref[k][0], k
    # a checkpoint with the reference's dead weight (second CLIP, text tower, LPIPS) loads
    w = g.weights()
    w["model.unet0.clip_model.visual.conv1.weight"] = torch.zeros(1)
    w["perceploss.net.slice1.0.weight"] = torch.zeros(1)
    w["model.unet0.dose_encoder.prompt_learner.ctx"] = torch.zeros(1)
    missing, unexpected = dif.load_state_dict(w, strict=False)
    assert not unexpected and not missing
    dif.load_state_dict(w, strict=True)


def test_load_weights_is_strict_on_live_keys(golden):
    """a checkpoint of another geometry / with renamed keys must raise, not leave zero weights
    (the reference's Trainer.load is a strict load_state_dict, src/DADiff.py:1655-1663)."""
    from founddiff_amd.DADiff import ResidualDiffusion, UnetRes, load_weights
    g = golden("e2e_da_tiny")
    clip = dict(layers=(2, 1, 1, 1), width=16, embed_dim=1024)

    def make(dim):
        net = UnetRes(dim=dim, dim_mults=(1, 2), num_unet=1, condition=True, objective="pred_res",
                      test_res_or_noise="res", clip_cfg=clip)
        return ResidualDiffusion(net, image_size=64, timesteps=1000, sampling_timesteps=10, objective="pred_res",
                                 loss_type="l2", condition=True, sum_scale=0.01)
    w = g.weights()
    sched = {"betas", "alphas_cumsum"}
    load_weights(make(32), {k: v for k, v in w.items() if k not in sched})      # schedule buffers may be absent
    renamed = dict(w)
    renamed["model.unet0.init_conv_RENAMED.weight"] = renamed.pop("model.unet0.init_conv.weight")
    with pytest.raises(RuntimeError, match="live keys missing"):
        load_weights(make(32), renamed)
    dropped = {k: v for k, v in w.items() if "mid_attn.mamba.A_logs" not in k}
    with pytest.raises(RuntimeError, match="A_logs"):
        load_weights(make(32), dropped)
    with pytest.raises(RuntimeError):                                            # other dim: shape mismatch
        load_weights(make(64), w)


def test_product_schedule_matches_reference(golden):
    from founddiff_amd.DADiff import residual_schedule
    g = golden("schedule")
    for tag, after in (("ctor", False), ("init", True)):
        s = residual_schedule(1000, after)
        for k, v in s.items():
            assert torch.equal(v, g[f"{tag}.{k}"]), (tag, k)


def test_cabi_exports_every_declared_symbol():
    from founddiff_amd import _lib as L
    hdr = open(os.path.join(ROOT, "include", "founddiff_hip.h")).read()
    declared = set(re.findall(r"\b(fd_[a-z0-9_]+)\s*\(", hdr)) - {"fd_conv_params"}
    assert declared == set(L.SIGNATURES), declared ^ set(L.SIGNATURES)
    # both builds of the library (founddiff_amd/build.py): the default one (bfloat16) and the FD_HALF_F16 one (IEEE binary16)
    for build, fmt in ((L.BF16, 0), (L.F16, 1)):
        lib = build.lib()          # raises if the .so is missing, lacks a symbol, or is the other build
        assert lib.fd_half_format() == fmt
        assert lib.fd_version() >= 100
        assert lib.fd_conv_mtiles(512, 512) == 4096      # 64-pixel GroupNorm bands
        assert lib.fd_chan_attn_nblk(512 * 512) == 256
        assert lib.fd_scan_ws_floats(1, 512, 512, 128, 4) > 0
        assert lib.fd_dev_options().decode().startswith("release build") or os.environ.get("FOUNDDIFF_DEV_BUILD") == "1"
    assert L.lib() is L.BF16.lib()


def test_fused_kernel_eligibility_rules():
    """Which Mamba-block shapes the fused kernels accept (pure host logic of the library).  The Gram-fused qkv kernel
    is the 64-channel form only: when fd_pw_dw3x3 learnt C = 128 its eligibility test (which asked fd_pw_dw3x3_ok about
    a 192-channel depthwise part) silently started to accept C = 128 blocks too."""
    from founddiff_amd import _lib as L
    lib = L.lib()
    bf = L.FD_BF16
    assert lib.fd_pw_dw3x3_ok(bf, 64, 128, 128, 512, 512) and lib.fd_pw_dw3x3_ok(bf, 64, 192, 0, 256, 256)
    assert lib.fd_pw_dw3x3_ok(bf, 128, 256, 256, 256, 256)              # C = 128 in_proj at 256x256
    assert not lib.fd_pw_dw3x3_ok(bf, 128, 384, 0, 256, 256)            # its qkv (3C = 384) stays on the row-GEMM
    assert not lib.fd_pw_dw3x3_ok(bf, 128, 256, 256, 128, 128)          # too few tiles
    assert not lib.fd_pw_dw3x3_ok(bf, 256, 512, 512, 256, 256)
    assert lib.fd_pw_dw3x3_gram_ok(bf, 64, 512, 512)
    assert not lib.fd_pw_dw3x3_gram_ok(bf, 128, 256, 256)
    assert lib.fd_dwconv_gram_ok(bf, 128, 256, 256)


def test_no_cpu_fallback():
    """The product path must fail loudly without a GPU, never fall back to torch/oracle."""
    from founddiff_amd import _lib as L
    from founddiff_amd.DADiff import Unet
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    u = Unet(32, dim_mults=(1, 2), clip_cfg=dict(layers=(1, 1, 1, 1), width=16, embed_dim=1024))
    with pytest.raises(L.FoundDiffHipError):
        u(torch.zeros(1, 2, 16, 16), torch.zeros(1))


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, "founddiff_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            assert "oracle" not in src.replace("CPU oracle", ""), fn


def test_bench_self_launch_command(monkeypatch):
    """`python bench.py --gpus N` from a bare shell spawns torch.distributed.run as a child
    (the parent never touches a GPU) and exits with the child's code."""
    import subprocess
    import sys as _sys
    import types
    sys_path0 = list(_sys.path)
    _sys.path.insert(0, ROOT)
    try:
        import bench
    finally:
        _sys.path[:] = sys_path0
    seen = {}

    def fake_run(cmd, env=None, **k):
        seen["cmd"], seen["env"] = cmd, env
        return types.SimpleNamespace(returncode=7)
    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.setattr(_sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3", "--warmup", "1"])
    with pytest.raises(SystemExit) as ei:
        bench.self_launch(types.SimpleNamespace(gpus=4))
    assert ei.value.code == 7
    cmd = seen["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-6:] == ["--gpus", "4", "--steps", "3", "--warmup", "1"] and cmd[-7].endswith("bench.py")
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_inspect_checkpoint_tool(tmp_path):
    """tools/inspect_checkpoint.py: the key / shape diff of a `model-N.pt` / `Dose-CLIP.pth` against
    arch.da_unet_spec, on synthetic checkpoints in the reference's format (src/DADiff.py:1626-1669): a good one with
    dead weight and an ema-pytorch 0.0.10 'ema' entry; one of another geometry; one with a missing and a reshaped key."""
    import importlib.util
    import os
    import torch
    from founddiff_amd import arch, synth
    from founddiff_amd.DADiff import residual_schedule
    spec_ = importlib.util.spec_from_file_location("inspect_checkpoint", os.path.join(os.path.dirname(os.path.dirname(
        os.path.abspath(__file__))), "tools", "inspect_checkpoint.py"))
    ic = importlib.util.module_from_spec(spec_)
    spec_.loader.exec_module(ic)
    clip = dict(layers=(2, 1, 1, 1), width=16, embed_dim=1024)
    spec = arch.da_unet_spec(32, (1, 2), prefix="model.unet0.", clip=clip)
    w = synth.synth_state_dict(spec, seed=3)
    model = dict(w)
    model.update(residual_schedule(1000))                                       # the 12 schedule buffers
    model["model.unet0.clip_model.visual.conv1.weight"] = torch.zeros(8, 3, 3, 3)      # the unused second CLIP (dead)
    model["model.unet0.dose_encoder.prompt_learner.ctx"] = torch.zeros(2, 16, 512)     # DA-CLIP text side (dead)
    model["perceploss.net.lin0.model.1.weight"] = torch.zeros(1, 64, 1, 1)             # LPIPS (dead)
    ema = {"ema_model." + k: v for k, v in model.items()}
    ema.update({"online_model." + k: v for k, v in model.items()})
    ema["initted"], ema["step"] = torch.tensor(True), torch.tensor(400000)
    good = tmp_path / "model-400.pt"
    torch.save({"step": 400000, "model": model, "opt0": {}, "ema": ema, "scaler": None}, good)
    rep = ic.inspect_checkpoint(str(good), dim=32, dim_mults=(1, 2), clip=clip)
    assert rep["ok"] and rep["step"] == 400000
    assert rep["model"]["missing"] == [] and rep["model"]["unexpected"] == [] and len(rep["model"]["dead"]) == 3
    assert rep["ema"]["layout_is_ema_pytorch_0_0_10"] and rep["ema"]["ok"] and rep["ema"]["n_online_model"] == len(model)
    assert rep["schedule_buffers"]["present"] == 12 and rep["schedule_buffers"]["match_ctor"]
    # another geometry: the shipped spec does not match a dim-32 checkpoint
    rep = ic.inspect_checkpoint(str(good), dim=64, dim_mults=(1, 2, 4, 8), clip=clip)
    assert not rep["ok"] and rep["model"]["missing"] and (rep["model"]["shape_mismatch"] or rep["model"]["unexpected"])
    # one key missing, one reshaped, one unknown
    bad = dict(model)
    bad.pop("model.unet0.final_conv.weight")
    bad["model.unet0.init_conv.weight"] = torch.zeros(32, 3, 7, 7)
    bad["model.unet0.some_new_module.weight"] = torch.zeros(4)
    badp = tmp_path / "model-1.pt"
    torch.save({"step": 1, "model": bad, "opt0": {}, "ema": None, "scaler": None}, badp)
    rep = ic.inspect_checkpoint(str(badp), dim=32, dim_mults=(1, 2), clip=clip)
    assert not rep["ok"] and rep["model"]["missing"] == ["model.unet0.final_conv.weight"]
    assert rep["model"]["shape_mismatch"][0][0] == "model.unet0.init_conv.weight"
    assert rep["model"]["unexpected"] == ["model.unet0.some_new_module.weight"] and not rep["ema"]["present"]
    # not a checkpoint dict at all
    torch.save({"weights": 1}, tmp_path / "x.pt")
    assert not ic.inspect_checkpoint(str(tmp_path / "x.pt"))["ok"]
    # Dose-CLIP.pth: CLIPIQA.state_dict()
    iqa = {k[len("model.unet0.dose_encoder."):]: v for k, v in w.items() if k.startswith("model.unet0.dose_encoder.")}
    iqa["prompt_learner.ctx"] = torch.zeros(2, 16, 512)
    torch.save(iqa, tmp_path / "Dose-CLIP.pth")
    rep = ic.inspect_dose_clip(str(tmp_path / "Dose-CLIP.pth"), clip=clip)
    assert rep["ok"] and rep["dead"] == ["dose_encoder.prompt_learner.ctx"]
    iqa.pop("head1.0.weight")
    torch.save(iqa, tmp_path / "Dose-CLIP.pth")
    rep = ic.inspect_dose_clip(str(tmp_path / "Dose-CLIP.pth"), clip=clip)
    assert not rep["ok"] and rep["missing"] == ["dose_encoder.head1.0.weight"]


def test_inline_asm_mfma_drains_in_the_isa():
    """Kernels whose MFMAs are inline asm wait out the last results behind an s_nop block the accumulators are tied to
    (FD_MFMA_ASM_DRAIN, csrc/fd_common.h).  The ISA is checked as well (tools/kres.py --check-drain, hipcc cross-compiles
    without a GPU): no VALU instruction reads an MFMA destination between the last v_mfma and the drain block."""
    import shutil
    import subprocess
    import sys
    if not shutil.which("/opt/rocm/bin/hipcc"):
        pytest.skip("no hipcc")
    files = [os.path.join(ROOT, "founddiff_amd", "csrc", f) for f in ("fd_conv3x3_rw.hip", "fd_downfuse.hip", "fd_pwgemm.hip", "fd_conv3x3.hip")]
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "kres.py"), "--check-drain"] + files, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert r.stdout.count("0 early read(s)") == 4, r.stdout


def test_kernel_scratch_ledger():
    """Register-allocation accidents fail here instead of waiting for a profile: every kernel of the library is held to a
    scratch ledger read from hipcc's kernel-resource-usage remarks of the objects build() compiled
    (founddiff_amd.build.resources()).  Round 5: an early return in fd_softplus_fast turned the unrolled steps of the fp32
    scan's phase A into control flow the allocator could not fit in its occupancy step -- 450-820 bytes of scratch, 3-5 x the
    time of the same kernel without it -- and nothing but a launch table of the fp32s mode showed it."""
    import re
    import shutil
    if not shutil.which("/opt/rocm/bin/hipcc"):
        pytest.skip("no hipcc")
    from founddiff_amd import build
    for half in ("bf16", "fp16"):              # both builds of the library: the same ledger
        build.build(half=half)
        _check_ledger(build.resources(half), half)


def _check_ledger(res, half):
    import re
    assert res.get("fd_scan.hip") and res.get("fd_conv3x3.hip"), "no resource remarks beside the objects: rebuild with build(force=True)"
    model_nr = {(4, 4), (8, 4), (16, 8), (32, 16), (32, 32), (16, 16), (8, 8)}     # (d_state, dt_rank) of the shipped architecture
    ledger = [          # (file, pattern of the mangled name, bytes of scratch per lane it may use)
        ("fd_conv.hip", r"conv_igemm_kernelI\w+Li256ELi256E", 192),        # the 256 x 256 tile sits at its 256-register cap
        ("fd_conv3x3.hip", r"conv3x3_halo_kernelILi128ELi8ELb0ELb0ELb1E", 32),     # split-bf16 form: fp32 halo registers
        ("fd_pwdw.hip", r"pwdw_gram_kernel|dwconv_gram_kernel|pwdw_kernelILi64ELb1E", 48),    # (round 6: built without packed-fp32 instructions, build.NO_PACKED_F32: 36 bytes in the project_out form)
    ]
    bad = []
    for f, tab in res.items():
        for name, r in tab.items():
            limit = 0
            m = re.search(r"scan_chunk_kernelI(DF16b|DF16_|f)Li(\d+)ELi(\d+)E", name)
            if f == "fd_scan.hip" and m:
                # combinations the architecture uses: a few spilled values at the 96-register occupancy step; the others
                # (dt_rank 32 at d_state 8 / 16) only have to build
                limit = 256 if (int(m.group(2)), int(m.group(3))) in model_nr else 512
            for lf, pat, lim in ledger:
                if lf == f and re.search(pat, name):
                    limit = lim
            if r.get("scratch", 0) > limit:
                bad.append((half, f, name, r["scratch"], limit))
    assert not bad, bad
    # conv3x3_halo_kernel's halo registers are written by inline-asm loads (hidden from the compiler's waitcnt pass).
    # A register parked in an AGPR or in scratch before its load has landed would park garbage: every instantiation that uses
    # the asm loads must have neither -- the one that spills (the 128-channel split form) is compiled with plain loads instead.
    for name, r in res["fd_conv3x3.hip"].items():
        if "conv3x3_halo_kernel" not in name:
            continue
        plain = re.search(r"conv3x3_halo_kernelILi128ELi8ELb0ELb0ELb1E", name) is not None     # <128, 8, F8 = 0, UP = 0, SPL = 1>
        if plain:
            continue
        assert r.get("agpr", 0) == 0 and r.get("scratch", 0) == 0, (half, name, r)


def test_traffic_json_serves_the_bench_rooflines():
    """profiles/traffic.json (PMC passes of tools/profile_round*.sh, summarised by tools/traffic.py) is where bench.py's roofline
    entries take `traffic` from.  The dominant entry is a scan GROUP: its lookup key is formed in bench.roofline_leg from the group's
    name -- a renamed group or a traffic.json written by an older traffic.py gives `traffic: null` in the driver's bench line without
    any error (round 6 shipped one such record).  Hold the keys together here."""
    import json
    t = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
    groups = t.get("scan_hbm_bytes_per_launch")
    assert isinstance(groups, dict), "traffic.json has no scan groups: python tools/traffic.py --resummarize profiles/traffic.json"
    src = open(os.path.join(ROOT, "bench.py")).read()
    for name, suffix in (("scan_seq_kernel (single pass, N >= 16, L <= 1024)", ""),
                         ("scan_chunk_kernel x2 + scan_carry_kernel, level 0 (d_inner 128, N 4, two channels per lane)", "|l0"),
                         ("scan_chunk_kernel x2 + scan_carry_kernel, levels 1-2 (N 8..16)", "|l12")):
        assert f'"{name}"' in src, name                       # the group names bench.py uses
        key = name.split(",")[0] + suffix                      # ... and the key it derives from them
        assert groups.get(key), (key, sorted(groups))
        assert 5e7 < groups[key] < 5e9
    for k in ("total_hbm_bytes_per_forward", "conv3x3_halo128_hbm_bytes_per_launch", "pwdw_gram_hbm_bytes_per_launch",
              "gemm_rows_zre_hbm_bytes_per_launch", "conv3x3_up_hbm_bytes_per_launch", "csrc_sha"):
        assert t.get(k), k
