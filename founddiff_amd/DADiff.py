"""Drop-in mirror of the reference's `src/DADiff.py` sampling API, executed on MI355X HIP kernels.

    from founddiff_amd.DADiff import ResidualDiffusion, Trainer, Unet, UnetRes, set_seed

Same class names, constructor kwargs, method names/signatures and state_dict key layout as the
reference (`/root/reference/src/DADiff.py`; contract in SURVEY.md section 8b), so
`checkpoints/<name>/sample/model-N.pt` loads unchanged.  The modules below hold parameters only:
`forward` never runs a torch op on activations -- it hands device pointers to libfounddiff_hip.so
through `founddiff_amd.engine.DAEngine`.  Training (`p_losses`, optimiser, EMA update) is out of
scope (inference engine).
"""
import ctypes as C
import math
import os
import random
from collections import namedtuple

import numpy as np
import torch
from torch import nn

from . import _lib as L
from . import arch
from .engine import DAEngine, _p

ModelResPrediction = namedtuple("ModelResPrediction", ["pred_res", "pred_noise", "pred_x_start"])


def set_seed(SEED):
    """src/DADiff.py:65-70"""
    torch.manual_seed(SEED)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(SEED)
    np.random.seed(SEED)
    random.seed(SEED)


def exists(x):
    return x is not None


def default(val, d):
    if exists(val):
        return val
    return d() if callable(d) else d


def normalize_to_neg_one_to_one(img):
    if isinstance(img, list):
        return [_affine(t, 2.0, -1.0) for t in img]
    return _affine(img, 2.0, -1.0)


def unnormalize_to_zero_to_one(img):
    if isinstance(img, list):
        return [_affine(t, 0.5, 0.5) for t in img]
    return _affine(img, 0.5, 0.5)


def _stream(t):
    import ctypes as C
    return C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def _affine(x, a, b):
    x = x.contiguous().float()
    out = torch.empty_like(x)
    L.call("fd_affine_f32", _p(x), a, b, _p(out), x.numel(), _stream(x))
    return out


class _ParamTree(nn.Module):
    """Parameter container reproducing a flat {dotted.key: shape} layout as nested modules."""

    def _attach(self, key, shape, dtype=torch.float32, buffer=False):
        parts = key.split(".")
        mod = self
        for p in parts[:-1]:
            if not hasattr(mod, p):
                mod.add_module(p, _ParamTree())
            mod = getattr(mod, p)
        if buffer or dtype != torch.float32:
            mod.register_buffer(parts[-1], torch.zeros(shape, dtype=dtype))
        else:
            mod.register_parameter(parts[-1], nn.Parameter(torch.zeros(shape), requires_grad=False))


def _build_tree(root, spec):
    for k, s in spec.items():
        if isinstance(s, tuple) and len(s) == 2 and isinstance(s[1], str):
            root._attach(k, s[0], getattr(torch, s[1]), buffer=True)
        else:
            leaf = k.rsplit(".", 1)[-1]
            root._attach(k, s, buffer=leaf in ("running_mean", "running_var"))


class Unet(_ParamTree):
    """DA-CLIP conditioned U-Net (reference src/DADiff.py:530-740).  `condition` is forced on
    as in the reference (line 586).  Extra kwargs: `precision` selects the kernel mode -- 'bf16' (default) | 'fp16' (the same kernels
    on the library's IEEE-binary16 build: 8 x smaller drift at 0.96 of the speed, binary16's range) | 'auto' ('fp16' until a sample()
    leaves that range, 'bf16' from then on) | 'fp32s' | 'fp32' | 'fp8' (bf16
    kernels with e4m3 weights on the fp8 MFMA for the 3x3 convolutions, BASELINE configs[4]); `clip_cfg` overrides the RN50 DA-CLIP geometry (tests use a shrunken one)."""

    def __init__(self, dim, init_dim=None, out_dim=None, dim_mults=(1, 2, 4, 8), channels=1,
                 self_condition=False, resnet_block_groups=8, learned_variance=False,
                 learned_sinusoidal_cond=False, random_fourier_features=False, learned_sinusoidal_dim=16,
                 condition=False, input_condition=False, precision=None, clip_cfg=None):
        super().__init__()
        if self_condition or learned_variance or learned_sinusoidal_cond or \
                random_fourier_features or resnet_block_groups != 8 or (init_dim not in (None, dim)) or channels != 1:
            raise NotImplementedError("Unet is built for channels=1, condition=True, no self-conditioning / learned "
                                      "variance / learned sinusoidal embedding (SURVEY 8f-4)")
        self.input_condition = input_condition
        self.channels = channels
        self.self_condition = self_condition
        self.dim, self.dim_mults = dim, tuple(dim_mults)
        self.out_dim = default(out_dim, channels)
        self.random_or_learned_sinusoidal_cond = False
        self.precision = precision or os.environ.get("FOUNDDIFF_PRECISION", "bf16")
        self._auto_bf16 = False          # precision='auto': set once a sample() left binary16's range (ResidualDiffusion.sample)
        # kernel set for one slice at a time (DAEngine low_latency); Trainer.test(batch_size=1) turns it on
        self.low_latency = bool(int(os.environ.get("FOUNDDIFF_LOW_LATENCY", "0")))
        self.clip_cfg = clip_cfg or arch.RN50
        _build_tree(self, arch.da_unet_spec(dim, self.dim_mults, channels, "", self.clip_cfg, input_condition))
        self._engine = None

    # --- checkpoint handling: accept-and-ignore the dead weight a real checkpoint carries
    def load_state_dict(self, state_dict, strict=True, assign=False):
        live = {k: v for k, v in state_dict.items() if not arch.is_dead_key("unet0." + k, "unet0.")}
        self._engine = None
        return super().load_state_dict(live, strict=strict, assign=assign)

    def _load_from_state_dict(self, *a, **k):
        self._engine = None
        return super()._load_from_state_dict(*a, **k)

    def _apply(self, fn, *a, **k):
        # .to(device) / .cuda() / dtype casts move the parameters: the packed weights of the engine are stale
        self._engine = None
        return super()._apply(fn, *a, **k)

    def load_dose_clip(self, path_or_state="Dose-CLIP.pth"):
        """The reference builds its dose encoder from a separate file at construction time:
        `clipiqa.load_state_dict(torch.load('Dose-CLIP.pth'), strict=True)` (src/DADiff.py:595-596).  A
        `model-N.pt` checkpoint carries the same tensors under `dose_encoder.*`, so this is only needed for a
        model that is not loaded from a checkpoint.  Strict on every key that is live on the sampling path (the
        RN50 visual tower + head1 / head2); the text tower and prompt learner of the file are dead weight."""
        sd = path_or_state
        if not isinstance(sd, dict):
            sd = torch.load(str(sd), map_location="cpu", weights_only=False)
        mine = {k for k in self.state_dict() if k.startswith("dose_encoder.")}
        live = {"dose_encoder." + k: v for k, v in sd.items() if "dose_encoder." + k in mine}
        missing = sorted(mine - set(live))
        if missing:
            raise RuntimeError(f"Dose-CLIP state_dict lacks {len(missing)} live keys, e.g. {[m[len('dose_encoder.'):] for m in missing[:3]]}")
        res = super().load_state_dict(live, strict=False)
        assert not res.unexpected_keys
        self._engine = None
        return res

    @property
    def kernel_precision(self):
        """the kernel mode `precision` stands for: 'auto' = 'fp16' (the 16-bit kernels on the binary16 build: 8 x smaller drift than
        'bf16' at 0.96 of its speed) until a sample() of this model has produced a non-finite image -- an activation beyond binary16's
        65504 -- and 'bf16' from then on"""
        if self.precision == "auto":
            return "bf16" if self._auto_bf16 else "fp16"
        return self.precision

    def engine(self, precision=None, slot=0):
        """The packed-weight HIP engine of this UNet for `precision` (default: self.precision); one per
        precision is kept (the samplers run their last step(s) on the fp32 engine, see ResidualDiffusion).
        `slot`: engines with their own workspaces for concurrent half-batches (ResidualDiffusion.sample)."""
        prec = precision or self.kernel_precision
        if self._engine is None:
            self._engine = {}
        key = (prec, slot) if not self.low_latency else (prec, slot, "ll")
        eng = self._engine.get(key)
        if eng is None:
            dev = next(self.parameters()).device
            if dev.type != "cuda":
                raise L.FoundDiffHipError("founddiff_amd runs on MI355X only: move the model to a ROCm device "
                                          "(`.to('cuda')`); there is no CPU path")
            eng = self._engine[key] = DAEngine(self.state_dict(), "", dev, prec, low_latency=self.low_latency)
        return eng

    @torch.no_grad()
    def encode_condition(self, x_cond):
        return self.engine().encode_condition(x_cond.contiguous().float())

    @torch.no_grad()
    def forward(self, x, time, x_self_cond=None, reuse_condition=False):
        """x (B,2,H,W) = cat(x_t, x_input) -- (B,3,H,W) with the input_condition plane; time (B,) float.
        Returns (B,1,H,W) fp32."""
        eng = self.engine()
        x = x.float()
        x_t, x_in = x[:, 0:1].contiguous(), x[:, 1:2].contiguous()
        x_c2 = x[:, 2:3].contiguous() if self.input_condition else None
        if not reuse_condition:
            eng.encode_condition(x_in)
        return eng.forward(x_t, x_in, time.float().contiguous(), x_cond2=x_c2).clone()


class UnetRes(nn.Module):
    """reference src/DADiff.py:743-836: one UNet, or two (residual + noise) for the objectives
    'pred_res_noise' / 'pred_x0_noise'."""

    def __init__(self, dim, init_dim=None, out_dim=None, dim_mults=(1, 2, 4, 8), channels=1, self_condition=False,
                 resnet_block_groups=8, learned_variance=False, learned_sinusoidal_cond=False,
                 random_fourier_features=False, learned_sinusoidal_dim=16, num_unet=1, condition=False,
                 input_condition=False, objective="pred_res_noise", test_res_or_noise="res_noise",
                 precision=None, clip_cfg=None):
        super().__init__()
        if num_unet not in (1, 2):
            raise ValueError("num_unet must be 1 or 2")
        self.condition = condition
        self.input_condition = input_condition
        self.channels = channels
        self.out_dim = default(out_dim, channels * (1 if not learned_variance else 2))
        self.random_or_learned_sinusoidal_cond = learned_sinusoidal_cond or random_fourier_features
        self.self_condition = self_condition
        self.num_unet = num_unet
        self.objective = objective
        self.test_res_or_noise = test_res_or_noise
        self.unet0 = Unet(dim, init_dim=init_dim, out_dim=out_dim, dim_mults=dim_mults, channels=channels,
                          self_condition=self_condition, resnet_block_groups=resnet_block_groups,
                          learned_variance=learned_variance, learned_sinusoidal_cond=learned_sinusoidal_cond,
                          random_fourier_features=random_fourier_features,
                          learned_sinusoidal_dim=learned_sinusoidal_dim, condition=condition,
                          input_condition=input_condition, precision=precision, clip_cfg=clip_cfg)
        if num_unet == 2:
            self.unet1 = Unet(dim, init_dim=init_dim, out_dim=out_dim, dim_mults=dim_mults, channels=channels,
                              self_condition=self_condition, resnet_block_groups=resnet_block_groups,
                              learned_variance=learned_variance, learned_sinusoidal_cond=learned_sinusoidal_cond,
                              random_fourier_features=random_fourier_features,
                              learned_sinusoidal_dim=learned_sinusoidal_dim, condition=condition,
                              input_condition=input_condition, precision=precision, clip_cfg=clip_cfg)

    def forward(self, x, time, x_self_cond=None, reuse_condition=False):
        """src/DADiff.py:817-836.  time = [alphas_cumsum[t]*T, betas_cumsum[t]*T]."""
        kw = dict(x_self_cond=x_self_cond, reuse_condition=reuse_condition)
        if self.num_unet == 2:
            if self.test_res_or_noise == "res_noise":
                return self.unet0(x, time[0], **kw), self.unet1(x, time[1], **kw)
            if self.test_res_or_noise == "res":
                return self.unet0(x, time[0], **kw), 0
            if self.test_res_or_noise == "noise":
                return 0, self.unet1(x, time[1], **kw)
            raise ValueError(f"test_res_or_noise={self.test_res_or_noise!r}")
        if self.objective == "pred_noise":
            time = time[1]
        elif self.objective == "pred_res":
            time = time[0]
        else:
            raise ValueError(f"objective {self.objective!r} needs num_unet=2 (src/DADiff.py:826-831)")
        return [self.unet0(x, time, **kw)]


def residual_schedule(timesteps=1000, after_init=False):
    """The 12 schedule vectors (src/DADiff.py:946-1027 for __init__, 1033-1118 for init())."""
    import torch.nn.functional as F
    betas = torch.linspace(1e-4, 0.02, timesteps, dtype=torch.float32)
    acp = torch.cumprod(1.0 - betas, dim=0)
    acs = 1 - acp ** 0.5
    b2cs = 1 - acp
    acs_prev = F.pad(acs[:-1], (1, 0), value=1.0)
    b2cs_prev = F.pad(b2cs[:-1], (1, 0), value=1.0)
    alphas = acs - acs_prev
    betas2 = b2cs - b2cs_prev
    alphas[0] = alphas[1] if after_init else 0
    betas2[0] = betas2[1] if after_init else 0
    pv = betas2 * b2cs_prev / b2cs
    pv[0] = 0
    out = dict(alphas=alphas, alphas_cumsum=acs, one_minus_alphas_cumsum=1 - acs, betas2=betas2,
               betas=torch.sqrt(betas2), betas2_cumsum=b2cs, betas_cumsum=torch.sqrt(b2cs),
               posterior_mean_coef1=b2cs_prev / b2cs,
               posterior_mean_coef2=(betas2 * acs_prev - b2cs_prev * alphas) / b2cs,
               posterior_mean_coef3=betas2 / b2cs, posterior_variance=pv,
               posterior_log_variance_clipped=torch.log(pv.clamp(min=1e-20)))
    out["posterior_mean_coef1"][0] = 0
    out["posterior_mean_coef2"][0] = 0
    out["posterior_mean_coef3"][0] = 1
    out["one_minus_alphas_cumsum"][-1] = 1e-6
    return {k: v.to(torch.float32) for k, v in out.items()}


def load_weights(module, state_dict, what="checkpoint"):
    """`load_state_dict` the way the reference's strict `Trainer.load` does (src/DADiff.py:1655-1663) for every
    key that is live on the sampling path: dead weight is dropped by the modules' own `load_state_dict`, the
    12 schedule buffers may be absent (they are re-derived by `init()`), anything else missing or unexpected
    raises -- a checkpoint of another dim / dim_mults / num_unet / input_condition must not leave zero-
    initialised weights behind silently."""
    res = module.load_state_dict(state_dict, strict=False)
    sched = set(residual_schedule(1000).keys())
    missing = [k for k in res.missing_keys if k not in sched]
    unexpected = list(res.unexpected_keys)
    if missing or unexpected:
        raise RuntimeError(f"{what} does not match this model: {len(missing)} live keys missing "
                           f"(e.g. {missing[:3]}), {len(unexpected)} unexpected (e.g. {unexpected[:3]})")
    return res


class ResidualDiffusion(nn.Module):
    """Residual (RDDM-style) diffusion sampler, reference src/DADiff.py:908-1380.

    Differences a caller can see: (1) the t-independent DA-CLIP branch is evaluated once per
    `sample()` instead of once per step; (2) `sample/ddim_sample/p_sample_loop/p_sample` take an
    optional `noise=` so that runs are reproducible against a CPU reference (the reference draws
    from the global generator; the default here does too)."""

    def __init__(self, model, *, image_size, timesteps=1000, sampling_timesteps=None, loss_type="l1",
                 objective="pred_res_noise", ddim_sampling_eta=0., condition=False, sum_scale=None,
                 input_condition=False, input_condition_mask=False, test_res_or_noise="None",
                 use_graph=True, final_fp32_steps=None):
        super().__init__()
        assert not (type(self) == ResidualDiffusion and model.channels != model.out_dim)
        assert not model.random_or_learned_sinusoidal_cond
        if objective not in ("pred_res", "pred_noise", "pred_res_noise", "pred_x0_noise"):
            raise ValueError(f"unknown objective {objective!r}")
        if not condition:
            raise NotImplementedError("condition=True is what the DA-CLIP conditioned Unet supports "
                                      "(it reads x[:,1], src/DADiff.py:692)")
        if objective in ("pred_res_noise", "pred_x0_noise") and getattr(model, "num_unet", 1) != 2:
            raise ValueError(f"objective {objective!r} needs UnetRes(num_unet=2) (src/DADiff.py:826-831)")
        if timesteps != 1000:
            raise NotImplementedError("init() of the reference hard-codes 1000 timesteps (src/DADiff.py:1034)")
        self.model = model
        self.channels = model.channels
        self.self_condition = model.self_condition
        self.image_size = image_size
        self.objective = objective
        self.condition = condition
        self.input_condition = input_condition
        self.input_condition_mask = input_condition_mask
        self.test_res_or_noise = test_res_or_noise
        self.sum_scale = sum_scale if sum_scale else 0.01
        ddim_sampling_eta = 0.  # forced when condition=True (src/DADiff.py:942)
        self.num_timesteps = int(timesteps)
        self.loss_type = loss_type
        self.sampling_timesteps = default(sampling_timesteps, timesteps)
        assert self.sampling_timesteps <= timesteps
        self.is_ddim_sampling = self.sampling_timesteps < timesteps
        self.ddim_sampling_eta = ddim_sampling_eta
        self.use_graph = use_graph
        # precision schedule of the bf16 mode: the last `final_fp32_steps` UNet forwards of a sampling loop run
        # on the fp32 (parity-mode) engine.  The returned image is clamp(x_input - pred_res) of the LAST forward
        # (src/DADiff.py:1317-1318, 1206), so its rounding error reaches the output undamped while the earlier
        # steps' errors only enter through x_t, weighted by their alpha increments.
        # Default 1: measured at 512x512 / 50 DDIM steps, bf16 alone drifts 1.08e-2 L2 (50.0 dB) from the fp32
        # engine, with the last step in fp32 4.4e-3 (57.8 dB) for +5..7 % time; a second fp32 step buys nothing.
        if final_fp32_steps is None:
            final_fp32_steps = int(os.environ.get("FOUNDDIFF_FINAL_FP32_STEPS", "1"))
        self.final_fp32_steps = int(final_fp32_steps)
        # ... and of that step only the outermost resolution level(s) -- init_conv, downs[0], ups[-1], the final
        # block: 53 % of a forward's bf16 drift originates there (profiles/r02_drift_table.md) at a third of the fp32
        # forward's time -- run one class up, the levels below stay on the fast engine (DAEngine.forward_hybrid).
        # 0: the whole step runs on the higher-precision engine.  Measured at 512x512 / 50 steps against the fp32
        # engine (tools/e2e_drift.py): pure bf16 1.12e-2 L2 / 49.8 dB at 11.0 slices/s; whole last step in fp32
        # 4.6e-3 / 57.5 dB at 10.2; levels 0-1 of the last step in fp32 (default 2) 7.0e-3 / 53.8 dB at 10.5;
        # level 0 alone 1.01e-2 (the error of the inner levels passes through the outer up path undamped).
        # precision='fp16' (the binary16 build of the same kernels): default 0 -- the whole last step on the fp32-storage engine.
        # Its 49 steps leave 1.6e-3 L2 against the CPU oracle at 512x512 (bf16: 1.3e-2); the last step on levels 0-1 makes that
        # 1.06e-3, on all levels 6.3e-4 / 75.6 dB for +2.6 % time (tools/probes/fp16_drift.py; profiles/r06/fp16_mode.md).
        lv = os.environ.get("FOUNDDIFF_FINAL_OUTER_LEVELS")
        self.final_outer_levels = None if lv is None else int(lv)       # (None: the default of the model's precision, see the property)
        self.check_fp16_range = os.environ.get("FOUNDDIFF_FP16_CHECK", "1") != "0"
        # how many of those levels also run their DOWN stage on the tail engine (default: all of them)
        self.final_down_levels = int(os.environ.get("FOUNDDIFF_FINAL_DOWN_LEVELS", "-1"))
        # adaLN vectors of all DDIM steps in one pass in front of the captured loop (DAEngine.time_cond_table) instead of six
        # small launches per step: "auto" = with the one-slice kernel set only (151.4 -> 149.8 ms per 50-step slice at batch 1;
        # in the two-stream throughput mode the small launches of one stream hide under the other's kernels and the table
        # measured -0.3 %: 13.27 vs 13.21 slices/s, alternated); 1 / 0 force it on / off.  Bit for bit the same vectors.
        self._time_table = os.environ.get("FOUNDDIFF_TIME_TABLE", "auto")
        for k, v in residual_schedule(timesteps, after_init=False).items():
            self.register_buffer(k, v)
        self._host_sched = None
        self._slot = 0
        # independent half-batches on concurrent HIP streams: kernels bound by different resources (VALU-issue
        # scan, HBM row-GEMMs, MFMA convolutions) overlap when they come from independent launch sequences
        # (measured: 2 x 8 slices on two streams 1.79 ms per slice-forward, one stream of 8 or 16: 1.90 / 1.84)
        self.streams = int(os.environ.get("FOUNDDIFF_STREAMS", "2"))
        # largest sub-batch one engine is asked to hold (sample() runs bigger batches as consecutive groups): bounds the workspace
        self.max_sub_batch = int(os.environ.get("FOUNDDIFF_MAX_SUB_BATCH", "16"))
        self._side_streams = {}

    def init(self):
        """Re-derive the schedule the way Trainer.test() does before sampling (src/DADiff.py:1033)."""
        dev = self.betas.device
        for k, v in residual_schedule(1000, after_init=True).items():
            setattr(self, k, v.to(dev))
        self.num_timesteps = 1000
        self._host_sched = None
        self._drop_graphs()

    def load_state_dict(self, state_dict, strict=True, assign=False):
        live = {k: v for k, v in state_dict.items()
                if not (arch.is_dead_key(k, "model.unet0.") or arch.is_dead_key(k, "model.unet1."))}
        return super().load_state_dict(live, strict=strict, assign=assign)     # new weights -> new engines -> new graphs

    # ---- helpers
    def _drop_graphs(self):
        """Forget every captured graph (they bake in scheduler constants): init() re-derives the schedule."""
        for name in ("unet0", "unet1"):
            u = getattr(self.model, name, None)
            for e in (getattr(u, "_engine", None) or {}).values():
                e.graphs.clear()
                e.loop_graphs.clear()

    def _hs(self):
        if self._host_sched is None:
            self._host_sched = {k: getattr(self, k).detach().cpu() for k in
                                ("alphas_cumsum", "betas_cumsum", "posterior_mean_coef1", "posterior_mean_coef2",
                                 "posterior_mean_coef3", "posterior_log_variance_clipped")}
        return self._host_sched

    def _eng(self):
        return self.model.unet0.engine(slot=self._slot)

    def _unet(self, x_input, x, t_idx, out=None):
        """raw model output for batched integer timesteps t_idx (B,) (src/DADiff.py:1160-1164)"""
        time = (self.alphas_cumsum[t_idx] * self.num_timesteps).float().contiguous()
        return self._eng().forward(x, x_input, time, out=out)

    # ---- objectives other than the shipped 'pred_res' 
    def _plan(self):
        """(mode of fd_res_step_obj, run unet0?, run unet1?, unet0's time entry) for this objective /
        test_res_or_noise -- UnetRes.forward (src/DADiff.py:817-836) + model_predictions (1168-1207)."""
        if getattr(self.model, "num_unet", 1) == 2:
            tst = self.test_res_or_noise
            if self.objective == "pred_x0_noise":
                if tst != "res_noise":
                    raise ValueError("pred_x0_noise reads both model outputs: test_res_or_noise must be 'res_noise'")
                return 3, True, True, 0
            if self.objective != "pred_res_noise":
                raise ValueError(f"objective {self.objective!r} with num_unet=2")
            return {"res_noise": (2, True, True, 0), "res": (0, True, False, 0), "noise": (1, False, True, 0)}[tst]
        if self.objective == "pred_noise":
            return 1, True, False, 1            # the single UNet predicts the noise, fed with time[1]
        return 0, True, False, 0

    def _is_shipped(self):
        return self._plan() == (0, True, False, 0) and not self.input_condition

    def _engines(self):
        mode, r0, r1, _ = self._plan()
        return ([self.model.unet0.engine()] if r0 else []) + ([self.model.unet1.engine()] if r1 else [])

    def _encode_all(self, x_in):
        for e in self._engines():
            e.encode_condition(x_in)

    def _outputs(self, x_in, x, t_idx):
        """raw outputs (o0, o1) of the UNets this configuration evaluates (None where it does not)."""
        mode, r0, r1, tsel0 = self._plan()
        T = self.num_timesteps
        t0 = ((self.alphas_cumsum if tsel0 == 0 else self.betas_cumsum)[t_idx] * T).float().contiguous()
        t1 = (self.betas_cumsum[t_idx] * T).float().contiguous()
        o0 = o1 = None
        c2 = self._xc2 if self.input_condition else None
        if mode == 1 and r0:                    # single-UNet pred_noise: its output plays the role of o1
            o1 = self.model.unet0.engine().forward(x, x_in, t0, x_cond2=c2).clone()
        elif r0:
            o0 = self.model.unet0.engine().forward(x, x_in, t0, x_cond2=c2).clone()
        if r1:
            o1 = self.model.unet1.engine().forward(x, x_in, t1, x_cond2=c2).clone()
        return mode, o0, o1

    def _set_cond2(self, x_input_condition):
        """third input plane (src/DADiff.py:1157-1158); already normalised by the caller / sample()."""
        if self.input_condition:
            if not torch.is_tensor(x_input_condition):
                raise ValueError("input_condition=True needs x_input_condition (sample(x_input=[a, b]))")
            self._xc2 = x_input_condition.contiguous().float()

    def _par(self, t_idx, k=(0., 0., 0., 0.), flag=0.):
        B = t_idx.shape[0]
        par = torch.zeros(B, 8, device=t_idx.device, dtype=torch.float32)
        par[:, 0], par[:, 1], par[:, 2] = self.alphas_cumsum[t_idx], self.betas_cumsum[t_idx], self.one_minus_alphas_cumsum[t_idx]
        for i, v in enumerate(k):
            par[:, 3 + i] = v
        par[:, 7] = flag
        return par.contiguous()

    def _step_obj(self, step, x_in, x, t_idx, noise, k=(0., 0., 0., 0.), flag=0., want_preds=False, out=None):
        mode, o0, o1 = self._outputs(x_in, x, t_idx)
        par = self._par(t_idx, k, flag)
        pr = pn = xs = None
        if want_preds:
            pr, pn, xs = torch.empty_like(x), torch.empty_like(x), torch.empty_like(x)
        else:
            xs = torch.empty_like(x)
        img = (out if out is not None else torch.empty_like(x)) if step else None
        L.call("fd_res_step_obj", mode, step, _p(o0), _p(o1), _p(x), _p(x_in), _p(noise), _p(par), _p(pr), _p(pn),
               _p(xs), _p(img), x.shape[0], x[0].numel(), _stream(x))
        return img, pr, pn, xs

    def predict_noise_from_res(self, x_t, t, x_input, pred_res):
        B = x_t.shape[0]
        pn = torch.empty_like(x_t)
        ac, bc = self.alphas_cumsum[t].contiguous(), self.betas_cumsum[t].contiguous()
        # pred_res is already clamped by the caller; clamp is idempotent
        L.call("fd_res_predictions", _p(pred_res), _p(x_t), _p(x_input), _p(ac), _p(bc), None, _p(pn), None,
               B, x_t[0].numel(), _stream(x_t))
        return pn

    def q_posterior(self, pred_res, x_start, x_t, t):
        """src/DADiff.py:1142-1151.  Small host-side glue over per-batch coefficients."""
        e = lambda a: a[t].reshape(-1, 1, 1, 1)
        mean = torch.empty_like(x_t)
        coef = torch.stack([self.posterior_mean_coef1[t], self.posterior_mean_coef2[t],
                            self.posterior_mean_coef3[t], torch.full_like(self.posterior_mean_coef1[t], -1e30)], 1)
        # mean = c1 x_t + c2 pred_res + c3 x_start, evaluated by the posterior kernel with noise = NULL;
        # the kernel recomputes x_start = clamp(x_in - pred_res); to honour an arbitrary x_start we pass
        # x_in := x_start + pred_res (both already clamped by model_predictions).
        xin_equiv = torch.empty_like(x_t)
        L.call("fd_axpy_f32", _p(x_start.contiguous()), _p(pred_res.contiguous()), 1.0, _p(xin_equiv),
               x_t.numel(), _stream(x_t))
        L.call("fd_res_posterior_step", _p(pred_res.contiguous()), _p(x_t.contiguous()), _p(xin_equiv), None,
               _p(coef.contiguous()), _p(mean), None, x_t.shape[0], x_t[0].numel(), _stream(x_t))
        return mean, e(self.posterior_variance), e(self.posterior_log_variance_clipped)

    @torch.no_grad()
    def model_predictions(self, x_input, x, t, x_input_condition=0, x_self_cond=None, clip_denoised=True,
                          reuse_condition=False):
        """src/DADiff.py:1153-1209."""
        assert clip_denoised, "clip_denoised=False is not built"
        x_input = x_input.contiguous().float()
        x = x.contiguous().float()
        if not self._is_shipped():
            self._set_cond2(x_input_condition)
            if not reuse_condition:
                self._encode_all(x_input)
            _, pr, pn, xs = self._step_obj(0, x_input, x, t, None, want_preds=True)
            return ModelResPrediction(pr, pn, xs)
        if not reuse_condition:
            self._eng().encode_condition(x_input)
        mo = self._unet(x_input, x, t)
        pred_res, pred_noise, x_start = torch.empty_like(x), torch.empty_like(x), torch.empty_like(x)
        ac, bc = self.alphas_cumsum[t].contiguous(), self.betas_cumsum[t].contiguous()
        L.call("fd_res_predictions", _p(mo), _p(x), _p(x_input), _p(ac), _p(bc), _p(pred_res), _p(pred_noise),
               _p(x_start), x.shape[0], x[0].numel(), _stream(x))
        return ModelResPrediction(pred_res, pred_noise, x_start)

    def p_mean_variance(self, x_input, x, t, x_input_condition=0, x_self_cond=None):
        preds = self.model_predictions(x_input, x, t, x_input_condition, x_self_cond)
        mean, var, logvar = self.q_posterior(preds.pred_res, preds.pred_x_start, x, t)
        return mean, var, logvar, preds.pred_x_start

    @torch.no_grad()
    def p_sample(self, x_input, x, t: int, x_input_condition=0, x_self_cond=None, noise=None,
                 reuse_condition=False, out=None):
        """One ancestral step (src/DADiff.py:1222-1230): UNet forward + fused posterior update."""
        x_input = x_input.contiguous().float()
        x = x.contiguous().float()
        B = x.shape[0]
        hs = self._hs()
        t_idx = torch.full((B,), t, device=x.device, dtype=torch.long)
        if not self._is_shipped():
            if torch.is_tensor(x_input_condition):
                self._set_cond2(x_input_condition)
            if not reuse_condition:
                self._encode_all(x_input)
            if t > 0 and noise is None:
                noise = torch.randn_like(x)
            k = (float(hs["posterior_mean_coef1"][t]), float(hs["posterior_mean_coef2"][t]),
                 float(hs["posterior_mean_coef3"][t]), float(hs["posterior_log_variance_clipped"][t]))
            img, _, _, xs = self._step_obj(2, x_input, x, t_idx, noise if t > 0 else None, k=k, out=out)
            return img, xs
        if not reuse_condition:
            self._eng().encode_condition(x_input)
        mo = self._unet(x_input, x, t_idx)
        coef = torch.tensor([[hs["posterior_mean_coef1"][t], hs["posterior_mean_coef2"][t],
                              hs["posterior_mean_coef3"][t], hs["posterior_log_variance_clipped"][t]]] * B,
                            dtype=torch.float32).to(x.device)
        if t > 0 and noise is None:
            noise = torch.randn_like(x)
        pred_img = out if out is not None else torch.empty_like(x)
        x_start = torch.empty_like(x)
        L.call("fd_res_posterior_step", _p(mo), _p(x), _p(x_input), _p(noise) if t > 0 else None, _p(coef),
               _p(pred_img), _p(x_start), B, x[0].numel(), _stream(x))
        return pred_img, x_start

    @property
    def final_outer_levels(self):
        if self._final_outer_levels is not None:
            return self._final_outer_levels
        return 0 if self.model.unet0.kernel_precision == "fp16" else 2

    @final_outer_levels.setter
    def final_outer_levels(self, v):
        self._final_outer_levels = None if v is None else int(v)

    # ---- the per-step hot loop: graph-captured UNet forward + one scheduler kernel
    def _tail_engine(self, eng):
        """(K, engine or None): the higher-precision engine of the last K steps of a loop (final_fp32_steps):
        fp32 behind the bf16 kernels, bf16 behind the fp8-weight kernels (one precision class up; the fp8 mode is
        the throughput configuration, BASELINE configs[4])."""
        K = self.final_fp32_steps if eng.mode not in ("fp32", "fp32s") else 0
        if K <= 0:
            return 0, None
        # 'fp32s' = fp32 storage with split-bf16 contractions (engine.py): 2^-16 per product is far below the bf16
        # error of the 49 steps before it, at a third of the exact-f32 MFMA's time; FOUNDDIFF_TAIL_EXACT=1 restores 'fp32'
        tail32 = "fp32" if os.environ.get("FOUNDDIFF_TAIL_EXACT") else "fp32s"
        e32 = self.model.unet0.engine(tail32 if eng.mode in ("bf16", "fp16") else "bf16", slot=self._slot)
        e32.share_condition(eng)
        return K, e32

    def _tail_forward(self, e32, eng, img, x_in, time_buf, mo, sched=None):
        """The forward of a tail step: hybrid (outer levels on e32, the rest on eng) or all of it on e32."""
        k = self.final_outer_levels
        if k > 0 and k < len(e32.downs):
            e32.forward_hybrid(eng, img, x_in, time_buf, out=mo, outer_levels=k, sched=sched,
                               down_levels=None if self.final_down_levels < 0 else self.final_down_levels)
        else:
            e32.forward(img, x_in, time_buf, out=mo, sched=sched)

    def _step_forward(self, x_in, img, time_buf, mo, eng=None, tail_of=None):
        eng = eng or self._eng()
        run = (lambda: self._tail_forward(eng, tail_of, img, x_in, time_buf, mo)) if tail_of is not None else \
            (lambda: eng.forward(img, x_in, time_buf, out=mo))
        key = (tuple(img.shape), eng.mode, eng.gen, tail_of.gen if tail_of is not None else 0, self.final_outer_levels, self.final_down_levels)
        if not self.use_graph:
            run()
            return
        cache = eng.graphs                 # captured graphs live and die with the engine whose buffers they bake in
        ent = cache.get(key)
        if ent is None:
            # warm-up (allocates every workspace buffer), then capture
            run()
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                run()
            cache.clear()                                              # one step graph per engine
            ent = cache[key] = (g, (x_in, img, time_buf, mo))
        g, (gx, gi, gt, gm) = ent
        if gx.data_ptr() != x_in.data_ptr():
            gx.copy_(x_in)
        if gi.data_ptr() != img.data_ptr():
            gi.copy_(img)
        if gt.data_ptr() != time_buf.data_ptr():
            gt.copy_(time_buf)
        g.replay()
        if gm.data_ptr() != mo.data_ptr():
            mo.copy_(gm)

    def _loop_buffers(self, x_input, shape):
        eng = self._eng()
        dev = x_input.device
        f = dict(device=dev, dtype=torch.float32)
        x_in = eng._b("loop_xin", shape, torch.float32)
        x_in.copy_(x_input)
        img = eng._b("loop_img", shape, torch.float32)
        mo = eng._b("model_out", shape, torch.float32)
        time_buf = eng._b("loop_time", (shape[0],), torch.float32)
        return x_in, img, mo, time_buf

    X_T_STEP = 0x7FFFFFFF      # the "step" under which the keyed stream draws x_T (fd_keyed_normal)

    def _slice_seeds(self, slice_seeds, B, dev):
        """(B,) int64 device tensor of per-slice seeds: given, or drawn from torch's global CPU generator (so
        torch.manual_seed makes a run reproducible, like the reference's set_seed)."""
        if slice_seeds is None:
            slice_seeds = torch.randint(0, 2 ** 62, (B,), dtype=torch.int64)
        slice_seeds = torch.as_tensor(slice_seeds, dtype=torch.int64).reshape(-1)
        if slice_seeds.numel() != B:
            raise ValueError(f"slice_seeds must hold one seed per slice ({B}), got {slice_seeds.numel()}")
        return slice_seeds.to(dev).contiguous()

    def _keyed_noise(self, seeds, t, shape):
        nz = torch.empty(shape, device=seeds.device, dtype=torch.float32)
        L.call("fd_keyed_normal", _p(seeds), int(t), _p(nz), shape[0], nz[0].numel(), _stream(nz))
        return nz

    @torch.no_grad()
    def p_sample_loop(self, x_input, shape, last=True, noise=None, step_noise=None, slice_seeds=None):
        """src/DADiff.py:1233-1273.  `noise`: the initial randn(shape); `step_noise`: callable t -> tensor (parity
        tests feed the reference's noise this way; one eager step at a time).  Without `step_noise` the step noise is
        the per-slice keyed stream of fd_sched.hip -- a function of (slice_seeds[b], t, pixel) only, so a slice's
        result does not depend on its batch, rank or stream -- and the loop runs from device tables: chunks of steps
        are captured in a HIP graph and replayed (no host work per step)."""
        if self.input_condition:
            self._set_cond2(x_input[1])
        x_input = x_input[0].contiguous().float()
        if not self._is_shipped():
            return self._generic_loop(x_input, shape, last, noise, step_noise, ddim=False)
        x_in, img, mo, time_buf = self._loop_buffers(x_input, shape)
        eng = self._eng()
        eng.encode_condition(x_in)
        seeds = None
        if step_noise is None:
            seeds = self._slice_seeds(slice_seeds, shape[0], x_in.device)
        if noise is None:
            noise = self._keyed_noise(seeds, self.X_T_STEP, shape) if seeds is not None else torch.randn(shape, device=x_in.device)
        L.call("fd_axpy_f32", _p(x_in), _p(noise.contiguous()), math.sqrt(self.sum_scale), _p(img), img.numel(),
               _stream(img))
        input_add_noise = img.clone()
        hs = self._hs()
        T = self.num_timesteps
        B = shape[0]
        coefs = torch.stack([hs["posterior_mean_coef1"], hs["posterior_mean_coef2"], hs["posterior_mean_coef3"],
                             hs["posterior_log_variance_clipped"]], 1).float().contiguous().to(x_in.device)      # (T,4)
        times = (hs["alphas_cumsum"] * T).float().contiguous().to(x_in.device)
        img_list = []
        K, e32 = self._tail_engine(eng)
        if seeds is not None:
            self._ancestral_keyed(eng, e32, K, x_in, img, mo, time_buf, coefs, times, seeds, last, img_list)
        else:
            for t in reversed(range(0, T)):
                time_buf.fill_(float(times[t]))
                if t < K:
                    self._step_forward(x_in, img, time_buf, mo, e32, tail_of=eng)
                else:
                    self._step_forward(x_in, img, time_buf, mo, eng)
                coef = coefs[t:t + 1].expand(B, 4).contiguous()
                nz = step_noise(t) if t > 0 else None
                L.call("fd_res_posterior_step", _p(mo), _p(img), _p(x_in), _p(nz), _p(coef), _p(img), None, B,
                       img[0].numel(), _stream(img))
                if not last:
                    img_list.append(img.clone())
        if not last:
            img_list = [input_add_noise] + img_list
        else:
            img_list = [input_add_noise, img.clone()]
        return unnormalize_to_zero_to_one(img_list)

    def _ancestral_keyed(self, eng, e32, K, x_in, img, mo, time_buf, coefs, times, seeds, last, img_list):
        """The T ancestral steps with the timestep in a device counter: fd_ancestral_begin (t <- t - 1, UNet time
        input) -> UNet forward -> fd_res_posterior_step_keyed (coefficient row t, keyed noise).  With last=True the
        T - K main steps run as replays of one captured chunk of G steps and the K tail steps (higher-precision
        engine) as a second graph."""
        T, B = self.num_timesteps, x_in.shape[0]
        npix = img[0].numel()
        t_dev = eng._b("loop_t", (1,), torch.int32)
        t_dev.fill_(T)
        # graphs bake the pointers of coefs / times / seeds: keep them in engine-owned buffers
        gco, gti, gse = eng._b("loop_coefs", tuple(coefs.shape), torch.float32), eng._b("loop_times", (T,), torch.float32), \
            eng._b("loop_seeds", (B,), torch.int64)
        gco.copy_(coefs)
        gti.copy_(times)
        gse.copy_(seeds)

        def one(e):
            L.call("fd_ancestral_begin", _p(t_dev), _p(gti), _p(time_buf), B, _stream(img))
            if e is eng:
                e.forward(img, x_in, time_buf, out=mo)
            else:
                self._tail_forward(e, eng, img, x_in, time_buf, mo)
            L.call("fd_res_posterior_step_keyed", _p(mo), _p(img), _p(x_in), _p(gco), _p(t_dev), _p(gse), _p(img), None,
                   B, npix, _stream(img))

        K = max(0, min(int(K), T))                                # a tail longer than the loop is the whole loop
        n_main = T - K
        if not (self.use_graph and last):
            for i in range(T):
                one(eng if i < n_main else e32)
                if not last:
                    img_list.append(img.clone())
            return
        # G steps per captured chunk: the largest divisor of n_main in [8, 64]; when there is none (n_main prime or
        # small) a fixed 40 with the n_main % 40 leftover steps run eagerly in front of the replays
        G = max([g for g in range(8, 65) if n_main > 0 and n_main % g == 0] or [40])
        graphs = eng.__dict__.setdefault("anc_graphs", {})
        key = (tuple(img.shape), eng.mode, eng.gen, G, K, e32.gen if e32 else 0, self.final_outer_levels, self.final_down_levels, T)
        reps, rem = n_main // G, n_main % G
        if key not in graphs:
            start, t0 = img.clone(), t_dev.clone()
            eng.forward(img, x_in, time_buf, out=mo)              # warm-up: every workspace buffer exists
            if e32 is not None:
                self._tail_forward(e32, eng, img, x_in, time_buf, mo)
            torch.cuda.synchronize()
            gm = None
            if reps > 0:
                gm = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gm):
                    for _ in range(G):
                        one(eng)
            gt = None
            if K > 0:
                gt = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gt):
                    for _ in range(K):
                        one(e32)
            graphs.clear()
            graphs[key] = (gm, gt)
            img.copy_(start)
            t_dev.copy_(t0)
        gm, gt = graphs[key]
        lim = getattr(self, "_anc_max_chunks", None)          # bench.py's bounded leg: time a few chunks, not the volume
        if lim:
            reps, rem = min(reps, int(lim)), 0
        for _ in range(rem):
            one(eng)
        for _ in range(reps):
            gm.replay()
        if gt is not None:
            if lim:
                t_dev.fill_(K)                                # jump to the tail steps
            gt.replay()
        self._anc_steps_run = rem + reps * G + K

    @property
    def time_table(self):
        if self._time_table in ("0", "1"):
            return self._time_table == "1"
        return bool(getattr(self._eng(), "low_latency", False))

    @torch.no_grad()
    def ddim_sample(self, x_input, shape, last=True, noise=None):
        """src/DADiff.py:1276-1365 (eta = 0, type 'use_pred_noise')."""
        if self.input_condition:
            self._set_cond2(x_input[1])
        x_input = x_input[0].contiguous().float()
        if not self._is_shipped():
            return self._generic_loop(x_input, shape, last, noise, None, ddim=True)
        x_in, img, mo, time_buf = self._loop_buffers(x_input, shape)
        eng = self._eng()
        eng.encode_condition(x_in)
        T, S = self.num_timesteps, self.sampling_timesteps
        times = torch.linspace(-1, T - 1, steps=S + 1)
        times = list(reversed(times.int().tolist()))
        time_pairs = list(zip(times[:-1], times[1:]))
        if noise is None:
            noise = torch.randn(shape, device=x_in.device)
        L.call("fd_axpy_f32", _p(x_in), _p(noise.contiguous()), math.sqrt(self.sum_scale), _p(img), img.numel(),
               _stream(img))
        input_add_noise = img.clone()
        hs = self._hs()
        acs = hs["alphas_cumsum"]
        img_list = []

        K, e32 = self._tail_engine(eng)

        def run_steps(forward, fold=False):
            """fold: the forward applies the DDIM update itself (DAEngine.forward sched=: in the bf16 mode inside the
            epilogue of its last kernel), with the step's constants baked into the captured loop graph; the adaLN vectors
            of all steps come from ONE pass in front of the loop (DAEngine.time_cond_table) instead of six small launches
            per step."""
            if fold and self.time_table:
                eng.time_cond_table()
            for i, (time, time_next) in enumerate(time_pairs):
                lastf = time_next < 0
                alpha = 0.0 if lastf else float(acs[time] - acs[time_next])
                if fold and self.time_table and i < S - K:
                    forward(eng, (alpha, lastf), i)
                    continue
                time_buf.fill_(float(acs[time] * T))
                if fold:
                    forward(e32 if i >= S - K else eng, (alpha, lastf))
                    continue
                forward(e32 if i >= S - K else eng)
                L.call("fd_res_ddim_step", _p(mo), _p(img), _p(x_in), None, alpha, 0.0, int(lastf), _p(img),
                       img.numel(), _stream(img))
                if not last:
                    img_list.append(img.clone())

        if self.use_graph and last and os.environ.get("FOUNDDIFF_LOOP_GRAPH", "1") != "0":
            # the whole S-step loop as ONE HIP graph (S x (time fill + 141 kernels + DDIM update), every
            # scheduler constant baked into its node): replayed per sample() on the persistent loop buffers
            key = ("ddim", tuple(shape), eng.mode, eng.gen, S, T, K, e32.gen if e32 else 0, self.final_outer_levels, self.final_down_levels,
                   self.time_table)

            def fwd(e, sched=None, step=None):
                if e is eng:
                    e.forward(img, x_in, time_buf, out=mo, sched=sched, step=step)
                else:
                    self._tail_forward(e, eng, img, x_in, time_buf, mo, sched=sched)
            loops = eng.loop_graphs
            if key not in loops:
                start = img.clone()
                if self.time_table:                                # (host -> device copy and buffers: before the capture)
                    eng.time_table_prepare([acs[t] * T for t, _ in time_pairs], shape[0])
                    eng.time_cond_table()
                for e in (eng, e32):
                    if e is not None:
                        fwd(e)                                    # warm-up: every workspace buffer exists
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    run_steps(fwd, fold=True)
                loops.clear()                                      # one loop graph per engine
                loops[key] = g
                img.copy_(start)                                   # capture does not execute: restore x_T
            loops[key].replay()
        else:
            run_steps(lambda e: self._step_forward(x_in, img, time_buf, mo, e, tail_of=None if e is eng else eng))
        if not last:
            img_list = [input_add_noise] + img_list
        else:
            img_list = [input_add_noise, img.clone()]
        return unnormalize_to_zero_to_one(img_list)

    def _generic_loop(self, x_in, shape, last, noise, step_noise, ddim):
        """p_sample_loop / ddim_sample for the objectives other than 'pred_res' and for the dual-UNet
        model: eager UNet forwards (one or two per step) + one fd_res_step_obj launch per step."""
        self._encode_all(x_in)
        B = shape[0]
        if noise is None:
            noise = torch.randn(shape, device=x_in.device)
        img = torch.empty_like(x_in)
        L.call("fd_axpy_f32", _p(x_in), _p(noise.contiguous()), math.sqrt(self.sum_scale), _p(img), img.numel(),
               _stream(img))
        input_add_noise = img.clone()
        hs = self._hs()
        T = self.num_timesteps
        img_list = []
        if ddim:
            times = list(reversed(torch.linspace(-1, T - 1, steps=self.sampling_timesteps + 1).int().tolist()))
            acs = hs["alphas_cumsum"]
            for time, time_next in zip(times[:-1], times[1:]):
                t_idx = torch.full((B,), time, device=x_in.device, dtype=torch.long)
                lastf = time_next < 0
                alpha = 0.0 if lastf else float(acs[time] - acs[time_next])
                img = self._step_obj(1, x_in, img, t_idx, None, k=(alpha, 0., 0., 0.), flag=float(lastf))[0]
                if not last:
                    img_list.append(img.clone())
        else:
            for t in reversed(range(0, T)):
                nz = None
                if t > 0:
                    nz = step_noise(t) if step_noise is not None else torch.randn(shape, device=x_in.device)
                img, _ = self.p_sample(x_in, img, t, noise=nz, reuse_condition=True)
                if not last:
                    img_list.append(img.clone())
        img_list = [input_add_noise] + img_list if not last else [input_add_noise, img.clone()]
        return unnormalize_to_zero_to_one(img_list)

    @torch.no_grad()
    def sample(self, x_input=0, batch_size=16, last=True, noise=None, step_noise=None, slice_seeds=None):
        """src/DADiff.py:1368-1380: x_input = [ldct (B,1,H,W) in [0,1]] -> list of images in ~[0,1].
        `slice_seeds` (B int64): per-slice keys of the ancestral sampler's step noise (and of x_T when `noise` is not
        given) -- see p_sample_loop; founddiff_amd.parallel.sample_volume passes seed + GLOBAL slice index."""
        x_input = list(x_input)
        res = self._sample(x_input, batch_size, last, noise, step_noise, slice_seeds)
        if self.model.unet0.kernel_precision == "fp16" and self.check_fp16_range:
            # binary16 ends at 65504: an activation beyond it is stored as infinity and reaches the image as NaN (the clamps of
            # the scheduler kernels propagate NaN like torch.clamp).  One reduction + one host read per sample() call.
            if not bool(torch.isfinite(res[-1]).all()):
                if self.model.unet0.precision != "auto":
                    raise L.FoundDiffHipError(
                        "precision='fp16': the sampled image is not finite -- an activation of this checkpoint left IEEE binary16's "
                        "range (65504).  Use precision='bf16' (same speed, fp32's range, 8 significand bits), 'auto' or 'fp32s'.")
                import warnings
                warnings.warn("founddiff_amd: precision='auto': this checkpoint's activations leave IEEE binary16's range; "
                              "this and every later sample() of the model run on the bfloat16 kernels")
                for u in (getattr(self.model, "unet0", None), getattr(self.model, "unet1", None)):
                    if u is not None:
                        u._auto_bf16 = True
                res = self._sample(x_input, batch_size, last, noise, step_noise, slice_seeds)
        return res

    def _sample(self, x_input, batch_size, last, noise, step_noise, slice_seeds):
        x_input = list(x_input)
        # Workspace bound: an engine's workspace grows with its sub-batch (~1.9 GB per 512x512 slice in bf16,
        # twice that in the fp32 modes) and a sample() keeps `streams` engines plus their tail engines alive.  Batches beyond
        # streams x max_sub_batch slices run as consecutive groups of that size on the same engines: a slice's result does not
        # depend on its batch, so the output is the same bits; the workspace stays that of one group whatever the batch.
        grp = self.streams * self.max_sub_batch
        if (self._is_shipped() and last and step_noise is None and not self.input_condition and x_input[0].shape[0] > grp
                and x_input[0].shape[0] % self.streams == 0):
            dev = x_input[0].device
            batch_size = x_input[0].shape[0]
            size = tuple(x_input[0].shape)
            if noise is None:
                noise = (torch.randn(size, device=dev) if (self.is_ddim_sampling and slice_seeds is None) else
                         self._keyed_noise(self._slice_seeds(slice_seeds, batch_size, dev), self.X_T_STEP, size))
            if not self.is_ddim_sampling:
                slice_seeds = self._slice_seeds(slice_seeds, batch_size, dev)
            x01 = x_input[0]                             # (still the caller's [0, 1] tensor: the groups normalise their own slices)
            parts = []
            for g0 in range(0, batch_size, grp):
                sl = slice(g0, min(g0 + grp, batch_size))
                parts.append(self._sample([x01[sl]], sl.stop - sl.start, True, noise[sl].contiguous(), None,
                                          None if slice_seeds is None else slice_seeds[sl]))
            return [torch.cat([p[j] for p in parts], 0) for j in range(len(parts[0]))]
        if self.input_condition and self.input_condition_mask:     # src/DADiff.py:1372-1375
            x_input[0] = normalize_to_neg_one_to_one(x_input[0])
        else:
            x_input = normalize_to_neg_one_to_one(x_input)
        batch_size, channels, h, w = x_input[0].shape
        size = (batch_size, channels, h, w)
        # two concurrent sub-batches: DDIM, and the ancestral sampler when its step noise is the keyed stream
        nsl = self.streams if (self._is_shipped() and last and batch_size >= 8 and batch_size % self.streams == 0 and
                               (self.is_ddim_sampling or step_noise is None)) else 1
        if nsl > 1:
            seeds = None
            if not self.is_ddim_sampling:
                seeds = self._slice_seeds(slice_seeds, batch_size, x_input[0].device)
                if noise is None:
                    noise = self._keyed_noise(seeds, self.X_T_STEP, size)
            return self._sample_concurrent(x_input[0], size, noise, nsl, seeds)
        if self.is_ddim_sampling:
            if noise is None and slice_seeds is not None:
                noise = self._keyed_noise(self._slice_seeds(slice_seeds, batch_size, x_input[0].device), self.X_T_STEP, size)
            return self.ddim_sample(x_input, size, last=last, noise=noise)
        return self.p_sample_loop(x_input, size, last=last, noise=noise, step_noise=step_noise, slice_seeds=slice_seeds)

    def _sample_concurrent(self, x_in, size, noise, nsl, seeds=None):
        """Sampling of a batch (DDIM; ancestral with keyed step noise: `seeds`) as `nsl` independent sub-batches, each on its own HIP stream with its own engine
        (workspaces, captured loop graph).  A slice's result does not depend on what else is in its batch (kernel
        configurations depend on the image size only), so the output is bit-identical to the single-stream run."""
        B = size[0]
        per = B // nsl
        main = torch.cuda.current_stream(x_in.device)
        if noise is None:
            noise = torch.randn(size, device=x_in.device)
        outs = []
        for k in range(nsl):
            st = self._side_streams.get(k)
            if st is None:
                st = self._side_streams[k] = torch.cuda.Stream(device=x_in.device)
            st.wait_stream(main)
            self._slot = k
            try:
                with torch.cuda.stream(st):
                    sl = slice(k * per, (k + 1) * per)
                    if seeds is None:
                        o = self.ddim_sample([x_in[sl].contiguous()], (per,) + tuple(size[1:]), last=True,
                                             noise=noise[sl].contiguous())
                    else:
                        o = self.p_sample_loop([x_in[sl].contiguous()], (per,) + tuple(size[1:]), last=True,
                                               noise=noise[sl].contiguous(), slice_seeds=seeds[sl])
            finally:
                self._slot = 0
            outs.append((st, o))
        res = []
        for st, o in outs:
            main.wait_stream(st)
            for t in o:
                t.record_stream(main)
        for j in range(len(outs[0][1])):
            res.append(torch.cat([o[j] for _, o in outs], 0))
        return res

    def forward(self, *a, **k):
        raise NotImplementedError("training (p_losses) is out of scope: founddiff_amd is a sampling engine")


class _Accel:
    """Stand-in for the accelerate attributes train.py reads (train.py:162-177)."""
    is_local_main_process = True
    is_main_process = True

    def __init__(self, device):
        self.device = device

    def wait_for_everyone(self):
        pass


class _EMAView:
    """`trainer.ema.ema_model` of the reference (ema_pytorch wrapper around the diffusion)."""

    def __init__(self, model):
        self.ema_model = model

    def to(self, device):
        self.ema_model.to(device)
        return self

    def load_state_dict(self, sd, strict=True):
        live = {k[len("ema_model."):]: v for k, v in sd.items() if k.startswith("ema_model.")}
        if strict:
            return load_weights(self.ema_model, live, "ema state")
        return self.ema_model.load_state_dict(live, strict=False)


class Trainer(object):
    """Sampling harness with the reference Trainer's surface (src/DADiff.py:1506-1966): `load`,
    `sample`, `test`, `.accelerator.is_local_main_process`, `.results_folder`, `.train_logger`.
    Training (`train`, `save`) is out of scope.  Unlike the reference (whose datasets glob
    private paths inside the class), the evaluation dataset is passed in: any object with
    `__len__`, `__getitem__ -> [ndct, ldct]` ((1,H,W) tensors in [0,1]) and `load_name(i)`."""

    def __init__(self, opt, diffusion_model, folder=None, *, train_batch_size=16, gradient_accumulate_every=1,
                 augment_flip=True, train_lr=1e-4, train_num_steps=100000, ema_update_every=10, ema_decay=0.995,
                 adam_betas=(0.9, 0.99), save_and_sample_every=1000, num_samples=25, results_folder=".results/sample",
                 amp=False, fp16=False, split_batches=True, convert_image_to=None, condition=False, sub_dir=False,
                 equalizeHist=False, crop_patch=False, generation=False, num_unet=2, checkpoint_folder=None,
                 is_train=True, train_logger=None, dataset=None, device=None):
        import logging
        import os as _os
        self.opt = opt
        self.checkpoint_folder = checkpoint_folder or "."
        self.results_folder = self.checkpoint_folder + "/sample"
        _os.makedirs(self.results_folder, exist_ok=True)
        assert int(math.sqrt(num_samples)) ** 2 == num_samples, "number of samples must have an integer square root"
        self.num_samples = num_samples
        self.condition = condition
        self.num_unet = num_unet
        self.sub_dir, self.crop_patch = sub_dir, crop_patch
        self.image_size = diffusion_model.image_size
        self.device = torch.device(device or ("cuda" if torch.cuda.is_available() else "cpu"))
        self.accelerator = _Accel(self.device)
        self.model = diffusion_model.to(self.device)
        self.ema = _EMAView(self.model)      # inference uses the EMA weights (src/DADiff.py:1818-1822)
        self.sample_dataset = dataset if dataset is not None else folder
        self.train_logger = train_logger or logging.getLogger("founddiff_amd")
        self.step = 0
        self.condition_type = 2
        self.test_running_psnr, self.test_running_ssim, self.test_running_rmse = [], [], []

    def load(self, milestone):
        """Read `<ckpt>/sample/model-N.pt` = {'step','model','opt0','ema','scaler'}
        (src/DADiff.py:1648-1669).  The EMA weights win when present; a missing file is skipped
        silently like the reference does."""
        from pathlib import Path
        path = Path(self.results_folder + "/" + f"model-{milestone}.pt")
        if not path.exists():
            return
        data = torch.load(str(path), map_location="cpu", weights_only=False)
        load_weights(self.model, data["model"], f"{path}['model']")
        self.step = data.get("step", 0)
        ema = data.get("ema")
        if ema:
            live = {k[len("ema_model."):]: v for k, v in ema.items() if k.startswith("ema_model.")}
            if live:
                load_weights(self.model, live, f"{path}['ema']")
        self.model.to(self.device)
        print("load model - " + str(path))

    def train(self):
        raise NotImplementedError("founddiff_amd is a sampling engine: training is out of scope")

    save = train

    # anatomy groups of the reference's 2020 test list, in item order: (name, slices per dose level); each holds
    # 4 dose levels back to back (src/DADiff.py:1918-1950).  "head" is taken from the END of the list, as there.
    test_groups = (("ab", 290), ("lung", 637), ("head", 159))

    @torch.no_grad()
    def sample(self, milestone, last=True, FID=False):
        """Preview grid of [NDCT, LDCT, x_T, output] in the HU display window, `<results>/sample-N.png`
        (src/DADiff.py:1765-1815); with FID, one PNG per generated image, `milestone` advancing."""
        from .data import hu_window, save_image
        n = min(self.num_samples, len(self.sample_dataset))
        items = [self.sample_dataset[i] for i in range(n)]
        show = [torch.stack([it[j] for it in items]).to(self.device) for j in range(len(items[0]))]
        outs = list(self.model.sample(show[1:], batch_size=n, last=last))
        all_images_list = show + outs
        all_images = hu_window(torch.cat(all_images_list, dim=0))
        nrow = int(math.sqrt(self.num_samples)) if last else all_images.shape[0]
        if FID:
            for i in range(n):
                file_name = f"sample-{milestone}.png"
                save_image(all_images_list[0][i].unsqueeze(0), os.path.join(self.results_folder, file_name), nrow=1)
                milestone += 1
                if milestone >= getattr(self, "total_n_samples", 50000):
                    break
        else:
            file_name = f"sample-{milestone}.png"
            save_image(all_images, self.results_folder + "/" + file_name, nrow=nrow)
        print("sampe-save " + file_name)
        return milestone

    @staticmethod
    def image_file_name(file_name):
        """PNG name of a result (src/DADiff.py:1904-1907): quarter-dose files keep the part before the first
        dot, simulated-dose files (`...-0.25-0012.npy`) keep the dose fraction's dot."""
        if "quarter" in file_name:
            return file_name.split(".")[0] + ".png"
        return file_name.split(".")[0] + "." + file_name.split(".")[1] + ".png"

    def _log_groups(self):
        """Per-anatomy and per-dose means of the running metrics (src/DADiff.py:1918-1950), same slicing rule
        and log lines; groups the evaluated list does not reach log nan, as np.mean of an empty slice does."""
        def mean(v):
            return float(np.mean(v)) if len(v) else float("nan")
        P, S, R = self.test_running_psnr, self.test_running_ssim, self.test_running_rmse
        out, off = {}, 0
        for gi, (name, length) in enumerate(self.test_groups):
            if gi == len(self.test_groups) - 1:
                sl = slice(-length * 4, None)
            else:
                sl = slice(off, off + length * 4)
            gp, gs, gr = P[sl], S[sl], R[sl]
            self.train_logger.info("(%s average mean: psnr: %.4f, ssim: %.4f,rmse: %.4f)" % (name, mean(gp), mean(gs), mean(gr)))
            out[name] = {"mean": (mean(gp), mean(gs), mean(gr)), "dose": []}
            for i in range(4):
                d = slice(int(i * length), int((i + 1) * length))
                row = (mean(gp[d]), mean(gs[d]), mean(gr[d]))
                out[name]["dose"].append(row)
                self.train_logger.info("(%s\u2014\u2014\u2014\u2014dose: %2d,average: psnr: %.4f, ssim: %.4f,rmse: %.4f)" % ((name, i) + row))
            off += length * 4
        return out

    @torch.no_grad()
    def test(self, sample=False, last=True, FID=False, batch_size=1):
        """Evaluation loop of src/DADiff.py:1817-1966: init() the schedule, denoise every slice,
        PSNR/SSIM/RMSE vs NDCT (on device), np.save each output in [0,1], per-anatomy / per-dose means.
        `batch_size` > 1 batches independent slices (the reference uses 1).  `sample=True` keeps the
        reference's branch that returns inputs + outputs without metrics; without `condition` the loop is the
        reference's unconditional `self.sample` rounds (100, or up to 50000 images with FID)."""
        # batch_size 1 is the reference's loop: one slice per sample() call -> the kernel set for a lone slice (DAEngine
        # low_latency: same arithmetic, chunked scans at every level; outputs differ from a batched run by fp32
        # summation order).  The switch is only ever turned ON here (FOUNDDIFF_LOW_LATENCY=1 stays in force for any batch
        # size) and is restored on the way out, so later sample() calls on the same model are not affected.  Cost: the
        # low-latency engine is a second DAEngine (its own weight copy, workspaces and graphs) next to the default one.
        unets = [u for u in (getattr(getattr(self.model, "model", None), n, None) for n in ("unet0", "unet1"))
                 if u is not None and hasattr(u, "low_latency")]
        saved = [u.low_latency for u in unets]
        try:
            if batch_size == 1:
                for u in unets:
                    u.low_latency = True
            return self._test(sample, last, FID, batch_size)
        finally:
            for u, v in zip(unets, saved):
                u.low_latency = v

    def _test(self, sample, last, FID, batch_size):
        from .metrics import compute_metrics
        self.model.init()
        print("test start")
        if not self.condition:
            if FID:
                import glob
                self.total_n_samples = 50000
                img_id = len(glob.glob(f"{self.results_folder}/*"))
                n_rounds = (self.total_n_samples - img_id) // self.num_samples + 1
            else:
                n_rounds = 100
            for i in range(n_rounds):
                if FID:
                    i = img_id
                img_id = self.sample(i, last=last, FID=FID)
            print("test end")
            return None
        self.test_running_psnr, self.test_running_ssim, self.test_running_rmse = [], [], []
        self.test_image_names = []
        ds = self.sample_dataset
        for s in range(0, len(ds), batch_size):
            idx = list(range(s, min(s + batch_size, len(ds))))
            items = [ds[i] for i in idx]
            y = torch.stack([it[0] for it in items]).to(self.device)
            xs = [torch.stack([it[k] for it in items]).to(self.device) for k in range(1, len(items[0]))]
            if sample:
                all_images_list = [y] + xs + list(self.model.sample(xs, batch_size=len(idx)))
                y_pred = all_images_list[-1]
            else:
                y_pred = list(self.model.sample(xs, batch_size=len(idx), last=last))[-1]
                # src/DADiff.py:1872-1886: the metrics use the UNCROPPED prediction; crop_patch only crops what is saved
                m = compute_metrics(y_pred, y).cpu().numpy()
            y_save = y_pred
            if self.crop_patch and not sample:
                # the reference calls get_pad_size with its 1-based item counter (1826-1838, 1875): the last item would
                # index one past the end of a 0-based table, so the index is clamped
                pad = [tuple(ds.get_pad_size(min(i + 1, len(ds) - 1))) for i in idx]
                if len(set(pad)) != 1:
                    raise ValueError("crop_patch needs one pad size per batch (use batch_size=1)")
                h, w = y_pred.shape[-2:]
                y_save = y_pred[:, :, 0:h - pad[0][0], 0:w - pad[0][1]]
            for j, i in enumerate(idx):
                file_name = ds.load_name(i, sub_dir=self.sub_dir)
                self.test_image_names.append(self.image_file_name(file_name))
                if not sample:
                    self.test_running_psnr.append(m[j, 0])
                    self.test_running_ssim.append(m[j, 1])
                    self.test_running_rmse.append(m[j, 2])
                    print("(psnr: %.4f, ssim: %.4f,rmse:.%.4f) " % (m[j, 0], m[j, 1], m[j, 2]))
                if not getattr(self.opt, "is_train", False) and not sample:
                    h, w = y_save.shape[-2:]
                    np.save(self.results_folder + "/" + file_name[:-4], y_save[j].detach().cpu().numpy().reshape(h, w))
                    print("test-save " + file_name)
        if sample:
            print("test end")
            return None
        self.test_group_means = self._log_groups()
        self.train_logger.info("test_psnr: {:.4f}, test_ssim: {:.4f},test_rmse:{:.4f}".format(
            np.mean(self.test_running_psnr), np.mean(self.test_running_ssim), np.mean(self.test_running_rmse)))
        print("test end")
        return float(np.mean(self.test_running_psnr)), float(np.mean(self.test_running_ssim)), \
            float(np.mean(self.test_running_rmse))
