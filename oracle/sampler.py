"""ORACLE (test infrastructure, not product code): the reverse-diffusion loops on CPU.

Restates, with every random draw passed in explicitly (CPU and GPU generators differ):
  ResidualDiffusion.model_predictions / q_posterior / p_sample / p_sample_loop / ddim_sample / sample
      /root/reference/src/DADiff.py:1120-1380  (condition=True, eta=0; every objective of 1168-1207 and
      the single / dual UNet dispatch of UnetRes.forward 817-836)
  GaussianDiffusion.model_predictions / p_sample / p_sample_loop / ddim_sample
      /root/reference/src/denoising_diffusion_pytorch.py:523-652
Pinned by tests/golden/e2e_*.npz captured from the reference with its torch.randn patched to
replay the same tensors (tests/golden/make_golden.py).
"""
import math

import torch

from . import nets, schedule


class ResidualOracle:
    def __init__(self, sd, prefix="model.unet0.", timesteps=1000, sampling_timesteps=None,
                 sum_scale=0.01, after_init=True, scan_fn=None, hoist_cond=True, objective="pred_res",
                 test_res_or_noise="res", num_unet=1, prefix1="model.unet1.", input_condition=False,
                 input_condition_mask=False):
        self.sd = nets.SD(sd, prefix)
        self.sd1 = nets.SD(sd, prefix1) if num_unet == 2 else None
        self.objective, self.test, self.num_unet = objective, test_res_or_noise, num_unet
        self._cond1 = None
        self.input_condition, self.input_condition_mask = input_condition, input_condition_mask
        self.x_cond2 = None           # third input plane (src/DADiff.py:1157-1158), set by sample() / the caller
        self.T = timesteps
        self.S = sampling_timesteps if sampling_timesteps is not None else timesteps
        self.sum_scale = sum_scale
        self.sch = schedule.residual_schedule(timesteps, after_init)
        self.scan_fn = scan_fn
        self.hoist_cond = hoist_cond  # the DA-CLIP branch is t-independent
        self._cond = None

    def unet(self, x_t, x_in, t_idx, which=0, time_sel=0):
        """t_idx (B,) long -> raw output (B,1,H,W) of unet `which`; time_sel 0: alphas_cumsum[t]*T, 1:
        betas_cumsum[t]*T -- the two entries of the time list model_predictions builds (1160-1163)."""
        time = self.sch["alphas_cumsum" if time_sel == 0 else "betas_cumsum"][t_idx] * self.T
        x = torch.cat((x_t, x_in, self.x_cond2), dim=1) if self.input_condition else torch.cat((x_t, x_in), dim=1)
        sd = self.sd if which == 0 else self.sd1
        cond = None
        if self.hoist_cond:
            if which == 0:
                if self._cond is None:
                    self._cond = nets.da_unet_cond(sd, x_in)
                cond = self._cond
            else:
                if self._cond1 is None:
                    self._cond1 = nets.da_unet_cond(sd, x_in)
                cond = self._cond1
        return nets.da_unet(sd, x, time, cond, self.scan_fn)

    def model_outputs(self, x_in, x_t, t_idx):
        """UnetRes.forward (817-836): which UNets run, and with which time entry."""
        if self.num_unet == 2:
            o0 = self.unet(x_t, x_in, t_idx, 0, 0) if self.test in ("res_noise", "res") else None
            o1 = self.unet(x_t, x_in, t_idx, 1, 1) if self.test in ("res_noise", "noise") else None
            return o0, o1
        if self.objective == "pred_noise":
            return self.unet(x_t, x_in, t_idx, 0, 1), None
        return self.unet(x_t, x_in, t_idx, 0, 0), None

    def model_predictions(self, x_in, x_t, t_idx):
        """src/DADiff.py:1168-1207 with clip_denoised=True."""
        o0, o1 = self.model_outputs(x_in, x_t, t_idx)
        cl = lambda v: v.clamp(-1.0, 1.0)
        a = self.sch["alphas_cumsum"][t_idx].view(-1, 1, 1, 1)
        b = self.sch["betas_cumsum"][t_idx].view(-1, 1, 1, 1)
        oma = self.sch["one_minus_alphas_cumsum"][t_idx].view(-1, 1, 1, 1)
        from_res = lambda pr: ((x_t - x_in - (a - 1) * pr) / b, cl(x_in - pr))        # 1120-1124, 1206
        start_from_noise = lambda pn: cl((x_t - a * x_in - b * pn) / oma)             # 1126-1130
        if self.objective == "pred_res_noise":
            if self.test == "res_noise":
                pred_res, pred_noise = cl(o0), o1
                x_start = cl(x_t - a * pred_res - b * pred_noise)                     # 1132-1136
            elif self.test == "res":
                pred_res = cl(o0)
                pred_noise, x_start = from_res(pred_res)
            else:
                pred_noise = o1
                x_start = start_from_noise(pred_noise)
                pred_res = cl(x_in - x_start)
        elif self.objective == "pred_x0_noise":
            pred_res, pred_noise, x_start = cl(x_in - o0), o1, cl(o0)
        elif self.objective == "pred_noise":
            pred_noise = o0
            x_start = start_from_noise(pred_noise)
            pred_res = cl(x_in - x_start)
        else:
            pred_res = cl(o0)
            pred_noise, x_start = from_res(pred_res)
        return pred_res, pred_noise, x_start

    def q_posterior(self, pred_res, x_start, x_t, t_idx):
        e = lambda k: self.sch[k][t_idx].view(-1, 1, 1, 1)
        mean = e("posterior_mean_coef1") * x_t + e("posterior_mean_coef2") * pred_res + \
            e("posterior_mean_coef3") * x_start
        return mean, e("posterior_variance"), e("posterior_log_variance_clipped")

    def p_sample(self, x_in, x_t, t, noise):
        t_idx = torch.full((x_t.shape[0],), t, dtype=torch.long)
        pred_res, _, x_start = self.model_predictions(x_in, x_t, t_idx)
        mean, _, logvar = self.q_posterior(pred_res, x_start, x_t, t_idx)
        nz = noise if t > 0 else 0.0
        return mean + (0.5 * logvar).exp() * nz, x_start

    def sample(self, x_input01, noise0, step_noise=None, trace=None, t_stop=0, x_cond2_01=None):
        """x_input01 (B,1,H,W) in [0,1]; noise0 the one randn(shape) draw; step_noise[t] for
        the ancestral loop.  Returns [x_T01, out01] like sample(last=True).  x_cond2_01: the second
        list entry of sample(x_input=[a, b]) when input_condition (normalised unless
        input_condition_mask, src/DADiff.py:1372-1375)."""
        self._cond = self._cond1 = None
        if self.input_condition:
            self.x_cond2 = x_cond2_01 if self.input_condition_mask else x_cond2_01 * 2 - 1
        x_in = x_input01 * 2 - 1
        img = x_in + math.sqrt(self.sum_scale) * noise0
        start = img
        if self.S < self.T:
            for t, t_next in schedule.ddim_time_pairs(self.T, self.S):
                t_idx = torch.full((img.shape[0],), t, dtype=torch.long)
                pred_res, _, x_start = self.model_predictions(x_in, img, t_idx)
                if trace is not None:
                    trace.setdefault("pred_res", []).append(pred_res.clone())
                if t_next < 0:
                    img = x_start
                else:
                    alpha = self.sch["alphas_cumsum"][t] - self.sch["alphas_cumsum"][t_next]
                    img = img - alpha * pred_res
                if trace is not None:
                    trace.setdefault("img", []).append(img.clone())
        else:
            for t in reversed(range(t_stop, self.T)):
                img, _ = self.p_sample(x_in, img, t, None if t == 0 else step_noise[t])
                if trace is not None:
                    trace.setdefault("img", []).append(img.clone())
        return [(start + 1) * 0.5, (img + 1) * 0.5]


class GaussianOracle:
    def __init__(self, sd, prefix="model.", timesteps=1000, sampling_timesteps=None,
                 beta_schedule="cosine", objective="pred_noise", eta=0.0):
        self.sd = nets.SD(sd, prefix)
        self.T = timesteps
        self.S = sampling_timesteps if sampling_timesteps is not None else timesteps
        self.sch = schedule.gaussian_schedule(timesteps, beta_schedule)
        self.objective = objective
        self.eta = eta

    def model_predictions(self, x, t_idx, clip_x_start=False):
        out = nets.vanilla_unet(self.sd, x, t_idx)
        e = lambda k: self.sch[k][t_idx].view(-1, 1, 1, 1)
        clip = (lambda v: v.clamp(-1.0, 1.0)) if clip_x_start else (lambda v: v)
        if self.objective == "pred_noise":
            pred_noise = out
            x_start = clip(e("sqrt_recip_alphas_cumprod") * x - e("sqrt_recipm1_alphas_cumprod") * out)
        elif self.objective == "pred_x0":
            x_start = clip(out)
            pred_noise = (e("sqrt_recip_alphas_cumprod") * x - x_start) / e("sqrt_recipm1_alphas_cumprod")
        else:  # pred_v
            x_start = clip(e("sqrt_alphas_cumprod") * x - e("sqrt_one_minus_alphas_cumprod") * out)
            pred_noise = (e("sqrt_recip_alphas_cumprod") * x - x_start) / e("sqrt_recipm1_alphas_cumprod")
        return pred_noise, x_start

    def p_sample(self, x, t, noise):
        t_idx = torch.full((x.shape[0],), t, dtype=torch.long)
        _, x_start = self.model_predictions(x, t_idx)
        x_start = x_start.clamp(-1.0, 1.0)
        e = lambda k: self.sch[k][t_idx].view(-1, 1, 1, 1)
        mean = e("posterior_mean_coef1") * x_start + e("posterior_mean_coef2") * x
        nz = noise if t > 0 else 0.0
        return mean + (0.5 * e("posterior_log_variance_clipped")).exp() * nz, x_start

    def sample(self, x_T, step_noise=None, t_stop=0):
        """x_T the initial randn(shape); step_noise: list indexed by loop iteration for DDIM
        (the reference draws randn_like every DDIM step even at eta=0) or dict t->noise."""
        img = x_T
        if self.S < self.T:
            for i, (t, t_next) in enumerate(schedule.ddim_time_pairs(self.T, self.S)):
                t_idx = torch.full((img.shape[0],), t, dtype=torch.long)
                pred_noise, x_start = self.model_predictions(img, t_idx, clip_x_start=True)
                if t_next < 0:
                    img = x_start
                    continue
                a, an = self.sch["alphas_cumprod"][t], self.sch["alphas_cumprod"][t_next]
                sigma = self.eta * ((1 - a / an) * (1 - an) / (1 - a)).sqrt()
                c = (1 - an - sigma ** 2).sqrt()
                nz = step_noise[i] if (step_noise is not None and self.eta != 0) else 0.0
                img = x_start * an.sqrt() + c * pred_noise + sigma * nz
        else:
            for t in reversed(range(t_stop, self.T)):
                img, _ = self.p_sample(img, t, None if t == 0 else step_noise[t])
        return [(img + 1) * 0.5]
