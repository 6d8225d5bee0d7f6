#!/usr/bin/env python3
"""CPU experiment (test infrastructure; NOT a test, NOT on the product path): what would the drift of the bf16 engine be if its
block-INTERNAL tensors were stored as fp16 instead of bf16?  ("measure the fp16-internal-storage direction".)

The CPU oracle's forward is re-walked with rounding inserted where the HIP engine stores or stages a tensor in 16 bits
(founddiff_amd/engine.py):
  main   = the residual stream / block outputs / conv outputs that live in HBM between kernels   (always bf16 here)
  inner  = block-internal tensors: the LayerNorm'd operand of in_proj / qkv, xc, y, the gated out_proj operand, z, q / k / v,
           the raw 3x3 output under GroupNorm                                                     (bf16 | fp16 | fp32)
  weight = every dense weight                                                                      (bf16 | fp16 | fp32)
and the S-step DDIM loop of BASELINE configs[1] (256x256) is run per variant against the unrounded oracle, with and without the
precision tail (last step unrounded = an upper bound of what the fp32s tail buys).  Rounding points are a model of the engine,
not the engine: the absolute numbers are indicative (the engine's production mode measures 8.1e-3 L2 at 256x256, this model's
bf16 / bf16 / bf16 row should land near it), the RATIOS between the rows are what the experiment is for.

    python tests/sim_storage_formats.py [--size 256] [--steps 50]  >  profiles/r06/storage_format_simulation.md
"""
import argparse
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from founddiff_amd import arch, synth  # noqa: E402
from oracle import nets, sampler, schedule  # noqa: E402


def rnd(kind):
    if kind == "fp32":
        return lambda t: t
    dt = torch.bfloat16 if kind == "bf16" else torch.float16
    return lambda t: t.to(dt).float()


class Sim:
    def __init__(self, main, inner, weight):
        self.qm, self.qi, self.qw = rnd(main), rnd(inner), rnd(weight)

    def block(self, sd, x):
        h = F.conv2d(x, self.qw(nets.ws_weight(sd["proj.weight"])), sd["proj.bias"], padding=1)
        h = self.qi(h)                                     # the raw 3x3 output (`res_h`) under GroupNorm
        return F.silu(F.group_norm(h, 8, sd["norm.weight"], sd["norm.bias"], eps=1e-5))

    def resblock(self, sd, x):
        h = self.block(sd.sub("block1."), x)
        if sd.has("res_conv.weight"):
            return self.qm(h + F.conv2d(x, self.qw(sd["res_conv.weight"]), sd["res_conv.bias"]))
        return self.qm(h + x)

    def ss2d(self, sd, h, c):
        local = F.silu(F.linear(c, sd["attn.0.weight"]))
        hq = self.qi(h)                                    # LayerNorm'd operand staged in 16 bits
        xz = F.linear(hq, self.qw(sd["in_proj.weight"]))
        xi, z = xz.chunk(2, dim=-1)
        z = self.qi(F.silu(z))
        xi = self.qi(xi).permute(0, 3, 1, 2)               # the 1x1 tile on chip
        xi = self.qi(F.silu(F.conv2d(xi, sd["conv2d.weight"], sd["conv2d.bias"], padding=1, groups=xi.shape[1])))    # xc
        # the scan itself: x_proj on the 16-bit xc, fp32 state, y stored in 16 bits
        Bn, D, H, W = xi.shape
        xs = nets.efficient_scan(xi)
        L = xs.shape[-1]
        xw, dtw, dtb = sd["x_proj_weight"], sd["dt_projs_weight"], sd["dt_projs_bias"]
        N = sd["A_logs"].shape[1]
        K, _, R = dtw.shape
        x_dbl = torch.einsum("bkdl,kcd->bkcl", xs, self.qw(xw))
        dts, Bs, Cs = torch.split(x_dbl, [R, N, N], dim=2)
        dts = torch.einsum("bkrl,kdr->bkdl", dts, dtw)
        ys = nets.selective_scan(xs.reshape(Bn, -1, L), dts.reshape(Bn, -1, L), -torch.exp(sd["A_logs"].float()),
                                 Bs.contiguous(), Cs.contiguous(), sd["Ds"].float(), dtb.reshape(-1).float(), True)
        y = self.qi(nets.efficient_merge(ys.view(Bn, K, -1, L), H, W)).transpose(1, 2)
        y = F.layer_norm(y, (D,), sd["out_norm.weight"], sd["out_norm.bias"], eps=1e-5).reshape(Bn, H, W, D)
        g = self.qi(y * z + local.unsqueeze(1))            # the gated operand of out_proj
        return F.linear(g, self.qw(sd["out_proj.weight"]))

    def attention(self, sd, h):
        b, C, H, W = h.shape
        heads = sd["temperature"].shape[0]
        qkv = self.qi(F.conv2d(self.qi(h), self.qw(sd["qkv.weight"])))
        qkv = self.qi(F.conv2d(qkv, sd["qkv_dwconv.weight"], padding=1, groups=3 * C))
        q, k, v = qkv.chunk(3, dim=1)
        q, k, v = (t.reshape(b, heads, C // heads, H * W) for t in (q, k, v))
        attn = ((F.normalize(q, dim=-1) @ F.normalize(k, dim=-1).transpose(-2, -1)) * sd["temperature"]).softmax(dim=-1)
        weff = self.qw(torch.einsum("ohi,bhij->bohj", sd["project_out.weight"].reshape(C, heads, C // heads), attn).reshape(b, C, C))
        return torch.einsum("boc,bcp->bop", weff, v.reshape(b, C, H * W)).reshape(b, C, H, W)

    def mamba(self, sd, x, c, t):
        C = x.shape[1]
        x = x.permute(0, 2, 3, 1)
        mod = F.linear(F.silu(t), sd["adaLN_modulation.1.weight"], sd["adaLN_modulation.1.bias"])
        sh1, sc1, g1, sh2, sc2, g2 = [m[:, None, None, :] for m in mod.chunk(6, dim=1)]
        h = F.layer_norm(x, (C,), sd["norm1.weight"], sd["norm1.bias"], eps=1e-5) * (1 + sc1) + sh1
        x = self.qm(x + g1 * self.ss2d(sd.sub("mamba."), h, c))
        h = F.layer_norm(x, (C,), None, None, eps=1e-6) * (1 + sc2) + sh2
        a = self.attention(sd.sub("attn_blk."), h.permute(0, 3, 1, 2))
        return self.qm(x + g2 * a.permute(0, 2, 3, 1)).permute(0, 3, 1, 2)

    def unet(self, sd, x, time, cond):
        dim = sd["init_conv.weight"].shape[0]
        c, pe = cond
        x = self.qm(F.conv2d(x, self.qw(sd["init_conv.weight"]), sd["init_conv.bias"], padding=3))
        r = x
        t = nets.time_mlp(sd.sub("time_mlp."), time, dim) + pe
        hs = []
        for i in range(nets._stage_count(sd, "downs")):
            s = sd.sub(f"downs.{i}.")
            x = self.mamba(s.sub("1."), x, c, t)
            x = self.resblock(s.sub("0."), x)
            hs.append(x)
            w = s["2.weight"]
            x = self.qm(F.conv2d(x, self.qw(w), s["2.bias"], stride=2, padding=1) if w.shape[-1] == 4 else
                        F.conv2d(x, self.qw(w), s["2.bias"], padding=1))
        x = self.resblock(sd.sub("mid_block."), x)
        x = self.mamba(sd.sub("mid_attn."), x, c, t)
        for i in range(nets._stage_count(sd, "ups")):
            s = sd.sub(f"ups.{i}.")
            x = torch.cat((x, hs.pop()), dim=1)
            x = self.resblock(s.sub("0."), x)
            x = self.mamba(s.sub("1."), x, c, t)
            if s.has("2.1.weight"):
                x = self.qm(F.conv2d(F.interpolate(x, scale_factor=2, mode="nearest"), self.qw(s["2.1.weight"]), s["2.1.bias"], padding=1))
            else:
                x = self.qm(F.conv2d(x, self.qw(s["2.weight"]), s["2.bias"], padding=1))
        x = torch.cat((x, r), dim=1)
        x = self.resblock(sd.sub("final_res_block."), x)
        return F.conv2d(x, sd["final_conv.weight"], sd["final_conv.bias"])


def ddim(orc, sims, x01, noise0, S):
    """the oracle's DDIM loop (oracle/sampler.py: ResidualOracle.sample) with the UNet of step i = sims[i]"""
    x_in = x01 * 2 - 1
    img = x_in + math.sqrt(orc.sum_scale) * noise0
    cond = nets.da_unet_cond(orc.sd, x_in)
    pairs = list(schedule.ddim_time_pairs(orc.T, S))
    for i, (t, t_next) in enumerate(pairs):
        tt = torch.full((img.shape[0],), t, dtype=torch.long)
        time = orc.sch["alphas_cumsum"][tt] * orc.T
        pred_res = sims[i].unet(orc.sd, torch.cat((img, x_in), 1), time.float(), cond).clamp(-1, 1)
        if t_next < 0:
            img = (x_in - pred_res).clamp(-1, 1)
        else:
            img = img - (orc.sch["alphas_cumsum"][t] - orc.sch["alphas_cumsum"][t_next]) * pred_res
    return (img + 1) * 0.5


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--variants", default="", help="main/inner/weight triples, comma separated (default: the standard table)")
    a = ap.parse_args()
    torch.set_num_threads(min(32, os.cpu_count()))
    spec = arch.da_unet_spec(64, (1, 2, 4, 8), prefix="model.unet0.")
    w = synth.synth_state_dict(spec, seed=0)
    _, ld = synth.ct_phantom(1, a.size, seed=10)
    x01 = torch.from_numpy(ld)
    noise0 = torch.randn(1, 1, a.size, a.size, generator=torch.Generator().manual_seed(7))
    orc = sampler.ResidualOracle(w, prefix="model.unet0.", sampling_timesteps=a.steps)
    exact = Sim("fp32", "fp32", "fp32")
    with torch.no_grad():
        ref = ddim(orc, [exact] * a.steps, x01, noise0, a.steps)
        chk = orc.sample(x01, noise0)[-1]
        print(f"# Storage-format simulation on the CPU oracle: {a.size}x{a.size}, {a.steps}-step DDIM (tests/sim_storage_formats.py)\n")
        print(f"(the unrounded walk reproduces oracle.sampler to {float((ref - chk).abs().max()):.1e})\n")
        print("| main stream | block-internal | weights | last step | L2 vs the unrounded loop | PSNR dB |")
        print("|---|---|---|---|---|---|")
        table = (("bf16", "bf16", "bf16"), ("bf16", "fp16", "bf16"), ("bf16", "fp16", "fp16"), ("fp16", "fp16", "fp16"),
                 ("bf16", "fp32", "bf16"), ("fp32", "fp32", "bf16"))
        if a.variants:
            table = tuple(tuple(v.split("/")) for v in a.variants.split(","))
        for main_, inner, wt in table:
            sim = Sim(main_, inner, wt)
            for tail in (False, True):
                out = ddim(orc, [sim] * (a.steps - 1) + [exact if tail else sim], x01, noise0, a.steps)
                l2 = float((out - ref).norm() / ref.norm())
                mse = float(((out - ref) ** 2).mean())
                print(f"| {main_} | {inner} | {wt} | {'unrounded' if tail else 'like the rest'} | {l2:.2e} | {10 * math.log10(1.0 / mse):.1f} |", flush=True)


if __name__ == "__main__":
    main()
