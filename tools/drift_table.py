#!/usr/bin/env python3
"""Per-stage bf16 drift of one UNet forward at the bench geometry (512x512, full architecture, CT phantom):
the bf16 engine against the fp32 (parity-mode) engine, stage by stage.

Three passes over the probe points of DAEngine.forward (engine.py `_pr`):
  cumulative   plain bf16 forward; L2-relative error of every probed tensor vs the fp32 engine's
  local        teacher-forced bf16 forward: every probed tensor is measured, then overwritten with the fp32
               engine's value (rounded to the storage type) -- the error a stage adds on (nearly) exact inputs
  one-at-a-time  for every main-stream point p: a bf16 forward whose main-stream tensors up to and including p
               are replaced by the fp32 engine's; the remaining error of the model output is what the stages
               AFTER p contribute.  The drop between consecutive rows is the share of that block.
Writes a markdown table (stdout or --out).  Development / evidence tool: profiles/rNN_drift_table.md.
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def l2(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / b.norm().clamp(min=1e-30))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--image", default="phantom", choices=["phantom", "uniform"])
    ap.add_argument("--t", type=float, default=500.0)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    import bench
    from founddiff_amd import synth
    dev = torch.device("cuda")
    if a.image == "phantom":
        _, ld = synth.ct_phantom(1, a.size, seed=10)
        x_in = torch.from_numpy(ld).to(dev) * 2 - 1
    else:
        x_in = torch.rand(1, 1, a.size, a.size, generator=torch.Generator().manual_seed(4)).to(dev) * 2 - 1
    x_t = (x_in + 0.1 * torch.randn(x_in.shape, generator=torch.Generator().manual_seed(4)).to(dev)).contiguous()
    tb = torch.full((1,), a.t, device=dev)

    def engine(prec):
        dif, _ = bench.build_model(dev, a.size, 50, prec)
        e = dif._eng()
        e.encode_condition(x_in)
        return dif, e

    ref, order = {}, []

    def rec(tag, t):
        ref[tag] = t.detach().float().clone()
        order.append(tag)
    d32, e32 = engine("fp32")
    e32.probe = rec
    e32.forward(x_t, x_in, tb)
    torch.cuda.synchronize()
    e32.probe = None
    cond32 = {k: getattr(e32, k).clone() for k in ("prompt_emb", "local_all", "mod_all")}
    del d32, e32
    torch.cuda.empty_cache()

    # share of the WEIGHT rounding alone: the fp32 engine with its dense weights rounded through bf16
    from founddiff_amd import engine as E
    orig_init = E.ConvW.__init__

    def rounded_init(self, *aa, **kk):
        orig_init(self, *aa, **kk)
        self.w = self.w.to(torch.bfloat16).to(self.w.dtype)
    E.ConvW.__init__ = rounded_init
    dw, ew = engine("fp32")
    E.ConvW.__init__ = orig_init
    for m in ew.mambas:
        m["x_proj"] = m["x_proj"].to(torch.bfloat16).to(m["x_proj"].dtype)
    w_only = l2(ew.forward(x_t, x_in, tb), ref["out"])
    del dw, ew
    torch.cuda.empty_cache()

    d16, e16 = engine("bf16")
    lines = ["# bf16 drift by stage: %dx%d, %s image, t=%.0f (L2-relative vs the fp32 engine)" % (a.size, a.size, a.image, a.t), ""]
    lines.append("conditioning vectors: " + ", ".join(f"{k} {l2(getattr(e16, k), v):.2e}" for k, v in cond32.items() if k != "mod_all"))

    def run(correct):
        errs = {}

        def hook(tag, t):
            errs[tag] = l2(t.float(), ref[tag])
            if correct(tag):
                t.copy_(ref[tag].to(t.dtype))
        e16.probe = hook
        out = e16.forward(x_t, x_in, tb)
        torch.cuda.synchronize()
        e16.probe = None
        return errs, l2(out, ref["out"])
    cum, fin = run(lambda tag: False)
    loc, _ = run(lambda tag: not tag.endswith(".conv3"))
    lines += ["", f"model output, plain bf16 forward: **{fin:.3e}**",
              f"model output, fp32 engine with the conv / 1x1 weights rounded to bf16 (weight rounding alone): **{w_only:.3e}**", "",
              "| # | stage | shape | cumulative | local (teacher-forced) |", "|---|---|---|---|---|"]
    for i, tag in enumerate(order):
        lines.append(f"| {i} | {tag} | {tuple(ref[tag].shape)} | {cum[tag]:.2e} | {loc[tag]:.2e} |")
    main_pts = [t for t in order if "." not in t or t.endswith(".x1")]
    lines += ["", "One replacement at a time: main-stream tensors up to and including the row's stage taken from the fp32 engine;",
              "`output error` = what the stages after it still add; `share` = drop from the previous row.", "",
              "| replaced up to | output error | share of the output error removed |", "|---|---|---|",
              f"| (nothing) | {fin:.3e} | |"]
    prev = fin
    for k, p in enumerate(main_pts[:-1]):
        upto = set(main_pts[:k + 1])
        _, f = run(lambda tag: tag in upto)
        lines.append(f"| {p} | {f:.3e} | {prev - f:+.2e} |")
        prev = f
    txt = "\n".join(lines) + "\n"
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        open(a.out, "w").write(txt)
    print(txt)


if __name__ == "__main__":
    main()
