#!/usr/bin/env python3
"""Development probe: the u2m.xc / u2m.z tensors of one bf16 forward (fused C = 128 in_proj kernel, or with
FD_NO_PWDW128=1 the unfused pair) against the fp32 engine's."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def main():
    import bench
    from founddiff_amd import synth
    dev = torch.device("cuda")
    _, ld = synth.ct_phantom(2, 512, seed=10)
    x_in = torch.from_numpy(ld).to(dev) * 2 - 1
    x_t = (x_in + 0.1 * torch.randn(x_in.shape, generator=torch.Generator().manual_seed(4)).to(dev)).contiguous()
    tb = torch.full((2,), 500.0, device=dev)
    outs = {}
    for prec in ("fp32", "bf16"):
        dif, _ = bench.build_model(dev, 512, 50, prec)
        e = dif._eng()
        e.encode_condition(x_in)
        got = {}
        e.probe = lambda tag, t: got.__setitem__(tag, t.detach().float().clone()) if tag.startswith("u2m") or tag.startswith("u2r") else None
        e.forward(x_t, x_in, tb)
        torch.cuda.synchronize()
        outs[prec] = got
        if prec == "bf16":
            for name, t in e.buf.items():
                tf = t.float()
                n = int(torch.isnan(tf).sum())
                if n or str(name[0] if isinstance(name, tuple) else name) in ("qkv", "attn_v", "gram", "weff", "xz", "xc"):
                    per_img = [int(torch.isnan(tf[i]).sum()) for i in range(min(tf.shape[0], 4))] if tf.dim() > 1 else []
                    print(f"buf {str(name):40s} shape {tuple(t.shape)} nan {n} per-image {per_img}")
            print("mod_all nan", int(torch.isnan(e.mod_all).sum()), "local nan", int(torch.isnan(e.local_all).sum()))
    for k in outs["bf16"]:
        a, b = outs["bf16"][k].double(), outs["fp32"][k].double()
        if a.shape != b.shape:
            continue
        err = (a - b)
        print(f"{k:12s} L2 {float(err.norm() / b.norm()):.3e}  max|err| {float(err.abs().max()):.3e}  max|ref| {float(b.abs().max()):.3e}"
              f"  nan {int(torch.isnan(a).sum())}  worst at {tuple(int(v) for v in torch.unravel_index(err.abs().argmax(), err.shape))}")


if __name__ == "__main__":
    main()
