#!/usr/bin/env python3
"""End-to-end bf16-vs-fp32-engine drift of a full S-step DDIM sample() (full architecture, CT phantom) and
the throughput of both modes: the figures DESIGN.md section 4 quotes."""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sizes", default="256,512")
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--batch", type=int, default=2)
    a = ap.parse_args()
    import bench
    from founddiff_amd import synth
    dev = torch.device("cuda")
    for size in [int(s) for s in a.sizes.split(",")]:
        _, ld = synth.ct_phantom(a.batch, size, seed=10)
        x = torch.from_numpy(ld).to(dev)
        nz = torch.randn(a.batch, 1, size, size, generator=torch.Generator().manual_seed(7)).to(dev)
        outs = {}
        for prec in ("fp32", "bf16", "bf16+1", "bf16+1/1", "bf16+1/2"):
            dif, _ = bench.build_model(dev, size, a.steps, prec.split("+")[0])
            tail = prec.split("+")[1] if "+" in prec else "0"
            dif.final_fp32_steps = int(tail.split("/")[0])
            dif.final_outer_levels = int(tail.split("/")[1]) if "/" in tail else 0
            dif.sample([x], batch_size=a.batch, noise=nz)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            outs[prec] = dif.sample([x], batch_size=a.batch, noise=nz)[-1].float().cpu()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            print(f"{size}x{size} S={a.steps} B={a.batch} {prec}: {a.batch / dt:.3f} slices/s ({dt * 1e3 / a.steps / a.batch:.2f} ms/forward/slice)", flush=True)
            del dif
            torch.cuda.empty_cache()
        for prec in list(outs)[1:]:
            d = outs[prec].double() - outs["fp32"].double()
            l2 = float(d.norm() / outs["fp32"].double().norm())
            mx = float(d.abs().max() / outs["fp32"].abs().max())
            psnr = float(10 * torch.log10(1.0 / (d ** 2).mean()))
            print(f"{size}x{size} S={a.steps}: {prec} vs fp32 engine: L2-rel {l2:.3e}, max-rel {mx:.3e}, PSNR {psnr:.1f} dB", flush=True)


if __name__ == "__main__":
    main()
