#!/usr/bin/env python3
"""fp8-weight mode (BASELINE configs[4]: 25-step DDIM, e4m3 weights on the fp8 MFMA): drift of a full sample()
against the fp32 engine and the throughput next to the bf16 mode.  Development / evidence tool."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def main():
    import bench
    from founddiff_amd import synth
    dev = torch.device("cuda")
    size, S, B = 512, 25, int(os.environ.get("B", "8"))
    _, ld = synth.ct_phantom(B, size, seed=10)
    x = torch.from_numpy(ld).to(dev)
    nz = torch.randn(B, 1, size, size, generator=torch.Generator().manual_seed(7)).to(dev)
    outs = {}
    for tag, prec, k in (("fp32", "fp32", 0), ("bf16", "bf16", 1), ("fp8", "fp8", 1), ("fp8 pure", "fp8", 0)):
        if tag == "fp32" and B > 2:
            xs, ns = x[:2], nz[:2]
        else:
            xs, ns = x, nz
        dif, _ = bench.build_model(dev, size, S, prec)
        dif.final_fp32_steps = k
        dif.sample([xs], batch_size=xs.shape[0], noise=ns)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        outs[tag] = dif.sample([xs], batch_size=xs.shape[0], noise=ns)[-1].float().cpu()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"{tag}: {xs.shape[0] / dt:.3f} slices/s ({dt * 1e3 / S / xs.shape[0]:.3f} ms/forward/slice)", flush=True)
        del dif
        torch.cuda.empty_cache()
    ref = outs["fp32"]
    for tag in ("bf16", "fp8", "fp8 pure"):
        d = outs[tag][:ref.shape[0]].double() - ref.double()
        print(f"{tag} vs fp32 engine: L2-rel {float(d.norm() / ref.double().norm()):.3e}, PSNR {float(10 * torch.log10(1.0 / (d ** 2).mean())):.1f} dB")


if __name__ == "__main__":
    main()
