#!/usr/bin/env python3
"""Per-launch timing table of one UNet forward (512x512 by default): traces every C-ABI call of
DAEngine.forward, then times each launch on its own (events on the launch stream, best of N),
and prints algorithmic GFLOP / MB and the implied TFLOP/s / GB/s.  Development tool."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--precision", default="bf16")
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--json", default=None)
    a = ap.parse_args()
    import bench
    from founddiff_amd import _lib as L
    from founddiff_amd import synth
    dif, w = bench.build_model(torch.device("cuda"), a.size, 50, a.precision)
    eng = dif._eng()
    _, ld = synth.ct_phantom(a.batch, a.size, seed=10)
    x = torch.from_numpy(ld).cuda()
    x_in = (x * 2 - 1).contiguous()
    img = (x_in + 0.1 * torch.randn_like(x_in)).contiguous()
    tb = torch.full((a.batch,), 500.0, device="cuda")
    eng.encode_condition(x_in)
    eng.forward(img, x_in, tb)
    L.TRACE = []
    eng.forward(img, x_in, tb)
    trace, L.TRACE = L.TRACE, None
    lib = L.lib()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    es = 2 if a.precision == "bf16" else 4
    rows = []
    for i, (n, args) in enumerate(trace):
        fn = getattr(lib, n)
        best = 1e9
        for _ in range(a.reps):
            e0.record()
            fn(*args)
            e1.record()
            e1.synchronize()
            best = min(best, e0.elapsed_time(e1))
        desc, gf, mb = n, 0.0, 0.0
        if n == "fd_conv2d":
            p = args[0]._obj
            cin = p.c0 + p.c1
            gf = 2.0 * p.B * p.OH * p.OW * p.Cout * p.KH * p.KW * cin * p.ndir / 1e9
            mb = (p.B * p.H * p.W * cin * es + p.B * p.OH * p.OW * p.Cout * p.ndir * (4 if p.out_f32 else es)) / 1e6
            desc = f"conv {p.KH}x{p.KW} s{p.stride} {cin}->{p.Cout} @{p.OH}x{p.OW} epi{p.epilogue}" + \
                   (" up" if p.upsample else "") + (" dir4" if p.ndir > 1 else "")
        rows.append((i, desc, best * 1e3, gf, mb))
    tot = sum(r[2] for r in rows)
    print(f"{'#':>3} {'op':58s} {'us':>8s} {'GFLOP':>8s} {'TF/s':>7s} {'MB':>7s} {'GB/s':>7s}")
    for i, d, us, gf, mb in rows:
        print(f"{i:3d} {d:58s} {us:8.1f} {gf:8.2f} {gf / us * 1e3 if gf else 0:7.1f} {mb:7.1f} {mb / us * 1e3 if mb else 0:7.0f}")
    print(f"total {tot / 1e3:.3f} ms over {len(rows)} launches")
    agg = {}
    for i, d, us, gf, mb in rows:
        k = d.split(" @")[0] if d.startswith("conv") else d
        agg.setdefault(k, [0, 0.0, 0.0])
        agg[k][0] += 1
        agg[k][1] += us
        agg[k][2] += gf
    print("\nby kind:")
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"  {k:50s} n={v[0]:3d} {v[1]:9.1f} us  {v[2]:8.1f} GF")
    if a.json:
        json.dump(rows, open(a.json, "w"))


if __name__ == "__main__":
    main()
