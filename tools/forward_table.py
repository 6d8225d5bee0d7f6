#!/usr/bin/env python3
"""Every launch of one batch-8 UNet forward (512x512, bf16), replayed alone between HIP events: C-ABI entry point, shape,
time, algorithmic GFLOP (dense contractions) and the rate it reaches -- the per-launch view behind profiles/*stage_detail*
and the kernel_stats summaries.  usage: python tools/forward_table.py [--batch 8] [--precision fp32s] > profiles/rNN_forward_launches.md"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
from founddiff_amd import _lib as L, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=8)
ap.add_argument("--precision", default="bf16", help="bf16 | fp16 | fp32s | fp32 | fp8: the kernel mode of the forward")
a = ap.parse_args()
dev = torch.device("cuda")
dif, _ = bench.build_model(dev, precision=a.precision)
eng = dif._eng()
B = a.batch
_, ld = synth.ct_phantom(B, 512, seed=10)
x_in = (torch.from_numpy(ld).to(dev) * 2 - 1).contiguous()
img = (x_in + 0.1 * torch.randn_like(x_in)).contiguous()
tb = torch.full((B,), 500.0, device=dev)
eng.encode_condition(x_in)
eng.forward(img, x_in, tb)
L.TRACE = []
eng.forward(img, x_in, tb)
trace, L.TRACE = L.TRACE, None
lib = eng.hip.lib()          # the build of the library this engine calls (the binary16 one for --precision fp16)
KID = {0: "igemm 128x128", 1: "igemm 128x64", 2: "igemm 64x128", 3: "igemm 64x64", 4: "igemm 128x256", 5: "igemm 256x256",
       6: "igemm 128x32", 7: "pw_gemm 256x256 persistent", 10: "row-GEMM", 11: "halo 3x3", 12: "halo 3x3 fp8", 13: "3x3 weights in registers", 14: "halo 3x3, up-sampling as four 2x2 (GFLOP of the 9-tap form)",
       15: "halo 3x3 split-bf16 (fp32 storage)", 16: "halo 3x3 split-bf16, up-sampling as four 2x2 (GFLOP of the 9-tap form)", 17: "row-GEMM fp32 storage split-bf16"}
rows, tot_ms, tot_fl = [], 0.0, 0.0
for n, args in trace:
    ms = bench._time_launches(lib, [(n, args)], reps=3)
    desc, fl = "", 0.0
    if n == "fd_conv2d":
        q = args[0]._obj
        fl = bench.conv_flops(q)
        desc = (f"{KID.get(lib.fd_conv_kernel_id(args[0]), '?')}: {q.KH}x{q.KW} s{q.stride} {q.c0 + q.c1} -> {q.Cout} @ {q.OH}x{q.OW}"
                f"{' x4 dirs' if q.ndir == 4 else ''} epi {q.epilogue} pro {q.prologue}{' (z recomputed)' if q.prologue == 3 else ''}")
    elif n in ("fd_selective_scan", "fd_selective_scan_xproj"):
        o = 1 if n.endswith("xproj") else 0
        Bq, H, W, D, N, R = args[9 + o:15 + o]
        fl = Bq * (H * W) * D * (9.0 * N + 1)            # 9 L D N + D L per direction, 4 directions of L = HW / 4
        if o:
            fl += 2.0 * Bq * H * W * D * (R + 2 * N)
        desc = f"D={D} N={N} R={R} @ {H}x{W}" + (" (+ x_proj)" if o else "")
    elif n in ("fd_pw_dw3x3", "fd_pw_dw3x3_gram"):
        if n == "fd_pw_dw3x3":
            Cin, Cdw, Cz, H, W = args[4], args[12], args[19], args[24], args[25]
            fl = 2.0 * B * H * W * Cin * (Cdw + Cz) + 2.0 * B * H * W * Cdw * 9
            desc = f"LN -> 1x1 {Cin} -> {Cdw}+{Cz} -> dw3x3 @ {H}x{W}"
        else:
            H, W = args[18], args[19]
            nch = 192 if getattr(args[13], "value", args[13]) else 128       # out_v == NULL: q, k only
            fl = 2.0 * B * H * W * 64 * nch + 2.0 * B * H * W * nch * 9 + 2.0 * B * H * W * 64 * 32
            desc = f"LN -> {'qkv' if nch == 192 else 'qk'} 64 -> {nch} -> dw3x3 -> Gram @ {H}x{W}"
    elif n == "fd_gn_apply_down4x4":
        Bq, H, W, Cq, Co = args[11:16]
        fl = 2.0 * Bq * (H // 2) * (W // 2) * Co * 16 * Cq
        desc = f"GN apply + SiLU + residual -> skip; 4x4 s2 {Cq} -> {Co} @ {H // 2}x{W // 2}"
    elif n == "fd_pw_dw3x3_proj":
        H, W = args[20], args[21]
        fl = 2.0 * B * H * W * 64 * 64 * 2 + 2.0 * B * H * W * 64 * 9
        desc = f"LN -> v 64 -> 64 -> dw3x3 -> Weff 64 -> 64 -> gated residual @ {H}x{W}"
    rows.append((n, desc, ms, fl))
    tot_ms += ms
    tot_fl += fl
print(f"# One batch-{B} forward ({a.precision}), launch by launch (each replayed alone; sum {tot_ms:.2f} ms = {tot_ms / B:.3f} ms per slice, "
      f"{tot_fl / 1e9 / B:.1f} GFLOP per slice counted)\n")
print("| # | entry point | what | us | GFLOP | TFLOP/s |")
print("|---|---|---|---|---|---|")
for i, (n, d, ms, fl) in enumerate(rows):
    print(f"| {i} | `{n}` | {d} | {ms * 1e3:.1f} | {fl / 1e9:.2f} | {fl / ms / 1e9 if fl else 0:.0f} |")
