#!/usr/bin/env python3
"""Micro-benchmarks of single C-ABI entry points at the shapes of a batch-8 512x512 forward (development tool).
usage: python tools/kbench.py scan|attn|... [--batch 8]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def timeit(fn, reps=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3      # us


def bench_scan(B, shapes=None):
    from founddiff_amd import _lib as L
    s = torch.cuda.current_stream().cuda_stream
    shapes = shapes or [(128, 4, 4, 512, 512), (128, 8, 4, 256, 256), (256, 8, 8, 256, 256), (256, 16, 8, 128, 128),
                        (512, 16, 16, 128, 128), (512, 32, 16, 64, 64), (1024, 32, 32, 64, 64)]
    for D, N, R, H, W in shapes:
        torch.manual_seed(0)
        CD, Lq = R + 2 * N, (H // 2) * (W // 2)
        xc = (torch.randn(B, H, W, D, device="cuda") * 0.5).to(torch.bfloat16)
        xdbl = torch.randn(4, B, Lq, CD, device="cuda")
        dtw = ((torch.rand(4, D, R, device="cuda") * 2 - 1) * R ** -0.5)
        dtb = torch.randn(4, D, device="cuda") * 0.5 - 3
        A = -torch.exp(torch.log(torch.arange(1, N + 1, device="cuda").float())[None].repeat(4 * D, 1))
        Ds = torch.ones(4 * D, device="cuda")
        ws = torch.empty(L.lib().fd_scan_ws_floats(B, H, W, D, N), device="cuda")
        y = torch.empty(B, H, W, D, device="cuda", dtype=torch.bfloat16)

        def run():
            L.call("fd_selective_scan", L.FD_BF16, xc.data_ptr(), xdbl.data_ptr(), dtw.data_ptr(), dtb.data_ptr(),
                   A.data_ptr(), Ds.data_ptr(), y.data_ptr(), ws.data_ptr(), B, H, W, D, N, R, s)
        t = timeit(run)
        print(f"scan D={D} N={N} R={R} {H}x{W} B={B}: {t:8.1f} us  ({t / B / 1e3:.4f} ms/slice)  y_sum={float(y.float().abs().mean()):.6f}", flush=True)


def bench_scanx(B):
    """selective scan with the x_proj einsum inside phase A (levels 0-2 of a 512x512 forward)"""
    from founddiff_amd import _lib as L
    s = torch.cuda.current_stream().cuda_stream
    for D, N, R, H, W in [(128, 4, 4, 512, 512), (128, 8, 4, 256, 256), (256, 8, 8, 256, 256), (256, 16, 8, 128, 128)]:
        if not L.lib().fd_selective_scan_plan(L.FD_BF16, D, N, R, H, W):
            print(f"scanx D={D} N={N} R={R} {H}x{W}: not fused", flush=True)
            continue
        torch.manual_seed(0)
        CD, Lq = R + 2 * N, (H // 2) * (W // 2)
        xc = (torch.randn(B, H, W, D, device="cuda") * 0.5).to(torch.bfloat16)
        xw = (torch.randn(4, CD, D, device="cuda") * D ** -0.5).to(torch.bfloat16)
        xdbl = torch.empty(4, B, Lq, CD, device="cuda")
        dtw = ((torch.rand(4, D, R, device="cuda") * 2 - 1) * R ** -0.5)
        dtb = torch.randn(4, D, device="cuda") * 0.5 - 3
        A = -torch.exp(torch.log(torch.arange(1, N + 1, device="cuda").float())[None].repeat(4 * D, 1))
        Ds = torch.ones(4 * D, device="cuda")
        ws = torch.empty(L.lib().fd_scan_ws_floats(B, H, W, D, N), device="cuda")
        y = torch.empty(B, H, W, D, device="cuda", dtype=torch.bfloat16)

        def run():
            L.call("fd_selective_scan_xproj", L.FD_BF16, xc.data_ptr(), xw.data_ptr(), xdbl.data_ptr(), dtw.data_ptr(),
                   dtb.data_ptr(), A.data_ptr(), Ds.data_ptr(), y.data_ptr(), ws.data_ptr(), B, H, W, D, N, R, s)
        t = timeit(run)
        print(f"scanx D={D} N={N} R={R} {H}x{W} B={B}: {t:8.1f} us  ({t / B / 1e3:.4f} ms/slice)  y={float(y.float().abs().mean()):.6f} xdbl={float(xdbl.abs().mean()):.6f}", flush=True)


def bench_attn(B, sizes=((512, 512), (256, 256))):
    """channel-attention branch of a 64-channel Mamba block: fused (qkv+dw+Gram) vs unfused (qkv+dw, Gram)"""
    from founddiff_amd import _lib as L
    from founddiff_amd.engine import DAEngine
    s = torch.cuda.current_stream().cuda_stream
    for H, W in sizes:
        torch.manual_seed(0)
        x = torch.randn(B, H, W, 64, device="cuda").to(torch.bfloat16)
        mod = torch.randn(B, 384, device="cuda") * 0.5
        wpw = (torch.randn(192, 64, device="cuda") / 8).to(torch.bfloat16)
        wm = DAEngine._dw_masked((torch.randn(192, 9, device="cuda") / 3).t().contiguous())
        qkv2 = torch.empty(B, H, W, 192, device="cuda", dtype=torch.bfloat16)
        v = torch.empty(B, H, W, 64, device="cuda", dtype=torch.bfloat16)
        nb0, nb1 = L.lib().fd_chan_attn_nblk(H * W), L.lib().fd_pw_dw3x3_gram_nblk(H, W)
        p0, p1 = torch.empty(B, 2, nb0, 1088, device="cuda"), torch.empty(B, 2, nb1, 1088, device="cuda")
        temp, wp = torch.ones(2, device="cuda"), torch.randn(64, 64, device="cuda") / 8
        weff = torch.empty(B, 64, 64, device="cuda", dtype=torch.bfloat16)

        def unf_a():
            L.call("fd_pw_dw3x3", L.FD_BF16, x.data_ptr(), 64, 0, 64, None, None, 1e-6, mod.data_ptr(), mod.data_ptr() + 256,
                   384, wpw.data_ptr(), 192, wm.data_ptr(), None, 0, qkv2.data_ptr(), 192, 0, 0, None, 0, 0, B, H, W, s)

        def unf_b():
            L.call("fd_chan_attn_gram", L.FD_BF16, qkv2.data_ptr(), B, H * W, 64, p0.data_ptr(), s)

        def unf_c():
            L.call("fd_chan_attn_weff", L.FD_BF16, p0.data_ptr(), nb0, temp.data_ptr(), wp.data_ptr(), weff.data_ptr(), B, 64, s)

        def fus_a():
            L.call("fd_pw_dw3x3_gram", L.FD_BF16, x.data_ptr(), 64, 0, 64, None, None, 1e-6, mod.data_ptr(), mod.data_ptr() + 256,
                   384, wpw.data_ptr(), wm.data_ptr(), v.data_ptr(), 64, 0, p1.data_ptr(), B, H, W, s)

        def fus_c():
            L.call("fd_chan_attn_weff", L.FD_BF16, p1.data_ptr(), nb1, temp.data_ptr(), wp.data_ptr(), weff.data_ptr(), B, 64, s)
        ta, tb, tc, fa, fc = timeit(unf_a), timeit(unf_b), timeit(unf_c), timeit(fus_a), timeit(fus_c)
        print(f"attn {H}x{W} B={B}: unfused qkv+dw {ta:.1f} + gram {tb:.1f} + weff {tc:.1f} = {ta + tb + tc:.1f} us | "
              f"fused {fa:.1f} + weff {fc:.1f} = {fa + fc:.1f} us  (nblk {nb0} -> {nb1})", flush=True)


def bench_conv3(B, shapes=None):
    """3x3 halo convolutions of a 512x512 forward: (c0, c1, cout, OH, upsample)"""
    from founddiff_amd import _lib as L
    from founddiff_amd.engine import DAEngine, ConvW, _T

    class Bare(DAEngine):
        def __init__(self):
            self.mode = "bf16"
            self.dt, self.tdt = _T["bf16"]
            self.dev = torch.device("cuda")
            self.buf = {}
    e = Bare()
    shapes = shapes or [(64, 0, 64, 512, False), (64, 64, 64, 512, False), (128, 0, 64, 512, True), (128, 64, 128, 256, False),
                        (256, 0, 128, 256, True), (256, 128, 256, 128, False), (512, 0, 256, 128, True), (512, 256, 512, 64, False)]
    for c0, c1, cout, OH, up in shapes:
        torch.manual_seed(0)
        H = OH // 2 if up else OH
        cin = c0 + c1
        cw = ConvW(torch.randn(cout, cin, 3, 3) / (3 * cin ** 0.5), torch.randn(cout), e.dev, e.tdt, up2x=up)
        xa = torch.randn(B, H, H, c0, device="cuda").to(torch.bfloat16)
        xb = torch.randn(B, H, H, c1, device="cuda").to(torch.bfloat16) if c1 else None
        out = torch.empty(B, OH, OH, cout, device="cuda", dtype=torch.bfloat16)
        part = torch.empty(B, L.lib().fd_conv_mtiles(OH, OH), cout, 2, device="cuda")
        kw = dict(c0=c0, stats=None if up else part, upsample=up)       # (the up-sampling convs of the forward emit no GroupNorm sums)
        if c1:
            kw.update(in1=xb, c1=c1)
        kid = e.conv(cw, xa, B, H, H, out, probe="kid", **kw)
        assert kid in (11, 13, 14)
        t = timeit(lambda: e.conv(cw, xa, B, H, H, out, **kw))
        fl = 2.0 * B * OH * OH * cout * 9 * cin
        print(f"conv3x3 {cin:4d}->{cout:4d} @{OH} up={int(up)} kid={kid} B={B}: {t:8.1f} us  {fl / t / 1e6:7.1f} TFLOP/s of the 9-tap form  ({fl / t / 1e6 / 2500:.3f} of peak)", flush=True)


def bench_pw(B):
    """1x1 convolutions of the deep levels (64x64, 128x128): (cin, cout, HW)"""
    from founddiff_amd.engine import DAEngine, ConvW, _T

    class Bare(DAEngine):
        def __init__(self):
            self.mode = "bf16"
            self.dt, self.tdt = _T["bf16"]
            self.dev = torch.device("cuda")
            self.buf = {}
    e = Bare()
    for cin, cout, hw in [(512, 2048, 64), (1024, 512, 64), (512, 1536, 64), (512, 512, 64), (256, 1024, 128), (512, 256, 128), (256, 768, 128)]:
        torch.manual_seed(0)
        cw = ConvW(torch.randn(cout, cin) / cin ** 0.5, torch.randn(cout), e.dev, e.tdt)
        x = torch.randn(B, hw, hw, cin, device="cuda").to(torch.bfloat16)
        out = torch.empty(B, hw, hw, cout, device="cuda", dtype=torch.bfloat16)
        kid = e.conv(cw, x, B, hw, hw, out, probe="kid")
        t = timeit(lambda: e.conv(cw, x, B, hw, hw, out))
        fl = 2.0 * B * hw * hw * cout * cin
        print(f"pw {cin:4d}->{cout:4d} @{hw} B={B} kid={kid}: {t:8.1f} us  {fl / t / 1e6:7.1f} TFLOP/s  ({fl / t / 1e6 / 2500:.3f} of peak)  chk={float(out.float().abs().mean()):.5f}", flush=True)


def bench_pwdw(B, sizes=((512, 512), (256, 256))):
    """in_proj variant of the fused LN -> 1x1 -> depthwise kernel (Cdw = 128 + SiLU, Cz = 128)"""
    from founddiff_amd import _lib as L
    from founddiff_amd.engine import DAEngine
    s = torch.cuda.current_stream().cuda_stream
    for H, W in sizes:
        torch.manual_seed(0)
        x = torch.randn(B, H, W, 64, device="cuda").to(torch.bfloat16)
        mod = torch.randn(B, 384, device="cuda") * 0.5
        g, be = torch.randn(64, device="cuda"), torch.randn(64, device="cuda")
        wpw = (torch.randn(256, 64, device="cuda") / 8).to(torch.bfloat16)
        wm = DAEngine._dw_masked((torch.randn(128, 9, device="cuda") / 3).t().contiguous())
        bdw = torch.randn(128, device="cuda")
        xc = torch.empty(B, H, W, 128, device="cuda", dtype=torch.bfloat16)
        xz = torch.empty(B, H, W, 256, device="cuda", dtype=torch.bfloat16)

        def run():
            L.call("fd_pw_dw3x3", L.FD_BF16, x.data_ptr(), 64, 0, 64, g.data_ptr(), be.data_ptr(), 1e-5, mod.data_ptr(), mod.data_ptr() + 256,
                   384, wpw.data_ptr(), 128, wm.data_ptr(), bdw.data_ptr(), 1, xc.data_ptr(), 128, 0, 128, xz.data_ptr(), 256, 128, B, H, W, s)
        t = timeit(run)
        print(f"pwdw in_proj {H}x{W} B={B}: {t:8.1f} us  chk={float(xc.float().abs().mean()):.5f} {float(xz[..., 128:].float().abs().mean()):.5f}", flush=True)


def bench_pwdwc(B, sizes=((512, 512),)):
    """fused LN -> 1x1 -> depthwise at Cdw = 64 / 128 / 192 without pass-through channels: base and per-chunk cost"""
    from founddiff_amd import _lib as L
    from founddiff_amd.engine import DAEngine
    s = torch.cuda.current_stream().cuda_stream
    for H, W in sizes:
        torch.manual_seed(0)
        x = torch.randn(B, H, W, 64, device="cuda").to(torch.bfloat16)
        mod = torch.randn(B, 384, device="cuda") * 0.5
        for cdw in (64, 128, 192):
            wpw = (torch.randn(cdw, 64, device="cuda") / 8).to(torch.bfloat16)
            wm = DAEngine._dw_masked((torch.randn(cdw, 9, device="cuda") / 3).t().contiguous())
            o = torch.empty(B, H, W, cdw, device="cuda", dtype=torch.bfloat16)

            def run():
                L.call("fd_pw_dw3x3", L.FD_BF16, x.data_ptr(), 64, 0, 64, None, None, 1e-6, mod.data_ptr(), mod.data_ptr() + 256,
                       384, wpw.data_ptr(), cdw, wm.data_ptr(), None, 0, o.data_ptr(), cdw, 0, 0, None, 0, 0, B, H, W, s)
            print(f"pwdw Cdw={cdw} Cz=0 {H}x{W} B={B}: {timeit(run):8.1f} us", flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("what")
    ap.add_argument("--batch", type=int, default=8)
    a = ap.parse_args()
    globals()["bench_" + a.what](a.batch)


if __name__ == "__main__":
    main()
