#!/usr/bin/env python3
"""Experiments on the two-stream production mode, all inside ONE process / one box:
  (the phase-offset experiment of round 4 -- sub-batch 1 started a fraction of a forward late through fd_stream_delay:
   12.17 / 12.11 / 11.98 / 12.02 against 12.15 slices/s, profiles/r04_overlap_experiments.txt -- was removed with that entry point in round 5)
  * CU-masked streams: each sub-batch's stream restricted to a subset of the 256 CUs (hipExtStreamCreateWithCUMask)
  * stream counts 1 / 2 / 4
Every configuration runs the real `ResidualDiffusion.sample()` of bench.py's workload (B = 16, 512x512, 50-step DDIM) and
reports slices/s.        usage: overlap_experiments.py [reps] [what,...]   what in {mask, streams}"""
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
from founddiff_amd import synth  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
what = set((sys.argv[2] if len(sys.argv) > 2 else "mask,streams").split(","))
dev = torch.device("cuda")
B = 16
dif, _ = bench.build_model(dev)
_, ld = synth.ct_phantom(B, 512, seed=10)
x = torch.from_numpy(ld).to(dev)
noise = torch.randn(B, 1, 512, 512, device=dev)


def run(label, **kw):
    for k, v in kw.items():
        setattr(dif, k, v)
    out = dif.sample([x], batch_size=B, noise=noise)[-1]      # warm-up / capture
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        out = dif.sample([x], batch_size=B, noise=noise)[-1]
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    best, mean = min(ts), sum(ts) / len(ts)
    print(json.dumps({"config": label, "slices_per_s_best": round(B / best, 3), "slices_per_s_mean": round(B / mean, 3),
                      "checksum": float(out.double().sum())}), flush=True)
    return out


ref = run("default: 2 streams", streams=2)
if "streams" in what:
    run("1 stream (B = 16)", streams=1)
    run("4 streams (B = 4 each)", streams=4)
    dif.streams = 2
if "mask" in what:
    hip = C.CDLL("libamdhip64.so")

    def masked_stream(bits):
        words = (C.c_uint32 * 8)(*[(bits >> (32 * i)) & 0xFFFFFFFF for i in range(8)])
        st = C.c_void_p()
        rc = hip.hipExtStreamCreateWithCUMask(C.byref(st), 8, words)
        assert rc == 0, rc
        return torch.cuda.ExternalStream(st.value, device=dev)

    full = (1 << 256) - 1
    low, high = (1 << 128) - 1, full ^ ((1 << 128) - 1)
    even = sum(1 << i for i in range(0, 256, 2))
    # bit i -> (XCD i % 8, CU i / 8) on this part as far as the runtime documents it: "low" = half of every XCD
    xcd_lo = sum(1 << i for i in range(256) if (i % 8) < 4)          # four whole XCDs per stream
    for label, m0, m1 in (("masks: bits 0-127 | 128-255", low, high), ("masks: even | odd bits", even, full ^ even),
                          ("masks: XCD 0-3 | XCD 4-7 (bit % 8)", xcd_lo, full ^ xcd_lo),
                          ("masks: all | all (external streams)", full, full),
                          ("masks: bits 0-191 | 64-255 (overlap 128)", (1 << 192) - 1, full ^ ((1 << 64) - 1))):
        dif._side_streams = {0: masked_stream(m0), 1: masked_stream(m1)}
        o = run(label, streams=2)
        print("   identical to default:", bool(torch.equal(o, ref)), flush=True)
