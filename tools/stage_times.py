#!/usr/bin/env python3
"""In-situ time of every stage of one UNet forward (events recorded at the engine's probe points, so every
kernel runs with the cache state the real forward leaves it in), for several batch sizes: ms per SLICE per
stage.  Shows which levels profit from a batch small enough for the producer -> consumer tensors to stay in
the 256 MB Infinity Cache.  Development tool."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--batches", default="1,2,4,8")
    ap.add_argument("--reps", type=int, default=7)
    ap.add_argument("--precision", default="bf16")
    ap.add_argument("--detail", action="store_true", help="every probe point instead of per block")
    a = ap.parse_args()
    import bench
    from founddiff_amd import synth
    dev = torch.device("cuda")
    dif, _ = bench.build_model(dev, a.size, 50, a.precision)
    eng = dif._eng()
    table, tags = {}, None
    for B in [int(b) for b in a.batches.split(",")]:
        _, ld = synth.ct_phantom(B, a.size, seed=10)
        x_in = (torch.from_numpy(ld).to(dev) * 2 - 1).contiguous()
        img = (x_in + 0.1 * torch.randn_like(x_in)).contiguous()
        tb = torch.full((B,), 500.0, device=dev)
        eng.encode_condition(x_in)
        eng.forward(img, x_in, tb)
        runs = []
        for _ in range(a.reps):
            evs = []

            def hook(tag, t):
                e = torch.cuda.Event(enable_timing=True)
                e.record()
                evs.append((tag, e))
            e0 = torch.cuda.Event(enable_timing=True)
            eng.probe = hook
            e0.record()
            eng.forward(img, x_in, tb)
            eng.probe = None
            torch.cuda.synchronize()
            prev, row = e0, []
            for tag, e in evs:
                row.append(prev.elapsed_time(e))
                prev = e
            runs.append(row)
            tags = [t for t, _ in evs]
        med = [sorted(r[i] for r in runs)[len(runs) // 2] for i in range(len(tags))]
        table[B] = [m / B for m in med]
    if not a.detail:        # fold the probe points of a block into the block
        keys, folded = [], {B: [] for B in table}
        for i, t in enumerate(tags):
            k = t.split(".")[0]
            if not keys or keys[-1] != k:
                keys.append(k)
                for B in table:
                    folded[B].append(0.0)
            for B in table:
                folded[B][-1] += table[B][i]
        tags, table = keys, folded
    Bs = list(table)
    print("| stage | " + " | ".join(f"B={B} ms/slice" for B in Bs) + " |")
    print("|---|" + "---|" * len(Bs))
    for i, t in enumerate(tags):
        print(f"| {t} | " + " | ".join(f"{table[B][i]:.4f}" for B in Bs) + " |")
    print("| **total** | " + " | ".join(f"**{sum(table[B]):.3f}**" for B in Bs) + " |")


if __name__ == "__main__":
    main()
