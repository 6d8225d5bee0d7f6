#!/usr/bin/env python3
"""The 16-bit storage formats against the CPU oracle over a whole DDIM loop: precision='bf16' and precision='fp16' (the same
kernels on the library's two builds), each with and without the precision tail, `--size` x `--size`, `--steps` steps, full
architecture; and the time of a batch-8 loop in both.   python tools/probes/fp16_drift.py [--size 256] [--steps 50]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from founddiff_amd import arch, synth  # noqa: E402
from founddiff_amd.DADiff import ResidualDiffusion, UnetRes, load_weights  # noqa: E402
from oracle import sampler  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=256)
ap.add_argument("--steps", type=int, default=50)
ap.add_argument("--modes", default="bf16,fp16")
ap.add_argument("--time-batch", type=int, default=8)
ap.add_argument("--weight-seed", type=int, default=0)
ap.add_argument("--phantom-seed", type=int, default=10)
ap.add_argument("--noise-seed", type=int, default=7)
ap.add_argument("--tails", default="1:2,0:2", help="tail steps : outer levels on the tail engine, comma separated")
a = ap.parse_args()
S, N = a.steps, a.size
spec = arch.da_unet_spec(64, (1, 2, 4, 8), prefix="model.unet0.")
w = synth.synth_state_dict(spec, seed=a.weight_seed)
_, ld = synth.ct_phantom(1, N, seed=a.phantom_seed)
x_in = torch.from_numpy(ld)
noise = torch.randn(1, 1, N, N, generator=torch.Generator().manual_seed(a.noise_seed))
torch.set_num_threads(min(32, os.cpu_count()))
t0 = time.time()
ref = sampler.ResidualOracle(w, prefix="model.unet0.", sampling_timesteps=S).sample(x_in, noise)[-1]
print(f"weights seed {a.weight_seed}, phantom seed {a.phantom_seed}, x_T seed {a.noise_seed}; oracle: {time.time() - t0:.0f} s", flush=True)


def model(prec, tail, levels):
    os.environ["FOUNDDIFF_FINAL_OUTER_LEVELS"] = str(levels)
    net = UnetRes(dim=64, dim_mults=(1, 2, 4, 8), num_unet=1, condition=True, objective="pred_res", test_res_or_noise="res", precision=prec)
    dif = ResidualDiffusion(net, image_size=N, timesteps=1000, sampling_timesteps=S, objective="pred_res", loss_type="l2",
                            condition=True, sum_scale=0.01, test_res_or_noise="res", final_fp32_steps=tail)
    load_weights(dif, w)
    dif = dif.to("cuda")
    dif.init()
    return dif


print("| mode | tail steps : outer levels | L2 vs oracle | PSNR dB | max-rel | batch-%d loop ms | L2 of the same slice inside that batch |" % a.time_batch)
print("|---|---|---|---|---|---|---|")
for prec in a.modes.split(","):
    for tl in a.tails.split(","):
        tail, levels = (int(v) for v in tl.split(":"))
        dif = model(prec, tail, levels)
        out = dif.sample([x_in.cuda()], batch_size=1, noise=noise.cuda())[-1].float().cpu()
        l2 = float((out - ref).norm() / ref.norm())
        mse = float(((out - ref) ** 2).mean())
        mr = float((out - ref).abs().max() / ref.abs().max())
        xb = x_in.cuda().repeat(a.time_batch, 1, 1, 1)
        nb = noise.cuda().repeat(a.time_batch, 1, 1, 1)
        for _ in range(2):
            ob = dif.sample([xb], batch_size=a.time_batch, noise=nb)[-1]
        lb = [float((ob[i:i + 1].float().cpu() - ref).norm() / ref.norm()) for i in range(a.time_batch)]
        torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(3):
            dif.sample([xb], batch_size=a.time_batch, noise=nb)
        torch.cuda.synchronize()
        ms = (time.time() - t0) / 3 * 1e3
        import math
        print(f"| {prec} | {tail} : {levels} | {l2:.3e} | {10 * math.log10(1.0 / mse):.1f} | {mr:.2e} | {ms:.1f} | {min(lb):.3e} .. {max(lb):.3e} |", flush=True)
        del dif
        torch.cuda.empty_cache()
