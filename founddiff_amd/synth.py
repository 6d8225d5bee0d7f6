"""Deterministic synthetic weights and CT phantoms.

There are no FoundDiff checkpoints or Mayo slices offline, so tests,
bench.py and the golden-vector generator all draw weights from this one function: values
depend only on (key name, shape, seed), never on dict order, so the reference model
(tests/golden/make_golden.py), the CPU oracle and the HIP engine see identical numbers.
"""
import math
import zlib

import numpy as np
import torch


def _gen(key, seed):
    g = torch.Generator(device="cpu")
    g.manual_seed((zlib.crc32(key.encode()) ^ (seed * 0x9E3779B1)) & 0x7FFFFFFF)
    return g


def synth_tensor(key, shape, seed=0, dtype=torch.float32):
    shape = tuple(int(s) for s in shape)
    g = _gen(key, seed)
    rn = lambda *s: torch.randn(*s, generator=g)
    ru = lambda *s: torch.rand(*s, generator=g)
    leaf = key.split(".")[-1]
    if dtype in (torch.int64, torch.int32):
        return torch.zeros(shape, dtype=dtype)
    if leaf == "running_var":
        v = 0.5 + ru(*shape)
    elif leaf == "running_mean":
        v = 0.1 * rn(*shape)
    elif leaf == "A_logs":
        n = shape[1]
        v = torch.log(torch.arange(1, n + 1, dtype=torch.float32))[None, :].repeat(shape[0], 1) \
            + 0.1 * rn(*shape)
    elif leaf == "Ds":
        v = 1.0 + 0.1 * rn(*shape)
    elif leaf == "dt_projs_bias":
        dt = torch.exp(ru(*shape) * (math.log(0.1) - math.log(0.001)) + math.log(0.001)).clamp(min=1e-4)
        v = dt + torch.log(-torch.expm1(-dt))
    elif leaf == "dt_projs_weight":
        v = (ru(*shape) * 2 - 1) * shape[-1] ** -0.5
    elif leaf == "temperature":
        v = 1.0 + 0.25 * rn(*shape)
    elif leaf == "prompt":
        v = ru(*shape)
    elif leaf == "g":                       # vanilla channel-LayerNorm gain (1,C,1,1)
        v = 1.0 + 0.1 * rn(*shape)
    elif "adaLN_modulation" in key:
        v = 0.05 * rn(*shape)               # zero-initialised in the reference (DADiff.py:473)
    elif len(shape) >= 2:
        fan_in = 1
        for s in shape[1:]:
            fan_in *= s
        v = rn(*shape) / math.sqrt(max(fan_in, 1))
    elif leaf == "weight":                  # 1-D norm gains
        v = 1.0 + 0.1 * rn(*shape)
    elif leaf == "bias":
        v = 0.05 * rn(*shape)
    elif len(shape) == 0:
        v = torch.tensor(1.0)
    else:
        v = 0.02 * rn(*shape)
    return v.to(dtype).reshape(shape)


def synth_state_dict(spec, seed=0):
    """spec: {key: shape} or {key: (shape, dtype_str)} -> {key: tensor}."""
    out = {}
    for k, s in spec.items():
        dt = torch.float32
        if isinstance(s, (tuple, list)) and len(s) == 2 and isinstance(s[1], str):
            s, dts = s
            dt = getattr(torch, dts)
        out[k] = synth_tensor(k, s, seed, dt)
    return out


def spec_of(state_dict):
    return {k: (tuple(v.shape), str(v.dtype).replace("torch.", "")) for k, v in state_dict.items()}


def ct_phantom(n, size, seed=10, sigma_hu=(25.0, 35.0, 50.0, 70.0, 110.0)):
    """n synthetic (NDCT, LDCT) slice pairs in the reference's [0,1] normalisation
    ((hu+1000)/3000 clipped, /root/reference/data/transforms.py:582-587): 8 seeded ellipses
    (air/water/soft tissue/bone) blurred by a 3x3 box, plus dose-dependent Gaussian noise.
    Returns float32 arrays (n,1,size,size)."""
    rng = np.random.RandomState(seed)
    yy, xx = np.mgrid[0:size, 0:size].astype(np.float32) / size - 0.5
    nd = np.empty((n, 1, size, size), np.float32)
    ld = np.empty_like(nd)
    for i in range(n):
        hu = np.full((size, size), -1000.0, np.float32)
        body = (xx / 0.42) ** 2 + (yy / 0.34) ** 2 < 1
        hu[body] = 0.0
        for _ in range(8):
            cx, cy = rng.uniform(-0.25, 0.25, 2)
            ax, ay = rng.uniform(0.03, 0.15, 2)
            val = rng.choice([-1000.0, 0.0, 60.0, 1000.0])
            m = ((xx - cx) / ax) ** 2 + ((yy - cy) / ay) ** 2 < 1
            hu[m & body] = val
        p = np.pad(hu, 1, mode="edge")
        hu = sum(p[a:a + size, b:b + size] for a in range(3) for b in range(3)) / 9.0
        s = sigma_hu[i % len(sigma_hu)]
        noisy = np.clip(hu + rng.normal(0, s, hu.shape).astype(np.float32), -1024, 2000)
        nd[i, 0] = np.clip((hu + 1000.0) / 3000.0, 0, 1)
        ld[i, 0] = np.clip((noisy + 1000.0) / 3000.0, 0, 1)
    return nd, ld
