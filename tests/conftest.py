import json
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


class Golden:
    """One tests/golden/*.npz: arrays as torch tensors + the synthetic weights it was made with."""

    def __init__(self, name):
        z = np.load(os.path.join(GOLDEN, name + ".npz"))
        self.a = {k: z[k] for k in z.files}
        self.spec = None
        if "spec_json" in self.a:
            self.spec = json.loads(bytes(self.a.pop("spec_json")).decode())
            self.seed = int(self.a.pop("weight_seed"))

    def __getitem__(self, k):
        return torch.from_numpy(self.a[k])

    def weights(self, prefix=""):
        from founddiff_amd import synth
        spec = {k: v for k, v in self.spec.items() if k.startswith(prefix)}
        return synth.synth_state_dict(spec, self.seed)


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = Golden(name)
        return cache[name]
    return get


def rel_err(a, b):
    a, b = a.double(), b.double()
    return float((a - b).abs().max() / b.abs().max().clamp(min=1e-12))


# ---- bf16 kernel gates at 2 x the MEASURED error (3e-2 / 1e-2 blanket gates would not catch a 2 x precision
# regression in one kernel).  tests/golden/bf16_measured_errors.json holds the error each gated comparison had when the table was
# recorded on an MI355X (FD_RECORD_ERRORS=<path> python -m pytest tests -m gpu -k "resnet_block or mamba_block or selective_scan"
# rewrites <path>; commit it as the table).  A comparison missing from the table falls back to its blanket gate.
_MEASURED = os.path.join(GOLDEN, "bf16_measured_errors.json")
_measured_tab = None
_recorded = {}


def gate2x(key, err, blanket):
    """assert err < min(blanket, 2 x measured[key]); the message carries all three numbers."""
    global _measured_tab
    if "bf16" not in key:                  # fp32 comparisons keep their blanket gates (1e-4 / 2e-5: already at the arithmetic's noise)
        assert err < blanket, f"{key}: error {err:.3e} >= blanket gate {blanket:.1e}"
        return
    rec = os.environ.get("FD_RECORD_ERRORS")
    if rec:
        _recorded[key] = float(err)
        with open(rec, "w") as f:
            json.dump(dict(sorted(_recorded.items())), f, indent=1)
        assert err < blanket, f"{key}: error {err:.3e} above the blanket gate {blanket:.1e}"
        return
    if _measured_tab is None:
        _measured_tab = json.load(open(_MEASURED)) if os.path.exists(_MEASURED) else {}
    meas = _measured_tab.get(key)
    lim = blanket if meas is None else min(blanket, max(2.0 * meas, 1e-6))
    assert err < lim, (f"{key}: error {err:.3e} >= gate {lim:.3e} (= 2 x the {meas:.3e} measured when the table was recorded; "
                       f"blanket gate {blanket:.1e})" if meas is not None else f"{key}: error {err:.3e} >= blanket gate {blanket:.1e}")


# ---- the library's two builds (founddiff_amd/build.py): the kernel tests of tests/test_gpu_kernels.py run once per build.  `HB.t` is
# the torch dtype of "the 16-bit type" of the build under test; in the fp16 flavour the test module's 'bf16' mode IS the binary16
# build (same kernels, same entry points, FD_HALF_F16): _lib's default library and the engine's dtype table are swapped for the
# duration of the module, the fp32-storage / fp8 parametrisations are skipped (they belong to the default build), and every gate
# stays the bf16 one (binary16 carries three more bits; a kernel that passes with bf16's gate but mishandles the format -- the
# (1, 1) constant of a packed dot product was one, round 6 -- fails it by an order of magnitude).
class _HalfBuild:
    t = torch.bfloat16
    fp16 = False


HB = _HalfBuild()


def use_half_build(fp16):
    """point the 'bf16' mode of founddiff_amd at the binary16 build (True) or back at the default build (False)"""
    from founddiff_amd import _lib, engine
    if fp16 != HB.fp16:
        _lib.BF16, _lib.F16 = _lib.F16, _lib.BF16
        _lib.lib, _lib.check, _lib.call = _lib.BF16.lib, _lib.BF16.check, _lib.BF16.call
        HB.fp16, HB.t = fp16, (torch.float16 if fp16 else torch.bfloat16)
        for m in ("bf16", "fp8"):
            engine._T[m] = (engine._T[m][0], HB.t)


# ---- the CPU oracle's 50-step DDIM loop of the shipped architecture at a BASELINE size: minutes of CPU per call (50 forwards; ~35 s at
# 256x256, ~160 s at 512x512 on 32 host threads), asked for by several GPU tests with the same inputs -- computed once per session.
_ORACLE_LOOPS = {}


def oracle_ddim_loop(size, steps=50, noise_seed=7):
    """(weights, x_in [0, 1], x_T noise, oracle.sampler.ResidualOracle.sample(...)[-1]) for the synthetic-weight full model on the
    seed-10 CT phantom"""
    key = (size, steps, noise_seed)
    if key not in _ORACLE_LOOPS:
        from founddiff_amd import arch, synth
        from oracle import sampler
        spec = arch.da_unet_spec(64, (1, 2, 4, 8), prefix="model.unet0.")
        w = synth.synth_state_dict(spec, seed=0)
        _, ld = synth.ct_phantom(1, size, seed=10)
        x_in = torch.from_numpy(ld)
        noise = torch.randn(1, 1, size, size, generator=torch.Generator().manual_seed(noise_seed))
        torch.set_num_threads(min(32, torch.get_num_threads()))
        ref = sampler.ResidualOracle(w, prefix="model.unet0.", sampling_timesteps=steps).sample(x_in, noise)[-1]
        _ORACLE_LOOPS[key] = (w, x_in, noise, ref)
    return _ORACLE_LOOPS[key]
