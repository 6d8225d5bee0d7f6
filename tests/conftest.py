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
