"""Multi-process (gloo, world_size 2) checks of the slice sharding + output all-gather, and that a
sharded run equals the single-process run exactly (slices are independent, noise is keyed by the
GLOBAL slice index).  CPU only: the diffusion object is a stand-in with the product's interface."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp


class _FakeDiffusion(torch.nn.Module):
    """sample([x], batch_size, noise) -> [x_T, f(x, noise)] with a per-slice deterministic f."""

    def __init__(self):
        super().__init__()
        self.p = torch.nn.Parameter(torch.zeros(1))

    def sample(self, x_input, batch_size, noise=None, **k):
        x = x_input[0]
        return [x + 0.1 * noise, torch.tanh(x * 3 - noise * 0.25) * x.flatten(1).mean(1).view(-1, 1, 1, 1)]


class _FakeAncestral(_FakeDiffusion):
    """An ancestral-style stand-in: a few steps, each adding the keyed per-slice step noise the product's
    p_sample_loop draws (oracle/keyed_noise.py restates fd_sched.hip's stream) -- the result depends on slice_seeds."""

    def sample(self, x_input, batch_size, noise=None, slice_seeds=None, **k):
        from oracle import keyed_noise
        x = x_input[0]
        assert slice_seeds is not None and len(slice_seeds) == x.shape[0]
        img = x + 0.1 * noise
        for t in (3, 2, 1):
            nz = torch.stack([torch.from_numpy(keyed_noise.keyed_normal(int(sd), t, x[0].numel())).view_as(x[0])
                              for sd in slice_seeds])
            img = 0.9 * img + 0.1 * torch.tanh(x - img) + 0.05 * nz
        return [x + 0.1 * noise, img]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n, q, ancestral=False):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from founddiff_amd import parallel
    torch.manual_seed(0)
    vol = torch.rand(n, 1, 8, 8)
    dif = _FakeAncestral() if ancestral else _FakeDiffusion()
    out = parallel.sample_volume(dif, vol, world=world, rank=rank, noise_seed=100, batch=2)
    lo, hi = parallel.shard_range(n, world, rank)
    q.put((rank, lo, hi, out))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n,ancestral", [(6, False), (5, False), (1, False), (5, True)])
def test_sharded_equals_single(n, ancestral):
    """ancestral=True: the sampler's per-step noise is keyed by the GLOBAL slice index (slice_seeds), so a sharded
    BASELINE configs[3] volume equals the single-process one whatever the batch composition."""
    from founddiff_amd import parallel
    torch.manual_seed(0)
    vol = torch.rand(n, 1, 8, 8)
    ref = parallel.sample_volume(_FakeAncestral() if ancestral else _FakeDiffusion(), vol, world=1, rank=0, noise_seed=100, batch=4)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n, q, ancestral)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    covered = sorted((lo, hi) for _, lo, hi, _ in res)
    assert covered[0][0] == 0 and covered[-1][1] == n and covered[0][1] == covered[1][0]
    for _, _, _, out in res:
        assert out.shape == ref.shape
        assert torch.equal(out, ref)          # bitwise: independent of the world size


def test_shard_range_partitions():
    from founddiff_amd import parallel
    for n in (0, 1, 7, 64, 65):
        for w in (1, 2, 3, 8):
            parts = [parallel.shard_range(n, w, r) for r in range(w)]
            flat = [i for lo, hi in parts for i in range(lo, hi)]
            assert flat == list(range(n))


def test_keyed_noise_oracle_known_answers():
    """oracle/keyed_noise.py: Philox4x32-10 against the published known-answer vectors of the Random123 library
    (kat_vectors: zero and all-ones counter / key), and the Box-Muller stream's first two moments."""
    import numpy as np
    from oracle import keyed_noise as kn
    r = kn.philox4x32_10([0], [0], [0], [0], 0, 0)
    assert [int(v[0]) for v in r] == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    r = kn.philox4x32_10([0xffffffff], [0xffffffff], [0xffffffff], [0xffffffff], 0xffffffff, 0xffffffff)
    assert [int(v[0]) for v in r] == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    r = kn.philox4x32_10([0x243f6a88], [0x85a308d3], [0x13198a2e], [0x03707344], 0xa4093822, 0x299f31d0)
    assert [int(v[0]) for v in r] == [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]
    z = kn.keyed_normal(12345, 7, 1 << 18)
    assert abs(float(z.mean())) < 1e-2 and abs(float(z.std()) - 1) < 1e-2
    assert not np.array_equal(z, kn.keyed_normal(12345, 8, 1 << 18)) and not np.array_equal(z, kn.keyed_normal(12346, 7, 1 << 18))
    assert np.array_equal(z[:1000], kn.keyed_normal(12345, 7, 1000))      # a prefix: position-keyed, not length-keyed
