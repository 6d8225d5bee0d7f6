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


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from founddiff_amd import parallel
    torch.manual_seed(0)
    vol = torch.rand(n, 1, 8, 8)
    out = parallel.sample_volume(_FakeDiffusion(), vol, world=world, rank=rank, noise_seed=100, batch=2)
    lo, hi = parallel.shard_range(n, world, rank)
    q.put((rank, lo, hi, out))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n", [6, 5, 1])
def test_sharded_equals_single(n):
    from founddiff_amd import parallel
    torch.manual_seed(0)
    vol = torch.rand(n, 1, 8, 8)
    ref = parallel.sample_volume(_FakeDiffusion(), vol, world=1, rank=0, noise_seed=100, batch=4)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n, q)) for r in range(2)]
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
