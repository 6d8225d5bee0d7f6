"""Multi-GPU sampling: slices are independent units (each slice's reverse process depends only on
its own x_input / x_T / noise, src/DADiff.py:1276-1365; GroupNorm / LayerNorm are per-sample and
the RN50 BatchNorm runs in eval mode), so a volume shards over ranks by slice index with NO
collective on the data path.  The only exchange is one all-gather (RCCL over xGMI on GPUs,
gloo in the CPU tests) that reassembles the output volume.
"""
import torch


def shard_range(n, world, rank):
    """Contiguous block split of range(n): rank r gets [lo, hi); ragged tails go to the last ranks."""
    per = (n + world - 1) // world
    lo = min(rank * per, n)
    return lo, min(lo + per, n)


def gather_volume(local, world, n_total=None):
    """All-gather equal-or-ragged per-rank blocks (B_local,1,H,W) into the full volume on every
    rank.  Ragged tails are padded to the largest block and masked on unpack."""
    if world == 1:
        return local
    import torch.distributed as dist
    per = torch.tensor([local.shape[0]], device=local.device, dtype=torch.int64)
    sizes = [torch.zeros_like(per) for _ in range(world)]
    dist.all_gather(sizes, per)
    sizes = [int(s.item()) for s in sizes]
    mx = max(sizes)
    pad = local
    if local.shape[0] < mx:
        pad = torch.cat([local, local.new_zeros((mx - local.shape[0],) + tuple(local.shape[1:]))], 0)
    out = torch.empty((world * mx,) + tuple(local.shape[1:]), device=local.device, dtype=local.dtype)
    dist.all_gather_into_tensor(out, pad.contiguous())
    if all(s == mx for s in sizes):
        vol = out
    else:
        vol = torch.cat([out[r * mx:r * mx + sizes[r]] for r in range(world)], 0)
    if n_total is not None:
        assert vol.shape[0] == n_total
    return vol


def sample_volume(diffusion, ldct, world=1, rank=0, noise_seed=0, batch=8, sampler_kwargs=None):
    """Denoise a whole volume ldct (N,1,H,W in [0,1], same tensor on every rank) with `diffusion`
    (a founddiff_amd.DADiff.ResidualDiffusion living on this rank's device).  x_T noise is keyed
    by the GLOBAL slice index, and so is the ancestral sampler's per-step noise (`slice_seeds`:
    fd_sched.hip's counter-based stream), so the result is invariant to `world` for both samplers.
    Returns the (N,1,H,W) volume on every rank."""
    n = ldct.shape[0]
    lo, hi = shard_range(n, world, rank)
    dev = next(diffusion.parameters()).device
    outs = []
    for s in range(lo, hi, batch):
        e = min(s + batch, hi)
        x = ldct[s:e].to(dev)
        nz = torch.stack([torch.randn(x.shape[1:], generator=torch.Generator().manual_seed(noise_seed + i))
                          for i in range(s, e)]).to(dev)
        seeds = torch.arange(s, e, dtype=torch.int64) + int(noise_seed)
        outs.append(diffusion.sample([x], batch_size=e - s, noise=nz, slice_seeds=seeds, **(sampler_kwargs or {}))[-1])
    local = torch.cat(outs, 0) if outs else ldct.new_zeros((0,) + tuple(ldct.shape[1:])).to(dev)
    return gather_volume(local, world, n)
