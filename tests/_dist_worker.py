"""Worker of tests/test_gpu_dist.py: launched by torch.distributed.run (one rank per GPU, nccl = RCCL).
Runs the PRODUCT sampler through founddiff_amd.parallel.sample_volume and writes the gathered volume."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

TINY_CLIP = dict(layers=(2, 1, 1, 1), width=16, embed_dim=1024)


def build(dev, ancestral=False):
    from founddiff_amd import arch, synth
    from founddiff_amd.DADiff import ResidualDiffusion, UnetRes, load_weights
    spec = arch.da_unet_spec(32, (1, 2), prefix="model.unet0.", clip=TINY_CLIP)
    w = synth.synth_state_dict(spec, seed=0)
    net = UnetRes(dim=32, dim_mults=(1, 2), num_unet=1, condition=True, objective="pred_res",
                  test_res_or_noise="res", precision="bf16", clip_cfg=TINY_CLIP)
    # ancestral: sampling_timesteps == timesteps selects p_sample_loop (src/DADiff.py:1370); the reference's init()
    # hard-codes 1000 timesteps, so this is the full-length loop (on the tiny model)
    T, S = (1000, 1000) if ancestral else (1000, 3)
    dif = ResidualDiffusion(net, image_size=64, timesteps=T, sampling_timesteps=S, objective="pred_res",
                            loss_type="l2", condition=True, sum_scale=0.01, test_res_or_noise="res")
    load_weights(dif, w)
    dif = dif.to(dev)
    dif.init()
    return dif


def main():
    out_path, n = sys.argv[1], int(sys.argv[2])
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist.init_process_group("nccl", device_id=dev)
    from founddiff_amd import parallel, synth
    _, ld = synth.ct_phantom(n, 64, seed=10)
    vol = parallel.sample_volume(build(dev), torch.from_numpy(ld), world=world, rank=rank, noise_seed=100, batch=2)
    # the gather itself, ragged: rank r contributes r + 1 rows
    rag = parallel.gather_volume(torch.full((rank + 1, 1, 4, 4), float(rank), device=dev), world)
    assert rag.shape[0] == world * (world + 1) // 2
    dist.barrier()
    # the ancestral sampler over the same shards: step noise keyed by the global slice index
    vol_a = parallel.sample_volume(build(dev, ancestral=True), torch.from_numpy(ld), world=world, rank=rank, noise_seed=100, batch=2)
    # what RCCL itself saw: an all-reduce of ones = the number of ranks in the communicator, and the devices they ran on
    ones = torch.ones(1, device=dev)
    dist.all_reduce(ones)
    devs = [None] * world
    dist.all_gather_object(devs, torch.cuda.current_device())
    if rank == 0:
        torch.save({"ddim": vol.cpu(), "ancestral": vol_a.cpu(), "world": world, "allreduce_ones": float(ones.item()),
                    "devices": devs, "backend": dist.get_backend()}, out_path)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
