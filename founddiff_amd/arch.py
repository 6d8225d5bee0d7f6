"""Parameter / buffer layout of the reference networks, restated as {key: shape} tables.

Checkpoints are the only thing that must be shared with the reference
(`checkpoints/<name>/sample/model-N.pt`, src/DADiff.py:1630-1636), so the key names and shapes
here are the contract; tests/test_arch_spec.py checks them against the layouts captured from
the reference itself (tests/golden/*.npz `spec_json`).
"""
import math

RN50 = dict(layers=(3, 4, 6, 3), width=64, embed_dim=1024)


def _mamba(spec, p, C, N, time_dim):
    D, R = 2 * C, math.ceil(C / 16)
    spec[p + "norm1.weight"] = (C,)
    spec[p + "norm1.bias"] = (C,)
    m = p + "mamba."
    spec[m + "x_proj_weight"] = (4, R + 2 * N, D)
    spec[m + "dt_projs_weight"] = (4, D, R)
    spec[m + "dt_projs_bias"] = (4, D)
    spec[m + "A_logs"] = (4 * D, N)
    spec[m + "Ds"] = (4 * D,)
    spec[m + "out_norm.weight"] = (D,)
    spec[m + "out_norm.bias"] = (D,)
    spec[m + "in_proj.weight"] = (2 * D, C)
    spec[m + "conv2d.weight"] = (D, 1, 3, 3)
    spec[m + "conv2d.bias"] = (D,)
    spec[m + "out_proj.weight"] = (C, D)
    spec[m + "attn.0.weight"] = (D, 256)
    spec[p + "adaLN_modulation.1.weight"] = (6 * C, time_dim)
    spec[p + "adaLN_modulation.1.bias"] = (6 * C,)
    a = p + "attn_blk."
    spec[a + "temperature"] = (C // 32, 1, 1)
    spec[a + "qkv.weight"] = (3 * C, C, 1, 1)
    spec[a + "qkv_dwconv.weight"] = (3 * C, 1, 3, 3)
    spec[a + "project_out.weight"] = (C, C, 1, 1)


def _resblock(spec, p, cin, cout):
    spec[p + "block1.proj.weight"] = (cout, cin, 3, 3)
    spec[p + "block1.proj.bias"] = (cout,)
    spec[p + "block1.norm.weight"] = (cout,)
    spec[p + "block1.norm.bias"] = (cout,)
    if cin != cout:
        spec[p + "res_conv.weight"] = (cout, cin, 1, 1)
        spec[p + "res_conv.bias"] = (cout,)


def _bn(spec, p, c):
    spec[p + "weight"] = (c,)
    spec[p + "bias"] = (c,)
    spec[p + "running_mean"] = (c,)
    spec[p + "running_var"] = (c,)
    spec[p + "num_batches_tracked"] = ((), "int64")


def clip_visual_spec(p, layers=RN50["layers"], width=RN50["width"], embed_dim=RN50["embed_dim"]):
    """CLIP ModifiedResNet (src/DACLIP.py:262-349) incl. AttentionPool2d (214-224)."""
    spec = {}
    w2 = width // 2
    for i, (ci, co) in enumerate(((3, w2), (w2, w2), (w2, width)), start=1):
        spec[p + f"conv{i}.weight"] = (co, ci, 3, 3)
        _bn(spec, p + f"bn{i}.", co)
    inpl = width
    for li, nb in enumerate(layers, start=1):
        planes = width * 2 ** (li - 1)
        for bi in range(nb):
            stride = 2 if (li > 1 and bi == 0) else 1
            q = p + f"layer{li}.{bi}."
            spec[q + "conv1.weight"] = (planes, inpl, 1, 1)
            _bn(spec, q + "bn1.", planes)
            spec[q + "conv2.weight"] = (planes, planes, 3, 3)
            _bn(spec, q + "bn2.", planes)
            spec[q + "conv3.weight"] = (planes * 4, planes, 1, 1)
            _bn(spec, q + "bn3.", planes * 4)
            if stride > 1 or inpl != planes * 4:
                spec[q + "downsample.0.weight"] = (planes * 4, inpl, 1, 1)
                _bn(spec, q + "downsample.1.", planes * 4)
            inpl = planes * 4
    E = width * 32
    spec[p + "attnpool.positional_embedding"] = ((224 // 32) ** 2 + 1, E)
    for n in ("k_proj", "q_proj", "v_proj"):
        spec[p + f"attnpool.{n}.weight"] = (E, E)
        spec[p + f"attnpool.{n}.bias"] = (E,)
    spec[p + "attnpool.c_proj.weight"] = (embed_dim, E)
    spec[p + "attnpool.c_proj.bias"] = (embed_dim,)
    return spec


def da_unet_spec(dim=64, dim_mults=(1, 2, 4, 8), channels=1, prefix="", clip=RN50, input_condition=False):
    """Live parameters of DADiff.Unet (src/DADiff.py:530-683); `prefix` e.g. 'model.unet0.'."""
    spec = {}
    p = prefix
    time_dim = dim * 4
    spec[p + "prompt"] = (1, time_dim)
    spec[p + "init_conv.weight"] = (dim, (3 if input_condition else 2) * channels, 7, 7)   # src/DADiff.py:553-555
    spec[p + "init_conv.bias"] = (dim,)
    spec[p + "time_mlp.1.weight"] = (time_dim, dim)
    spec[p + "time_mlp.1.bias"] = (time_dim,)
    spec[p + "time_mlp.3.weight"] = (time_dim, time_dim)
    spec[p + "time_mlp.3.bias"] = (time_dim,)
    spec[p + "text_mlp.0.weight"] = (time_dim, 1024)
    spec[p + "text_mlp.0.bias"] = (time_dim,)
    spec[p + "text_mlp.2.weight"] = (time_dim, time_dim)
    spec[p + "text_mlp.2.bias"] = (time_dim,)
    spec[p + "prompt_mlp.weight"] = (time_dim, time_dim)
    spec[p + "prompt_mlp.bias"] = (time_dim,)
    dims = [dim] + [dim * m for m in dim_mults]
    in_out = list(zip(dims[:-1], dims[1:]))
    n = len(in_out)
    for ind, (di, do) in enumerate(in_out):
        q = p + f"downs.{ind}."
        _resblock(spec, q + "0.", di, di)
        _mamba(spec, q + "1.", di, 4 if ind == 0 else int(4 * 2 ** ind), time_dim)
        if ind < n - 1:
            spec[q + "2.weight"] = (do, di, 4, 4)
        else:
            spec[q + "2.weight"] = (do, di, 3, 3)
        spec[q + "2.bias"] = (do,)
    mid = dims[-1]
    _resblock(spec, p + "mid_block.", mid, mid)
    _mamba(spec, p + "mid_attn.", mid, 32, time_dim)
    for ind, (di, do) in enumerate(reversed(in_out)):
        q = p + f"ups.{ind}."
        _resblock(spec, q + "0.", do + di, do)
        # d_state uses the hard-coded `3 - ind` (src/DADiff.py:663-666)
        _mamba(spec, q + "1.", do, 4 if (3 - ind) == 0 else int(4 * 2 ** (3 - ind)), time_dim)
        if ind < n - 1:
            spec[q + "2.1.weight"] = (di, do, 3, 3)
            spec[q + "2.1.bias"] = (di,)
        else:
            spec[q + "2.weight"] = (di, do, 3, 3)
            spec[q + "2.bias"] = (di,)
    _resblock(spec, p + "final_res_block.", dim * 2, dim)
    spec[p + "final_conv.weight"] = (channels, dim, 1, 1)
    spec[p + "final_conv.bias"] = (channels,)
    # DA-CLIP: live part only (visual tower + two heads)
    de = p + "dose_encoder."
    spec.update(clip_visual_spec(de + "clip_model.visual.", clip["layers"], clip["width"], clip["embed_dim"]))
    spec[de + "head1.0.weight"] = (1024, 1024)
    spec[de + "head1.0.bias"] = (1024,)
    spec[de + "head1.2.weight"] = (1024, 1024)
    spec[de + "head1.2.bias"] = (1024,)
    spec[de + "head2.0.weight"] = (1024, 1024)
    spec[de + "head2.0.bias"] = (1024,)
    spec[de + "head2.2.weight"] = (256, 1024)
    spec[de + "head2.2.bias"] = (256,)
    return spec


# keys a real checkpoint carries that the sampling path never reads
DEAD_KEY_MARKERS = (
    ".clip_model.transformer.", ".clip_model.token_embedding", ".clip_model.positional_embedding",
    ".clip_model.ln_final", ".clip_model.text_projection", ".clip_model.logit_scale",
    ".prompt_learner.", "perceploss.",
)


def is_dead_key(k, unet_prefix="model.unet0."):
    if k.startswith(unet_prefix + "clip_model.") or k.startswith("unet0.clip_model."):
        return True            # the second, never-used CLIP (src/DADiff.py:590)
    return any(m in k for m in DEAD_KEY_MARKERS)


def _vresblock(spec, p, cin, cout, time_dim):
    spec[p + "mlp.1.weight"] = (2 * cout, time_dim)
    spec[p + "mlp.1.bias"] = (2 * cout,)
    for blk, ci in (("block1.", cin), ("block2.", cout)):
        spec[p + blk + "proj.weight"] = (cout, ci, 3, 3)
        spec[p + blk + "proj.bias"] = (cout,)
        spec[p + blk + "norm.weight"] = (cout,)
        spec[p + blk + "norm.bias"] = (cout,)
    if cin != cout:
        spec[p + "res_conv.weight"] = (cout, cin, 1, 1)
        spec[p + "res_conv.bias"] = (cout,)


def _vlinattn(spec, p, dim, hidden=128):
    spec[p + "fn.norm.g"] = (1, dim, 1, 1)
    spec[p + "fn.fn.to_qkv.weight"] = (3 * hidden, dim, 1, 1)
    spec[p + "fn.fn.to_out.0.weight"] = (dim, hidden, 1, 1)
    spec[p + "fn.fn.to_out.0.bias"] = (dim,)
    spec[p + "fn.fn.to_out.1.g"] = (1, dim, 1, 1)


def vanilla_unet_spec(dim=32, dim_mults=(1, 2), channels=1, prefix=""):
    """Parameters of denoising_diffusion_pytorch.Unet (src/denoising_diffusion_pytorch.py:283-369)."""
    spec, p, td = {}, prefix, dim * 4
    spec[p + "init_conv.weight"] = (dim, channels, 7, 7)
    spec[p + "init_conv.bias"] = (dim,)
    spec[p + "time_mlp.1.weight"] = (td, dim)
    spec[p + "time_mlp.1.bias"] = (td,)
    spec[p + "time_mlp.3.weight"] = (td, td)
    spec[p + "time_mlp.3.bias"] = (td,)
    dims = [dim] + [dim * m for m in dim_mults]
    in_out = list(zip(dims[:-1], dims[1:]))
    n = len(in_out)
    for i, (di, do) in enumerate(in_out):
        q = p + f"downs.{i}."
        _vresblock(spec, q + "0.", di, di, td)
        _vresblock(spec, q + "1.", di, di, td)
        _vlinattn(spec, q + "2.", di)
        spec[q + "3.weight"] = (do, di, 4, 4) if i < n - 1 else (do, di, 3, 3)
        spec[q + "3.bias"] = (do,)
    mid = dims[-1]
    _vresblock(spec, p + "mid_block1.", mid, mid, td)
    spec[p + "mid_attn.fn.norm.g"] = (1, mid, 1, 1)
    spec[p + "mid_attn.fn.fn.to_qkv.weight"] = (384, mid, 1, 1)
    spec[p + "mid_attn.fn.fn.to_out.weight"] = (mid, 128, 1, 1)
    spec[p + "mid_attn.fn.fn.to_out.bias"] = (mid,)
    _vresblock(spec, p + "mid_block2.", mid, mid, td)
    for i, (di, do) in enumerate(reversed(in_out)):
        q = p + f"ups.{i}."
        _vresblock(spec, q + "0.", do + di, do, td)
        _vresblock(spec, q + "1.", do + di, do, td)
        _vlinattn(spec, q + "2.", do)
        if i < n - 1:
            spec[q + "3.1.weight"] = (di, do, 3, 3)
            spec[q + "3.1.bias"] = (di,)
        else:
            spec[q + "3.weight"] = (di, do, 3, 3)
            spec[q + "3.bias"] = (di,)
    _vresblock(spec, p + "final_res_block.", dim * 2, dim, td)
    spec[p + "final_conv.weight"] = (channels, dim, 1, 1)
    spec[p + "final_conv.bias"] = (channels,)
    return spec
