"""ORACLE (test infrastructure, not product code): CPU fp32 restatement of the denoisers.

Functional PyTorch-CPU code over a *reference-layout* state_dict (`sd`, key names exactly as
`ResidualDiffusion(...).state_dict()` prints them minus a caller-chosen prefix), so that the
same weights can be loaded into the reference (tests/golden/make_golden.py) and into this
oracle, and the outputs compared.  Follows:

  DA-conditioned U-Net      /root/reference/src/DADiff.py:139-154 (WS conv), 173-185 (sin emb),
                            213-229 (Block), 252-285 (TransposedAttention), 397-430 (ResnetBlock),
                            450-488 (modulate, Mamba_block), 685-740 (Unet.forward)
  SS2D mixer                /root/reference/src/emamba2.py:182-262 (scan/merge index maps),
                            295-367 (cross_selective_scan), 713-751 (SS2D.forward)
  DA-CLIP visual + heads    /root/reference/src/DACLIP.py:198-211, 226-259, 329-349, 1189-1221
  vanilla U-Net             /root/reference/src/denoising_diffusion_pytorch.py:183-279, 371-410

Pinned by tests/golden/*.npz captured from the reference itself (tests/test_oracle_golden.py).
The selective scan op is third-party and un-pinned by the reference: see oracle/csrc/scan_oracle.c.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
"""
import ctypes
import math
import os

import torch
import torch.nn.functional as F

_LIB = None


def _lib():
    global _LIB
    if _LIB is None:
        from . import build as _b
        _LIB = ctypes.CDLL(_b.build())
        _LIB.fd_oracle_selective_scan.restype = None
    return _LIB


def selective_scan(u, delta, A, B, C, D, delta_bias, softplus=True):
    """u, delta (b,KD,L); A (KD,N); B, C (b,K,N,L); D, delta_bias (KD) -> y (b,KD,L).  C kernel."""
    u = u.contiguous().float()
    delta = delta.contiguous().float()
    A = A.contiguous().float()
    B = B.contiguous().float()
    C = C.contiguous().float()
    D = D.contiguous().float()
    delta_bias = delta_bias.contiguous().float()
    b, KD, L = u.shape
    K, N = B.shape[1], A.shape[1]
    assert N <= 256
    y = torch.empty_like(u)
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    _lib().fd_oracle_selective_scan(p(u), p(delta), p(A), p(B), p(C), p(D), p(delta_bias),
                                    ctypes.c_int(1 if softplus else 0), p(y), ctypes.c_int(b),
                                    ctypes.c_int(KD), ctypes.c_int(K), ctypes.c_int(N),
                                    ctypes.c_long(L))
    return y


def selective_scan_f64(u, delta, A, B, C, D, delta_bias, softplus=True):
    """Same arguments (fp32 tensors), state and arithmetic in fp64 -> y float64.  C kernel."""
    u, delta, A, B, C, D, delta_bias = [t.contiguous().float() for t in (u, delta, A, B, C, D, delta_bias)]
    b, KD, L = u.shape
    K, N = B.shape[1], A.shape[1]
    assert N <= 256
    y = torch.empty(u.shape, dtype=torch.float64)
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    fn = _lib().fd_oracle_selective_scan_f64
    fn.restype = None
    fn(p(u), p(delta), p(A), p(B), p(C), p(D), p(delta_bias), ctypes.c_int(1 if softplus else 0), p(y),
       ctypes.c_int(b), ctypes.c_int(KD), ctypes.c_int(K), ctypes.c_int(N), ctypes.c_long(L))
    return y


def selective_scan_torch(u, delta, A, B, C, D, delta_bias, softplus=True):
    """Same recurrence in plain torch (slow; cross-checks the C kernel on small cases)."""
    b, KD, L = u.shape
    K = B.shape[1]
    Dg = KD // K
    dt = delta + delta_bias[None, :, None]
    if softplus:
        dt = F.softplus(dt)
    Bx = B.repeat_interleave(Dg, dim=1)
    Cx = C.repeat_interleave(Dg, dim=1)
    h = torch.zeros(b, KD, A.shape[1])
    ys = []
    for t in range(L):
        h = torch.exp(dt[:, :, t, None] * A[None]) * h + dt[:, :, t, None] * Bx[..., t] * u[:, :, t, None]
        ys.append((h * Cx[..., t]).sum(-1))
    return torch.stack(ys, -1) + D[None, :, None] * u


class SD:
    """Prefix view over a state_dict."""

    def __init__(self, sd, prefix=""):
        self.sd, self.prefix = sd, prefix

    def __getitem__(self, k):
        return self.sd[self.prefix + k]

    def has(self, k):
        return (self.prefix + k) in self.sd

    def sub(self, p):
        return SD(self.sd, self.prefix + p)


# ------------------------------------------------------------------ shared pieces
def ws_weight(w, eps=1e-5):
    """Weight standardisation, biased variance, fp32 eps (DADiff.py:145-152)."""
    mean = w.mean(dim=(1, 2, 3), keepdim=True)
    var = w.var(dim=(1, 2, 3), unbiased=False, keepdim=True)
    return (w - mean) * torch.rsqrt(var + eps)


def sinusoidal_emb(x, dim):
    half = dim // 2
    e = math.log(10000) / (half - 1)
    f = torch.exp(torch.arange(half, dtype=torch.float32) * -e)
    a = x[:, None].float() * f[None, :]
    return torch.cat((a.sin(), a.cos()), dim=-1)


def time_mlp(sd, time, dim):
    h = sinusoidal_emb(time, dim)
    h = F.linear(h, sd["1.weight"], sd["1.bias"])
    h = F.gelu(h)
    return F.linear(h, sd["3.weight"], sd["3.bias"])


def block(sd, x, groups=8, scale_shift=None):
    x = F.conv2d(x, ws_weight(sd["proj.weight"]), sd["proj.bias"], padding=1)
    x = F.group_norm(x, groups, sd["norm.weight"], sd["norm.bias"], eps=1e-5)
    if scale_shift is not None:
        scale, shift = scale_shift
        x = x * (scale + 1) + shift
    return F.silu(x)


def da_resnet_block(sd, x, groups=8):
    """DADiff ResnetBlock: ONE Block + 1x1 res_conv, no time conditioning (DADiff.py:397-430)."""
    h = block(sd.sub("block1."), x, groups)
    if sd.has("res_conv.weight"):
        return h + F.conv2d(x, sd["res_conv.weight"], sd["res_conv.bias"])
    return h + x


# ------------------------------------------------------------------ SS2D
def efficient_scan(x):
    """(B,C,H,W) -> (B,4,C,L): 4 stride-2 sub-grids; 0/2 row-major, 1/3 column-major."""
    Bn, C, H, W = x.shape
    if W % 2:
        x = F.pad(x, (0, 1, 0, 0))
    if H % 2:
        x = F.pad(x, (0, 0, 0, 1))
    xt = x.transpose(2, 3)
    return torch.stack([
        x[:, :, 0::2, 0::2].reshape(Bn, C, -1),
        xt[:, :, 0::2, 1::2].reshape(Bn, C, -1),
        x[:, :, 0::2, 1::2].reshape(Bn, C, -1),
        xt[:, :, 1::2, 1::2].reshape(Bn, C, -1),
    ], dim=1)


def efficient_merge(ys, H, W):
    """(B,4,C,L) -> (B,C,H*W), inverse of efficient_scan (crops the odd-size padding)."""
    Bn, K, C, L = ys.shape
    h2, w2 = math.ceil(H / 2), math.ceil(W / 2)
    y = ys.new_empty(Bn, C, 2 * h2, 2 * w2)
    y[:, :, 0::2, 0::2] = ys[:, 0].reshape(Bn, C, h2, w2)
    y[:, :, 1::2, 0::2] = ys[:, 1].reshape(Bn, C, w2, h2).transpose(2, 3)
    y[:, :, 0::2, 1::2] = ys[:, 2].reshape(Bn, C, h2, w2)
    y[:, :, 1::2, 1::2] = ys[:, 3].reshape(Bn, C, w2, h2).transpose(2, 3)
    return y[:, :, :H, :W].reshape(Bn, C, H * W)


def cross_selective_scan(sd, x, scan_fn=None):
    """x (B,D,H,W) after dwconv+SiLU -> (B,H,W,D) after out_norm (emamba2.py:295-367)."""
    scan_fn = scan_fn or selective_scan
    Bn, D, H, W = x.shape
    xw, dtw, dtb = sd["x_proj_weight"], sd["dt_projs_weight"], sd["dt_projs_bias"]
    A_logs, Ds = sd["A_logs"], sd["Ds"]
    N = A_logs.shape[1]
    K, _, R = dtw.shape
    xs = efficient_scan(x)
    L = xs.shape[-1]
    x_dbl = torch.einsum("bkdl,kcd->bkcl", xs, xw)
    dts, Bs, Cs = torch.split(x_dbl, [R, N, N], dim=2)
    dts = torch.einsum("bkrl,kdr->bkdl", dts, dtw)
    ys = scan_fn(xs.reshape(Bn, -1, L), dts.reshape(Bn, -1, L), -torch.exp(A_logs.float()),
                 Bs.contiguous(), Cs.contiguous(), Ds.float(), dtb.reshape(-1).float(), True)
    y = efficient_merge(ys.view(Bn, K, -1, L), H, W)
    y = y.transpose(1, 2)
    y = F.layer_norm(y, (D,), sd["out_norm.weight"], sd["out_norm.bias"], eps=1e-5)
    return y.reshape(Bn, H, W, D)


def ss2d(sd, x, c, scan_fn=None):
    """x (B,H,W,C) channel-last, c (B,1,256) -> (B,H,W,C) (emamba2.py:713-751)."""
    local = F.silu(F.linear(c, sd["attn.0.weight"]))           # (B,1,2C)
    xz = F.linear(x, sd["in_proj.weight"])
    xi, z = xz.chunk(2, dim=-1)
    z = F.silu(z)
    xi = xi.permute(0, 3, 1, 2)
    xi = F.silu(F.conv2d(xi, sd["conv2d.weight"], sd["conv2d.bias"], padding=1, groups=xi.shape[1]))
    y = cross_selective_scan(sd, xi, scan_fn)
    y = y * z
    return F.linear(y + local.unsqueeze(1), sd["out_proj.weight"])


def transposed_attention(sd, x):
    """Restormer channel attention: softmax over a (C/heads x C/heads) Gram (DADiff.py:263-285)."""
    b, C, H, W = x.shape
    heads = sd["temperature"].shape[0]
    qkv = F.conv2d(x, sd["qkv.weight"])
    qkv = F.conv2d(qkv, sd["qkv_dwconv.weight"], padding=1, groups=3 * C)
    q, k, v = qkv.chunk(3, dim=1)
    q = q.reshape(b, heads, C // heads, H * W)
    k = k.reshape(b, heads, C // heads, H * W)
    v = v.reshape(b, heads, C // heads, H * W)
    q = F.normalize(q, dim=-1)
    k = F.normalize(k, dim=-1)
    attn = (q @ k.transpose(-2, -1)) * sd["temperature"]
    attn = attn.softmax(dim=-1)
    out = (attn @ v).reshape(b, C, H, W)
    return F.conv2d(out, sd["project_out.weight"])


def mamba_block(sd, x, c, t, scan_fn=None):
    """adaLN-gated SS2D + channel attention (DADiff.py:477-488).  x NCHW in/out."""
    C = x.shape[1]
    x = x.permute(0, 2, 3, 1)
    mod = F.linear(F.silu(t), sd["adaLN_modulation.1.weight"], sd["adaLN_modulation.1.bias"])
    sh1, sc1, g1, sh2, sc2, g2 = [m[:, None, None, :] for m in mod.chunk(6, dim=1)]
    h = F.layer_norm(x, (C,), sd["norm1.weight"], sd["norm1.bias"], eps=1e-5) * (1 + sc1) + sh1
    x = x + g1 * ss2d(sd.sub("mamba."), h, c, scan_fn)
    h = F.layer_norm(x, (C,), None, None, eps=1e-6) * (1 + sc2) + sh2
    a = transposed_attention(sd.sub("attn_blk."), h.permute(0, 3, 1, 2))
    x = x + g2 * a.permute(0, 2, 3, 1)
    return x.permute(0, 3, 1, 2)


# ------------------------------------------------------------------ DA-CLIP (visual + heads)
def _bn(sd, x):
    return F.batch_norm(x, sd["running_mean"], sd["running_var"], sd["weight"], sd["bias"],
                        training=False, eps=1e-5)


def _bottleneck(sd, x, stride):
    out = F.relu(_bn(sd.sub("bn1."), F.conv2d(x, sd["conv1.weight"])))
    out = F.relu(_bn(sd.sub("bn2."), F.conv2d(out, sd["conv2.weight"], padding=1)))
    if stride > 1:
        out = F.avg_pool2d(out, stride)
    out = _bn(sd.sub("bn3."), F.conv2d(out, sd["conv3.weight"]))
    idn = x
    if sd.has("downsample.0.weight"):
        idn = F.avg_pool2d(x, stride) if stride > 1 else x
        idn = _bn(sd.sub("downsample.1."), F.conv2d(idn, sd["downsample.0.weight"]))
    return F.relu(out + idn)


def clip_visual(sd, x):
    """ModifiedResNet.forward with attnpool(pos_embedding=False) (DACLIP.py:329-349, 226-259)."""
    x = F.relu(_bn(sd.sub("bn1."), F.conv2d(x, sd["conv1.weight"], stride=2, padding=1)))
    x = F.relu(_bn(sd.sub("bn2."), F.conv2d(x, sd["conv2.weight"], padding=1)))
    x = F.relu(_bn(sd.sub("bn3."), F.conv2d(x, sd["conv3.weight"], padding=1)))
    x = F.avg_pool2d(x, 2)
    for li in range(1, 5):
        bi = 0
        while sd.has(f"layer{li}.{bi}.conv1.weight"):
            stride = 2 if (li > 1 and bi == 0) else 1
            x = _bottleneck(sd.sub(f"layer{li}.{bi}."), x, stride)
            bi += 1
    # attention pool: token 0 = mean token; only its output is used
    ap = sd.sub("attnpool.")
    n, c, h, w = x.shape
    tok = x.reshape(n, c, h * w).permute(2, 0, 1)                # (HW, N, C)
    tok = torch.cat([tok.mean(dim=0, keepdim=True), tok], dim=0)  # (HW+1, N, C)
    heads = c // 64
    q = F.linear(tok[:1], ap["q_proj.weight"], ap["q_proj.bias"])
    k = F.linear(tok, ap["k_proj.weight"], ap["k_proj.bias"])
    v = F.linear(tok, ap["v_proj.weight"], ap["v_proj.bias"])
    T = tok.shape[0]
    dh = c // heads
    q = q.reshape(1, n, heads, dh).permute(1, 2, 0, 3) * dh ** -0.5
    k = k.reshape(T, n, heads, dh).permute(1, 2, 0, 3)
    v = v.reshape(T, n, heads, dh).permute(1, 2, 0, 3)
    a = (q @ k.transpose(-2, -1)).softmax(dim=-1)
    o = (a @ v).permute(2, 0, 1, 3).reshape(1, n, c)
    return F.linear(o, ap["c_proj.weight"], ap["c_proj.bias"])[0]


def dose_encoder(sd, x3):
    """CLIPIQA.forward live outputs: (dose_emb (B,1024), ctx_emb (B,256)) (DACLIP.py:1203-1210)."""
    feat = clip_visual(sd.sub("clip_model.visual."), x3)
    img = F.linear(F.relu(F.linear(feat, sd["head1.0.weight"], sd["head1.0.bias"])),
                   sd["head1.2.weight"], sd["head1.2.bias"])
    ctx = F.linear(F.relu(F.linear(feat, sd["head2.0.weight"], sd["head2.0.bias"])),
                   sd["head2.2.weight"], sd["head2.2.bias"])
    ctx = F.normalize(ctx, dim=1)
    img = img / img.norm(dim=-1, keepdim=True)
    return img, ctx


# ------------------------------------------------------------------ DA U-Net
def da_unet_cond(sd, x_cond):
    """t-independent conditioning: (ctx (B,1,256), prompt_emb (B,time_dim)).  x_cond (B,1,H,W)."""
    dose, ctx = dose_encoder(sd.sub("dose_encoder."), x_cond.repeat(1, 3, 1, 1))
    tm = F.linear(F.silu(F.linear(dose, sd["text_mlp.0.weight"], sd["text_mlp.0.bias"])),
                  sd["text_mlp.2.weight"], sd["text_mlp.2.bias"])
    pe = torch.softmax(tm, dim=1) * sd["prompt"]
    pe = F.linear(pe, sd["prompt_mlp.weight"], sd["prompt_mlp.bias"])
    return ctx.unsqueeze(1), pe


def _stage_count(sd, name):
    n = 0
    while sd.has(f"{name}.{n}.0.block1.proj.weight"):
        n += 1
    return n


def da_unet(sd, x, time, cond=None, scan_fn=None, taps=None):
    """DADiff.Unet.forward (685-740).  x (B,2,H,W) = cat(x_t, x_input); time (B,) float."""
    dim = sd["init_conv.weight"].shape[0]
    if cond is None:
        cond = da_unet_cond(sd, x[:, 1:2])
    c, pe = cond
    x = F.conv2d(x, sd["init_conv.weight"], sd["init_conv.bias"], padding=3)
    r = x
    t = time_mlp(sd.sub("time_mlp."), time, dim) + pe
    if taps is not None:
        taps["t"] = t
        taps["init"] = x
    hs = []
    nd = _stage_count(sd, "downs")
    for i in range(nd):
        s = sd.sub(f"downs.{i}.")
        x = mamba_block(s.sub("1."), x, c, t, scan_fn)
        if taps is not None:
            taps[f"downs.{i}.mamba"] = x
        x = da_resnet_block(s.sub("0."), x)
        hs.append(x)
        w = s["2.weight"]
        x = F.conv2d(x, w, s["2.bias"], stride=2, padding=1) if w.shape[-1] == 4 else \
            F.conv2d(x, w, s["2.bias"], padding=1)
        if taps is not None:
            taps[f"downs.{i}.out"] = x
    x = da_resnet_block(sd.sub("mid_block."), x)
    x = mamba_block(sd.sub("mid_attn."), x, c, t, scan_fn)
    if taps is not None:
        taps["mid"] = x
    nu = _stage_count(sd, "ups")
    for i in range(nu):
        s = sd.sub(f"ups.{i}.")
        x = torch.cat((x, hs.pop()), dim=1)
        x = da_resnet_block(s.sub("0."), x)
        x = mamba_block(s.sub("1."), x, c, t, scan_fn)
        if s.has("2.1.weight"):
            x = F.interpolate(x, scale_factor=2, mode="nearest")
            x = F.conv2d(x, s["2.1.weight"], s["2.1.bias"], padding=1)
        else:
            x = F.conv2d(x, s["2.weight"], s["2.bias"], padding=1)
        if taps is not None:
            taps[f"ups.{i}.out"] = x
    x = torch.cat((x, r), dim=1)
    x = da_resnet_block(sd.sub("final_res_block."), x)
    return F.conv2d(x, sd["final_conv.weight"], sd["final_conv.bias"])


# ------------------------------------------------------------------ vanilla U-Net
def _chan_layernorm(x, g):
    var = torch.var(x, dim=1, unbiased=False, keepdim=True)
    mean = torch.mean(x, dim=1, keepdim=True)
    return (x - mean) * torch.rsqrt(var + 1e-5) * g


def v_resnet_block(sd, x, t, groups=8):
    ss = None
    if t is not None and sd.has("mlp.1.weight"):
        te = F.linear(F.silu(t), sd["mlp.1.weight"], sd["mlp.1.bias"])[:, :, None, None]
        ss = te.chunk(2, dim=1)
    h = block(sd.sub("block1."), x, groups, ss)
    h = block(sd.sub("block2."), h, groups)
    if sd.has("res_conv.weight"):
        return h + F.conv2d(x, sd["res_conv.weight"], sd["res_conv.bias"])
    return h + x


def v_linear_attention(sd, x, heads=4, dim_head=32):
    """Residual(PreNorm(LinearAttention)) (denoising_diffusion_pytorch.py:227-255)."""
    b, c, h, w = x.shape
    xn = _chan_layernorm(x, sd["norm.g"])
    f = sd.sub("fn.")
    qkv = F.conv2d(xn, f["to_qkv.weight"]).chunk(3, dim=1)
    q, k, v = [t.reshape(b, heads, dim_head, h * w) for t in qkv]
    q = q.softmax(dim=-2) * dim_head ** -0.5
    k = k.softmax(dim=-1)
    v = v / (h * w)
    ctx = torch.einsum("bhdn,bhen->bhde", k, v)
    out = torch.einsum("bhde,bhdn->bhen", ctx, q).reshape(b, heads * dim_head, h, w)
    out = F.conv2d(out, f["to_out.0.weight"], f["to_out.0.bias"])
    return _chan_layernorm(out, f["to_out.1.g"]) + x


def v_attention(sd, x, heads=4, dim_head=32):
    """Residual(PreNorm(Attention)) (denoising_diffusion_pytorch.py:257-279)."""
    b, c, h, w = x.shape
    xn = _chan_layernorm(x, sd["norm.g"])
    f = sd.sub("fn.")
    qkv = F.conv2d(xn, f["to_qkv.weight"]).chunk(3, dim=1)
    q, k, v = [t.reshape(b, heads, dim_head, h * w) for t in qkv]
    q = q * dim_head ** -0.5
    sim = torch.einsum("bhdi,bhdj->bhij", q, k)
    attn = sim.softmax(dim=-1)
    out = torch.einsum("bhij,bhdj->bhid", attn, v)
    out = out.permute(0, 1, 3, 2).reshape(b, heads * dim_head, h, w)
    return F.conv2d(out, f["to_out.weight"], f["to_out.bias"]) + x


def vanilla_unet(sd, x, time):
    """denoising_diffusion_pytorch.Unet.forward (371-410).  time (B,) long or float."""
    dim = sd["init_conv.weight"].shape[0]
    x = F.conv2d(x, sd["init_conv.weight"], sd["init_conv.bias"], padding=3)
    r = x
    t = time_mlp(sd.sub("time_mlp."), time, dim)
    hs = []
    nd = 0
    while sd.has(f"downs.{nd}.0.block1.proj.weight"):
        nd += 1
    for i in range(nd):
        s = sd.sub(f"downs.{i}.")
        x = v_resnet_block(s.sub("0."), x, t)
        hs.append(x)
        x = v_resnet_block(s.sub("1."), x, t)
        x = v_linear_attention(s.sub("2.fn."), x)
        hs.append(x)
        w = s["3.weight"]
        x = F.conv2d(x, w, s["3.bias"], stride=2, padding=1) if w.shape[-1] == 4 else \
            F.conv2d(x, w, s["3.bias"], padding=1)
    x = v_resnet_block(sd.sub("mid_block1."), x, t)
    x = v_attention(sd.sub("mid_attn.fn."), x)
    x = v_resnet_block(sd.sub("mid_block2."), x, t)
    for i in range(nd):
        s = sd.sub(f"ups.{i}.")
        x = torch.cat((x, hs.pop()), dim=1)
        x = v_resnet_block(s.sub("0."), x, t)
        x = torch.cat((x, hs.pop()), dim=1)
        x = v_resnet_block(s.sub("1."), x, t)
        x = v_linear_attention(s.sub("2.fn."), x)
        if s.has("3.1.weight"):
            x = F.interpolate(x, scale_factor=2, mode="nearest")
            x = F.conv2d(x, s["3.1.weight"], s["3.1.bias"], padding=1)
        else:
            x = F.conv2d(x, s["3.weight"], s["3.bias"], padding=1)
    x = torch.cat((x, r), dim=1)
    x = v_resnet_block(sd.sub("final_res_block."), x, t)
    return F.conv2d(x, sd["final_conv.weight"], sd["final_conv.bias"])
