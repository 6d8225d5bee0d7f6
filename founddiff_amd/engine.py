"""Host side of the MI355X denoiser: weight packing, workspace planning and the launch
sequence of one `Unet.forward` (src/DADiff.py:685-740 of the reference) over the C ABI in
include/founddiff_hip.h.  PyTorch is used for device memory, streams and graph capture only;
every FLOP of the forward runs in libfounddiff_hip.so.

Layout: activations NHWC; `mode` 'fp32' (parity: fp32 storage, exact-f32 MFMA) or 'bf16'
(bf16 storage + bf16 MFMA, fp32 accumulation/statistics/scan state).
"""
import ctypes as C
import os

import torch

from . import _lib as L

# 'fp8' (BASELINE configs[4]): bf16 activations and kernels, plus e4m3 weights with one scale per output channel
# for the convolutions the fp8 MFMA path takes (3x3 halo kernel, K axis in 128-channel slabs)
# 'fp32s': fp32 storage and kernels like 'fp32', but the dense contractions run split-bf16 (3 bf16 MFMAs per product,
# ~2^-16) instead of the exact-f32 MFMA: the engine behind the LAST step of a bf16 sampling loop (ResidualDiffusion
# final_fp32_steps).  The parity mode 'fp32' (the 1e-3 gate) stays exact.
# 'fp16' (round 6): the 'bf16' engine -- same kernels, same dataflow, same bytes -- on the library's second build, whose 16-bit
# type is IEEE binary16 (csrc/fd_common.h: FD_HALF_F16; lib/libfounddiff_hip_f16.so): 11 significand bits instead of 8 in every
# stored activation and weight.  Range: 65504 / 6e-8 (DESIGN.md section 5 for what that means for a checkpoint).
_T = {"fp32": (L.FD_F32, torch.float32), "bf16": (L.FD_BF16, torch.bfloat16), "fp8": (L.FD_BF16, torch.bfloat16),
      "fp32s": (L.FD_F32, torch.float32), "fp16": (L.FD_BF16, torch.float16)}
_HALF = (torch.bfloat16, torch.float16)
FP8_ACT_SCALE = 8.0     # activations are multiplied by this power of two before the e4m3 conversion (|x| <= 56 exact range)


def _dev(name, default):
    """Development switch of the host side (A/B tools): honoured only when FOUNDDIFF_DEV=1 is set as well -- a production process
    ignores it, like the release build of the library ignores the FD_* switches (include/founddiff_hip.h: fd_dev_options)."""
    return os.environ.get(name, default) if os.environ.get("FOUNDDIFF_DEV") == "1" else default


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _po(t, off_elems):
    """pointer to element offset inside tensor t"""
    return C.c_void_p(t.data_ptr() + off_elems * t.element_size())


class ConvW:
    """Packed convolution / linear weight: [Cout][KH*KW*Cin] in the compute dtype, fp32 bias."""

    def __init__(self, w_oihw, bias, dev, tdt, cin_pad=None, fp8=False, up2x=False, split=False):
        w = w_oihw.detach().float()
        if w.dim() == 2:
            w = w[:, :, None, None]
        o, i, kh, kw = w.shape
        if cin_pad is not None and cin_pad > i:
            w = torch.cat([w, w.new_zeros(o, cin_pad - i, kh, kw)], dim=1)
            i = cin_pad
        self.Cout, self.Cin, self.KH, self.KW = o, i, kh, kw
        wk = w.permute(0, 2, 3, 1).reshape(o, kh * kw * i).contiguous()
        self.w = wk.to(dev, tdt)
        self.b = bias.detach().float().contiguous().to(dev) if bias is not None else None
        self.w8 = self.ws = self.w_up = self.w_hi = self.w_lo = self.w_up_hi = self.w_up_lo = None
        if split and kh == 1 and kw == 1 and tdt == torch.float32 and i == 64:
            # fp32s, the 64-channel Mamba blocks: in_proj / qkv pre-split for the fused LN -> 1x1 -> depthwise kernels (fd_pwdw32.hip)
            hi = wk.to(torch.bfloat16)
            self.w_hi, self.w_lo = hi.contiguous().to(dev), (wk - hi.float()).to(torch.bfloat16).contiguous().to(dev)
        if split and kh == 3 and kw == 3 and tdt == torch.float32 and i % 64 == 0:
            # the fp32s engine's 3x3 convolutions on the halo-tiled kernel: w = hi + lo in bf16 (include/founddiff_hip.h: weight_split_hi / _lo)
            hi = wk.to(torch.bfloat16)
            self.w_hi, self.w_lo = hi.contiguous().to(dev), (wk - hi.float()).to(torch.bfloat16).contiguous().to(dev)
            if up2x and cin_pad is None:
                # ... and the up-sampling ones as four split 2x2 convolutions: the fp32 sub-pixel matrix, split the same way
                wu = pack_up2x(w)
                uh = wu.to(torch.bfloat16)
                self.w_up_hi, self.w_up_lo = uh.contiguous().to(dev), (wu - uh.float()).to(torch.bfloat16).contiguous().to(dev)
        if (up2x and kh == 3 and kw == 3 and tdt in _HALF and cin_pad is None
                and not (fp8 and _dev("FOUNDDIFF_FP8_UPCONV", "0") == "1")):       # (development: e4m3 9-tap up-sampling convs again)
            self.w_up = pack_up2x(w).to(dev, tdt)          # the up-sampling convs as four 2x2 convs on the source grid
        # (the fp8 mode runs its up-sampling convolutions on the bf16 four-2x2 form: as fast as 9 e4m3 taps at twice the rate
        #  -- 29.5 vs 29.5 slices/s alternated -- and nothing lost to the weight quantisation: drift 3.7e-2 -> 3.1e-2)
        if fp8 and self.w_up is None and kh == 3 and kw == 3 and i % 128 == 0:
            # per-output-channel scaled OCP e4m3 (largest finite value 448): w ~= w8 * ws[n]
            ws = (wk.abs().amax(dim=1).clamp(min=1e-12) / 448.0)
            self.w8 = (wk / ws[:, None]).to(torch.float8_e4m3fn).contiguous().to(dev)
            self.ws = ws.contiguous().to(dev)


def pack_up2x(w_oihw):
    """3x3 weights (O, I, 3, 3) of an up-sampling convolution (nn.Upsample(scale_factor=2, nearest) -> Conv2d, src/DADiff.py:121-127)
    -> the sub-pixel matrix [O][cls = 2 a + b][r][c][I] of include/founddiff_hip.h (fd_conv_params.weight_up2x): the 3x3 taps that
    read the same SOURCE pixel from output parity (a, b) summed in fp32 (a = 0: rows {0}, {1, 2}; a = 1: rows {0, 1}, {2};
    columns likewise).  Four 2x2 convolutions on the source grid = the same sums with 4 instead of 9 MACs per output."""
    w = w_oihw.detach().float()
    o, i = w.shape[:2]
    sets = {0: ([0], [1, 2]), 1: ([0, 1], [2])}
    out = w.new_zeros(o, 4, 2, 2, i)
    for a in (0, 1):
        for b in (0, 1):
            for r in (0, 1):
                for c in (0, 1):
                    out[:, 2 * a + b, r, c] = w[:, :, sets[a][r]][:, :, :, sets[b][c]].sum(dim=(2, 3))
    return out.reshape(o, 16 * i).contiguous()


def ws_standardize(w, eps=1e-5):
    """Weight standardisation folded at pack time (src/DADiff.py:145-152, fp32 eps)."""
    w = w.detach().float()
    mean = w.mean(dim=(1, 2, 3), keepdim=True)
    var = w.var(dim=(1, 2, 3), unbiased=False, keepdim=True)
    return (w - mean) * torch.rsqrt(var + eps)


def fold_bn(w, bn_w, bn_b, mean, var, eps=1e-5):
    s = bn_w.float() / torch.sqrt(var.float() + eps)
    return w.float() * s[:, None, None, None], bn_b.float() - mean.float() * s


class _Sub:
    def __init__(self, sd, prefix):
        self.sd, self.prefix = sd, prefix

    def __getitem__(self, k):
        return self.sd[self.prefix + k]

    def has(self, k):
        return (self.prefix + k) in self.sd

    def sub(self, p):
        return _Sub(self.sd, self.prefix + p)


class DAEngine:
    """DA-conditioned U-Net denoiser (reference `Unet`, src/DADiff.py:530-740) on HIP kernels."""
    _GEN = 0        # every engine gets a unique generation number: captured HIP graphs are keyed on it
    probe = None    # development hook: probe(tag, tensor) after each stage (tools/drift_table.py, stage_times.py)
    hip = L.BF16    # the build of the C ABI an engine calls: the default one, or L.F16 for mode 'fp16' (__init__)

    @property
    def half(self):
        """16-bit storage (bfloat16, or binary16 on the second build): the fused kernel set"""
        return self.tdt in _HALF

    def __init__(self, state_dict, prefix="", device="cuda", mode="bf16", low_latency=False):
        self.hip = L.F16 if mode == "fp16" else L.BF16    # which build of the C ABI this engine calls (founddiff_amd/_lib.py)
        self.hip.lib()  # fail loudly if the HIP library is missing
        DAEngine._GEN += 1
        self.gen = DAEngine._GEN
        if mode not in _T:
            raise ValueError(f"mode must be 'fp32', 'fp32s', 'bf16', 'fp16' or 'fp8', got {mode!r}")
        self.mode = mode
        self.dt, self.tdt = _T[mode]
        self.fp8 = mode == "fp8"
        self.f32_split = int(mode == "fp32s")
        # low_latency: the kernel set for ONE slice at a time (the reference's Trainer.test loop).  The default set is
        # chosen for throughput at a batch that fills the chip; both are functions of the image size only.
        self.low_latency = bool(low_latency)
        self.scan_dt = self.dt | (L.FD_OPT_LOW_LATENCY if low_latency else 0) | (L.FD_OPT_F32_SPLIT if mode == "fp32s" else 0)
        # z gate of SS2D recomputed inside out_proj instead of written by in_proj and read back (mamba_block); 0 = round-3 dataflow
        # (development: 64 = only in the 64-channel blocks)
        self.z_recompute = int(_dev("FOUNDDIFF_Z_RECOMPUTE", "1"))
        # v of the 64-channel TransposedAttention recomputed inside the kernel that applies Weff (mamba_block); 0 = stored v
        self.v_recompute = _dev("FOUNDDIFF_V_RECOMPUTE", "1") == "1"
        # GroupNorm apply of the down-path blocks fused with the 4x4 / stride-2 convolution behind them (_down); 0 = two passes
        self.down_fuse = _dev("FOUNDDIFF_DOWN_FUSE", "1") == "1"
        self.dev = torch.device(device)
        self.f32 = dict(device=self.dev, dtype=torch.float32)
        sd = _Sub(state_dict, prefix)
        self._pack(sd)
        self._plan_key = None
        self.buf = {}
        self.graphs, self.loop_graphs = {}, {}     # HIP graphs captured over this engine's buffers (ResidualDiffusion)

    # ------------------------------------------------------------------ packing
    def _f(self, t):
        return t.detach().float().contiguous().to(self.dev)

    def _convw(self, w, b=None, cin_pad=None, up2x=False):
        return ConvW(w, b, self.dev, self.tdt, cin_pad, fp8=getattr(self, "fp8", False), up2x=up2x, split=bool(getattr(self, "f32_split", 0)))

    def _pack_res(self, s):
        r = {"conv": self._convw(ws_standardize(s["block1.proj.weight"]), s["block1.proj.bias"]),
             "gamma": self._f(s["block1.norm.weight"]), "beta": self._f(s["block1.norm.bias"]),
             "res": None}
        if s.has("res_conv.weight"):
            r["res"] = self._convw(s["res_conv.weight"], s["res_conv.bias"])
        return r

    def _pack_init7(self, w):
        """init_conv weight (Cout, C<=3, 7, 7) -> bf16 [Cout][7 kh][8 kw][4 c] (include/founddiff_hip.h:
        fd_init_conv7): one filter row = one K32 MFMA step."""
        co, c = w.shape[0], w.shape[1]
        if getattr(self, "f32_split", 0) and c == 2 and tuple(w.shape[2:]) == (7, 7):
            # fp32s engine (fd_init_conv7_f32s): bf16(w) in all four slots (planes + their rounding residuals) and the
            # weights' own residual bf16(w - bf16(w)) in slots 0, 1 -- the three terms of a split-bf16 product
            wf = w.detach().float().permute(0, 2, 3, 1)
            hi = wf.to(torch.bfloat16)
            ph, pl = torch.zeros(co, 7, 8, 4), torch.zeros(co, 7, 8, 4)
            ph[:, :, :7, 0:2] = hi.float()
            ph[:, :, :7, 2:4] = hi.float()
            pl[:, :, :7, 0:2] = wf - hi.float()
            return (ph.reshape(co, 224).contiguous().to(self.dev, torch.bfloat16),
                    pl.reshape(co, 224).contiguous().to(self.dev, torch.bfloat16))
        if self.tdt not in _HALF or c > 3 or tuple(w.shape[2:]) != (7, 7):
            return None
        p = torch.zeros(co, 7, 8, 4, dtype=torch.float32)
        p[:, :, :7, :c] = w.detach().float().permute(0, 2, 3, 1)
        if c <= 2:      # free slots 2, 3 take the rounding residuals of planes 0, 1 (fd_init_conv7)
            p[:, :, :7, 2:2 + c] = p[:, :, :7, :c]
        return p.reshape(co, 224).contiguous().to(self.dev, self.tdt)

    @staticmethod
    def _dw_masked(w9c):
        """[9][C] fp32 taps (tap = 3*dy + dx) -> [9][C/2] int32 words of fp16 CHANNEL pairs (low half = channel 2j, high =
        2j + 1): the operand layout of the packed-fp16 depthwise of fd_pw_dw3x3 / fd_pw_dw3x3_gram / fd_dwconv_gram
        (include/founddiff_hip.h)."""
        h = w9c.to(torch.float16).view(torch.int16).to(torch.int32) & 0xFFFF
        return (h[:, 0::2] | (h[:, 1::2] << 16)).contiguous()

    @staticmethod
    def _qk_prescale(qkv_w, dw_w, C_):
        """Per-channel power-of-two scales for the q and k thirds of TransposedAttention's qkv / qkv_dwconv weights
        (src/DADiff.py:266-276).  q and k are consumed only through F.normalize(dim=-1) (273-274), so any positive
        per-channel factor cancels; a power of two commutes with every rounding (bf16 weights, fp16 on-chip tiles, fp32
        sums), so the result is unchanged BIT FOR BIT as long as nothing leaves its range -- and that is the point:
        the fused kernels keep q / k as fp16 on chip (fd_pwdw.hip), where a channel of ~1e-6 magnitude would be
        subnormal or zero.  Rows are brought to unit norm (to the nearest power of two): the 1x1 output of a
        LayerNorm'd pixel then has O(1) rms, the depthwise output likewise.  qkv_w (3C, C) / dw_w (3C, 9) fp32."""
        qkv_w, dw_w = qkv_w.clone(), dw_w.clone()
        for w in (qkv_w, dw_w):
            n = w[:2 * C_].double().norm(dim=1)
            e = torch.frexp(n)[1].clamp(-100, 100)
            sc = torch.where((n > 0) & torch.isfinite(n), torch.ldexp(torch.ones_like(n), -e), torch.ones_like(n))
            w[:2 * C_] *= sc[:, None].to(w.dtype)
        return qkv_w, dw_w

    def _pack_mamba(self, s):
        m = s.sub("mamba.")
        C_ = s["norm1.weight"].shape[0]
        xw = m["x_proj_weight"]            # (4, R+2N, 2C)
        dtw = m["dt_projs_weight"]         # (4, 2C, R)
        N = m["A_logs"].shape[1]
        R = dtw.shape[2]
        D = dtw.shape[1]
        a = s.sub("attn_blk.")
        qkv_w = a["qkv.weight"].detach().float().reshape(3 * C_, C_)
        qdw = a["qkv_dwconv.weight"].detach().float().reshape(3 * C_, 9)
        if self.tdt in _HALF:      # (the fp32 parity modes keep the checkpoint's values untouched)
            qkv_w, qdw = self._qk_prescale(qkv_w, qdw, C_)
        d = dict(
            C=C_, N=N, R=R, D=D, CD=R + 2 * N, heads=a["temperature"].shape[0],
            n1w=self._f(s["norm1.weight"]), n1b=self._f(s["norm1.bias"]),
            in_proj=self._convw(m["in_proj.weight"]),
            dw_w=self._f(m["conv2d.weight"].reshape(D, 9).t()), dw_b=self._f(m["conv2d.bias"]),
            x_proj=xw.detach().float().contiguous().to(self.dev, self.tdt),
            dtw=self._f(dtw), dtb=self._f(m["dt_projs_bias"]),
            A=self._f(-torch.exp(m["A_logs"].detach().float())), Ds=self._f(m["Ds"]),
            onw=self._f(m["out_norm.weight"]), onb=self._f(m["out_norm.bias"]),
            out_proj=self._convw(m["out_proj.weight"]),
            qkv=self._convw(qkv_w),
            qdw_w=self._f(qdw.t()),
            temp=self._f(a["temperature"].reshape(-1)),
            wproj=self._f(a["project_out.weight"].reshape(C_, C_)),
        )
        # tap weights of the two depthwise convs in the operand layout of the fused 1x1 -> 3x3 kernel
        d["dw_wm"], d["qdw_wm"] = self._dw_masked(d["dw_w"]), self._dw_masked(d["qdw_w"])
        d["qdw_w_v"] = d["qdw_w"][:, 2 * C_:].contiguous()          # the v third alone (fd_dwconv_gram serves q and k)
        d["qdw_wm_v"] = self._dw_masked(d["qdw_w_v"])               # ... in the fused kernels' layout (fd_pw_dw3x3_proj)
        assert d["heads"] * 32 == C_, "TransposedAttention heads must be C/32 (src/DADiff.py:468)"
        d["adaln_w"] = s["adaLN_modulation.1.weight"].detach().float()
        d["adaln_b"] = s["adaLN_modulation.1.bias"].detach().float()
        d["local_w"] = m["attn.0.weight"].detach().float()
        return d

    def _pack(self, sd):
        self.dim = sd["init_conv.weight"].shape[0]
        self.time_dim = sd["time_mlp.1.weight"].shape[0]
        self.init_conv = self._convw(sd["init_conv.weight"], sd["init_conv.bias"], cin_pad=8)
        self.in_planes = sd["init_conv.weight"].shape[1]          # 2, or 3 with input_condition
        self.init_w7 = self._pack_init7(sd["init_conv.weight"])
        self.tm = dict(w1=self._f(sd["time_mlp.1.weight"]), b1=self._f(sd["time_mlp.1.bias"]),
                       w2=self._f(sd["time_mlp.3.weight"]), b2=self._f(sd["time_mlp.3.bias"]))
        self.prompt = dict(
            w0=self._f(sd["text_mlp.0.weight"]), b0=self._f(sd["text_mlp.0.bias"]),
            w2=self._f(sd["text_mlp.2.weight"]), b2=self._f(sd["text_mlp.2.bias"]),
            p=self._f(sd["prompt"].reshape(-1)),
            wp=self._f(sd["prompt_mlp.weight"]), bp=self._f(sd["prompt_mlp.bias"]))
        self.downs, self.ups = [], []
        i = 0
        while sd.has(f"downs.{i}.0.block1.proj.weight"):
            s = sd.sub(f"downs.{i}.")
            w = s["2.weight"]
            self.downs.append(dict(res=self._pack_res(s.sub("0.")), mamba=self._pack_mamba(s.sub("1.")),
                                   samp=self._convw(w, s["2.bias"]), stride=2 if w.shape[-1] == 4 else 1))
            i += 1
        self.mid_res = self._pack_res(sd.sub("mid_block."))
        self.mid_mamba = self._pack_mamba(sd.sub("mid_attn."))
        i = 0
        while sd.has(f"ups.{i}.0.block1.proj.weight"):
            s = sd.sub(f"ups.{i}.")
            up = s.has("2.1.weight")
            w, b = (s["2.1.weight"], s["2.1.bias"]) if up else (s["2.weight"], s["2.bias"])
            self.ups.append(dict(res=self._pack_res(s.sub("0.")), mamba=self._pack_mamba(s.sub("1.")),
                                 samp=self._convw(w, b, up2x=up), up=up))
            i += 1
        self.final_res = self._pack_res(sd.sub("final_res_block."))
        fw = sd["final_conv.weight"]
        if fw.shape[0] != 1:
            raise NotImplementedError("final_conv with out_dim != 1 (learned_variance) is not built")
        self.final_w, self.final_b = self._f(fw.reshape(-1)), self._f(sd["final_conv.bias"])
        self.final_b_host = float(sd["final_conv.bias"].detach().float().reshape(-1)[0])
        # all adaLN / local projections concatenated: ONE matvec per step / per slice
        mambas = [d["mamba"] for d in self.downs] + [self.mid_mamba] + [u["mamba"] for u in self.ups]
        off_m = off_l = 0
        for m in mambas:
            m["mod_off"], m["loc_off"] = off_m, off_l
            off_m += 6 * m["C"]
            off_l += m["D"]
        self.mod_total, self.loc_total = off_m, off_l
        self.adaln_w = torch.cat([m.pop("adaln_w") for m in mambas]).contiguous().to(self.dev)
        self.adaln_b = torch.cat([m.pop("adaln_b") for m in mambas]).contiguous().to(self.dev)
        self.local_w = torch.cat([m.pop("local_w") for m in mambas]).contiguous().to(self.dev)
        self.mambas = mambas
        self._pack_clip(sd.sub("dose_encoder."))

    def _pack_clip(self, sd):
        v = sd.sub("clip_model.visual.")

        def cbn(cw, bn, cin_pad=None, sum_in=False):
            w = v[cw + ".weight"].detach().float()
            if sum_in:   # 3 identical input channels (x.repeat(1,3,1,1), src/DADiff.py:692) -> 1
                w = w.sum(dim=1, keepdim=True)
            w, b = fold_bn(w, v[bn + ".weight"], v[bn + ".bias"], v[bn + ".running_mean"], v[bn + ".running_var"])
            return self._convw(w, b, cin_pad)
        stem = [cbn("conv1", "bn1", cin_pad=8, sum_in=True), cbn("conv2", "bn2"), cbn("conv3", "bn3")]
        layers = []
        for li in range(1, 5):
            bi = 0
            while v.has(f"layer{li}.{bi}.conv1.weight"):
                p = f"layer{li}.{bi}."
                blk = dict(c1=cbn(p + "conv1", p + "bn1"), c2=cbn(p + "conv2", p + "bn2"),
                           c3=cbn(p + "conv3", p + "bn3"), stride=2 if (li > 1 and bi == 0) else 1, ds=None)
                if v.has(p + "downsample.0.weight"):
                    blk["ds"] = cbn(p + "downsample.0", p + "downsample.1")
                layers.append(blk)
                bi += 1
        ap = v.sub("attnpool.")
        Cf = ap["q_proj.weight"].shape[0]
        qkv_w = torch.cat([ap["q_proj.weight"], ap["k_proj.weight"], ap["v_proj.weight"]]).detach().float()
        qkv_b = torch.cat([ap["q_proj.bias"], ap["k_proj.bias"], ap["v_proj.bias"]]).detach().float()
        self.clip = dict(
            stem=stem, layers=layers, Cf=Cf, heads=Cf // 64,
            qkv=self._convw(qkv_w, qkv_b),
            cw=self._f(ap["c_proj.weight"]), cb=self._f(ap["c_proj.bias"]),
            h1=[self._f(sd[k]) for k in ("head1.0.weight", "head1.0.bias", "head1.2.weight", "head1.2.bias")],
            h2=[self._f(sd[k]) for k in ("head2.0.weight", "head2.0.bias", "head2.2.weight", "head2.2.bias")])

    # ------------------------------------------------------------------ buffers
    def _b(self, name, shape, dtype=None):
        """named, shape-keyed workspace tensor (same role + shape => same memory)"""
        dtype = dtype or self.tdt
        key = (name, tuple(shape), dtype)
        t = self.buf.get(key)
        if t is None:
            t = torch.empty(shape, device=self.dev, dtype=dtype)
            self.buf[key] = t
        return t

    def workspace_bytes(self):
        """bytes of the named workspace tensors this engine holds (grows with the largest batch it has run)"""
        return sum(t.numel() * t.element_size() for t in self.buf.values())

    def _pr(self, tag, t):
        if self.probe is not None:
            self.probe(tag, t)

    @property
    def stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.dev).cuda_stream)

    # ------------------------------------------------------------------ op wrappers
    def conv(self, cw, in0, B, H, W, out, *, c0=None, ld0=None, off0=0, in1=None, c1=0, ld1=0, off1=0,
             stride=1, pad=None, upsample=False, ndir=1, w_batch_stride=0, w_dir_stride=0, weight=None,
             bias="auto", Cout=None, KH=None, KW=None, ldo=None, offo=0, out_dir_stride=0, out_f32=False,
             epi=L.EPI_NONE, split=0, res=None, ld_res=0, off_res=0, gate=None, gate_ld=0, h=None,
             gn=None, gamma=None, beta=None, groups=8, stats=None, OH=None, OW=None,
             prologue=L.PRO_NONE, ln_gamma=None, ln_beta=None, ln_eps=1e-5, ln_shift=None, ln_scale=None,
             ln_ld=0, ln_z=None, ln_ldz=0, ln_offz=0, probe=False, fin=None, zre=None):
        """One fd_conv2d launch.  `probe=True` only asks the library whether this conv can take the
        fused LayerNorm prologue (bf16 streaming row-GEMM path) and launches nothing; `probe="kid"`
        returns fd_conv_kernel_id (tests pin which kernel a shape exercises)."""
        def ptr(v):
            if v is None:
                return None
            return v.value if isinstance(v, C.c_void_p) else v.data_ptr()
        p = L.ConvParams()
        KH = KH or cw.KH
        KW = KW or cw.KW
        Cout = Cout or cw.Cout
        c0 = c0 if c0 is not None else (cw.Cin - c1)
        p.dtype, p.out_f32 = self.dt, int(out_f32)
        p.in0, p.in1 = in0.data_ptr(), (in1.data_ptr() if in1 is not None else None)
        p.c0, p.ld0, p.off0 = c0, (ld0 if ld0 is not None else c0), off0
        p.c1, p.ld1, p.off1 = c1, (ld1 or c1), off1
        p.B, p.H, p.W, p.upsample = B, H, W, int(upsample)
        if pad is None:
            pad = (KH - 1) // 2
        p.KH, p.KW, p.stride, p.pad_h, p.pad_w = KH, KW, stride, pad, pad
        Hs, Ws = (2 * H, 2 * W) if upsample else (H, W)
        p.OH = OH if OH is not None else (Hs + 2 * pad - KH) // stride + 1
        p.OW = OW if OW is not None else (Ws + 2 * pad - KW) // stride + 1
        p.ndir = ndir
        wt = weight if weight is not None else cw.w
        p.weight, p.w_batch_stride, p.w_dir_stride = ptr(wt), w_batch_stride, w_dir_stride
        bt = (cw.b if cw is not None else None) if isinstance(bias, str) else bias
        p.bias = bt.data_ptr() if bt is not None else None
        p.Cout = Cout
        p.out, p.ldo, p.offo, p.out_dir_stride = out.data_ptr(), (ldo or Cout), offo, out_dir_stride
        p.epilogue, p.epi_split = epi, split
        p.res = res.data_ptr() if res is not None else None
        p.ld_res, p.off_res = (ld_res or Cout), off_res
        p.gate = gate.value if isinstance(gate, C.c_void_p) else (gate.data_ptr() if gate is not None else None)
        p.gate_ld = gate_ld
        p.h = h.data_ptr() if h is not None else None
        p.gn_mean_rstd = gn.data_ptr() if gn is not None else None
        p.gn_gamma = gamma.data_ptr() if gamma is not None else None
        p.gn_beta = beta.data_ptr() if beta is not None else None
        p.gn_groups = groups
        p.stats_partial = stats.data_ptr() if stats is not None else None
        p.prologue, p.ln_eps = prologue, ln_eps
        p.ln_gamma, p.ln_beta = ptr(ln_gamma), ptr(ln_beta)
        p.ln_shift, p.ln_scale, p.ln_ld = ptr(ln_shift), ptr(ln_scale), ln_ld
        p.ln_z, p.ln_ldz, p.ln_offz = ptr(ln_z), ln_ldz, ln_offz
        p.f32_split = getattr(self, "f32_split", 0)
        if zre is not None:      # PRO_LN_GATE_ZRE: z = SiLU(w . LNmod(res)) recomputed in the operand load
            p.zre_w, p.zre_gamma, p.zre_beta = ptr(zre["w"]), ptr(zre.get("gamma")), ptr(zre.get("beta"))
            p.zre_shift, p.zre_scale, p.zre_ld, p.zre_eps = ptr(zre["shift"]), ptr(zre["scale"]), zre["ld"], zre["eps"]
        if fin is not None:      # EPI_GNSILU_ADD_FINAL: final_conv (+ DDIM update) in the epilogue
            p.fin_w, p.fin_b, p.fin_out = fin["w"].data_ptr(), float(fin["b"]), fin["out"].data_ptr()
            p.fin_mode, p.fin_last = int(fin.get("mode", 0)), int(fin.get("last", 0))
            p.fin_img = fin["img"].data_ptr() if fin.get("img") is not None else None
            p.fin_xin = fin["xin"].data_ptr() if fin.get("xin") is not None else None
            p.fin_alpha = float(fin.get("alpha", 0.0))
        if weight is None and cw is not None and getattr(cw, "w8", None) is not None:
            p.weight_f8, p.w_scale, p.act_scale = cw.w8.data_ptr(), cw.ws.data_ptr(), FP8_ACT_SCALE
        # (the one-slice kernel set asks for one workgroup per (tile, parity class) -- `upsample` = 2: a low-resolution tile grid
        #  alone is 64..256 workgroups for a lone 512x512 slice, 150.8 against 146.3 ms per 50-step slice with the 9-tap form,
        #  profiles/r05/latency_b1_sweep.txt; any split gives the same bits)
        if weight is None and cw is not None and getattr(cw, "w_hi", None) is not None and KH == 3:
            p.weight_split_hi, p.weight_split_lo = cw.w_hi.data_ptr(), cw.w_lo.data_ptr()
        if upsample and weight is None and cw is not None and getattr(cw, "w_up_hi", None) is not None:
            p.weight_up2x_split_hi, p.weight_up2x_split_lo = cw.w_up_hi.data_ptr(), cw.w_up_lo.data_ptr()
            if getattr(self, "low_latency", False):
                p.upsample = 2
        if upsample and weight is None and cw is not None and getattr(cw, "w_up", None) is not None:
            p.weight_up2x = cw.w_up.data_ptr()
            if getattr(self, "low_latency", False):
                p.upsample = 2
        if probe == "kid":          # which kernel would run (include/founddiff_hip.h: fd_conv_kernel_id)
            return int(self.hip.lib().fd_conv_kernel_id(C.byref(p)))
        if probe:
            return bool(self.hip.lib().fd_conv_prologue_ok(C.byref(p)))
        self.hip.call("fd_conv2d", C.byref(p), self.stream)
        return p.OH, p.OW

    def conv_cols(self, cw, x, B, H, W, out, ldo, nchunks, *, split=None, **kw):
        """A bias-free 1x1 convolution as `nchunks` launches over equal column (output channel) ranges of its weight matrix, each
        writing its channels of `out` (row stride `ldo`): the fp32s row-GEMM keeps TWO bf16 images of its weights in LDS
        (fd_gemm_rows32.hip), and the widest layers (in_proj 128 -> 512, qkv 128 -> 384) only fit in halves / thirds.  Every
        launch reads the (narrow) input again and runs the fused LayerNorm prologue again; the wide output is written once.
        `split`: SiLU on output channels >= split (EPI_SILU_SPLIT over the whole matrix).  Returns False -- nothing launched --
        when a chunk does not fit either."""
        assert cw.KH == 1 and cw.KW == 1 and cw.b is None and cw.Cout % nchunks == 0
        n = cw.Cout // nchunks
        es = cw.w.element_size()
        calls = []
        for j in range(nchunks):
            k2 = dict(kw, weight=C.c_void_p(cw.w.data_ptr() + j * n * cw.Cin * es), bias=None, Cout=n, ldo=ldo, offo=j * n)
            if split is not None:
                k2.update(epi=L.EPI_SILU_SPLIT, split=min(max(split - j * n, 0), n))
            if not self.conv(cw, x, B, H, W, out, probe=True, **k2):
                return False
            calls.append(k2)
        for k2 in calls:
            self.conv(cw, x, B, H, W, out, **k2)
        return True

    def linear(self, x, w, b, out, act=L.ACT_NONE, pre_silu=False):
        M, K = x.shape
        N = w.shape[0]
        self.hip.call("fd_linear", _p(x), _p(w), _p(b), _p(out), M, N, K, act, int(pre_silu), self.stream)
        return out

    # ------------------------------------------------------------------ blocks
    def res_block(self, r, in0, c0, in1, c1, B, H, W, tag, defer_apply=False):
        """DADiff ResnetBlock: conv3x3(WS)+GN+SiLU, + res_conv(x) or x (src/DADiff.py:397-430).  defer_apply (identity
        residual only): stop after the GroupNorm statistics and return (h, mean_rstd, out buffer) -- the caller fuses the
        apply pass into its consumer (_down: fd_gn_apply_down4x4)."""
        cw = r["conv"]
        Co = cw.Cout
        hw = H * W
        mt = self.hip.lib().fd_conv_mtiles(H, W)
        hraw = self._b("res_h", (B, H, W, Co))
        part = self._b("gn_part", (B, mt, Co, 2), torch.float32)
        mr = self._b("gn_mr", (B, 8, 2), torch.float32)
        self.conv(cw, in0, B, H, W, hraw, c0=c0, in1=in1, c1=c1, stats=part)
        self._pr(tag + ".conv3", hraw)
        self.hip.call("fd_gn_finalize", _p(part), B, mt, Co, 8, hw, 1e-5, _p(mr), self.stream)
        out = self._b(tag, (B, H, W, Co))
        if defer_apply:
            assert r["res"] is None and in1 is None
            return hraw, mr, out
        if r["res"] is not None:
            self.conv(r["res"], in0, B, H, W, out, c0=c0, in1=in1, c1=c1, epi=L.EPI_GNSILU_ADD, h=hraw, gn=mr,
                      gamma=r["gamma"], beta=r["beta"], groups=8)
        else:
            assert in1 is None
            self.hip.call("fd_gn_silu_apply", self.dt, _p(hraw), _p(mr), _p(r["gamma"]), _p(r["beta"]), _p(in0), _p(out),
                   B, hw, Co, 8, self.stream)
        self._pr(tag, out)
        return out

    def mamba_block(self, m, x, B, H, W, tag):
        """adaLN-gated SS2D + channel attention (src/DADiff.py:477-488)."""
        Cc, D, N, R, CD = m["C"], m["D"], m["N"], m["R"], m["CD"]
        hw = H * W
        s = self.stream
        mod = self.mod_all
        ml = self.mod_total
        mo = m["mod_off"]
        f4 = 4  # bytes per float
        mp = lambda k: C.c_void_p(mod.data_ptr() + (mo + k * Cc) * f4)
        # --- SS2D branch.  Where the library's streaming row-GEMM can run the projection (bf16,
        # high-resolution levels) the LayerNorm+modulate / out_norm*z+local producers are fused into
        # its operand load; otherwise they run as separate row kernels.
        xz = self._b("xz", (B, H, W, 2 * D))
        ln1 = dict(prologue=L.PRO_LN_MOD, ln_gamma=m["n1w"], ln_beta=m["n1b"], ln_eps=1e-5, ln_shift=mp(0),
                   ln_scale=mp(1), ln_ld=ml)
        xc = self._b("xc", (B, H, W, D))
        fused = bool(self.hip.lib().fd_pw_dw3x3_ok(getattr(self, 'scan_dt', self.dt), Cc, D, D, H, W))      # (low latency: C = 128 unfused)
        # out_proj's operands (decided here: whether in_proj has to write z at all depends on them)
        y = self._b("scan_y", (B, H, W, D))
        x1 = self._b(tag + ".x1", (B, H, W, Cc))
        loc = C.c_void_p(self.local_all.data_ptr() + m["loc_off"] * f4)
        ep1 = dict(epi=L.EPI_GATE_RES, res=x, gate=mp(2), gate_ld=ml)
        lng = dict(prologue=L.PRO_LN_GATE, ln_gamma=m["onw"], ln_beta=m["onb"], ln_eps=1e-5, ln_shift=loc,
                   ln_ld=self.loc_total, ln_z=xz, ln_ldz=2 * D, ln_offz=D)
        # z recomputed inside out_proj from the block input it reads anyway as its residual (fd_gemm_rows.hip:
        # gemm_rows_zre_kernel): the fused in_proj then writes the depthwise half only and z never exists in HBM
        lngz = dict(prologue=L.PRO_LN_GATE_ZRE, ln_gamma=m["onw"], ln_beta=m["onb"], ln_eps=1e-5, ln_shift=loc,
                    ln_ld=self.loc_total,
                    zre=dict(w=C.c_void_p(m["in_proj"].w.data_ptr() + D * Cc * m["in_proj"].w.element_size()),
                             gamma=m["n1w"], beta=m["n1b"], shift=mp(0), scale=mp(1), ld=ml, eps=1e-5))
        xonly = dict(Cout=D, ldo=2 * D)                  # in_proj restricted to its x half (rows 0 .. D-1), z columns of xz untouched
        # fp32s engine (fp32 storage, split-bf16 contractions): its own fused LN -> in_proj -> conv2d kernel (fd_pwdw32.hip)
        f32s = bool(getattr(self, "f32_split", 0))
        fused32 = f32s and m["in_proj"].w_hi is not None and bool(self.hip.lib().fd_pw_dw3x3_f32_ok(self.dt, Cc, D, H, W))
        zre = (getattr(self, "z_recompute", 0) in (1, Cc) and self.conv(m["out_proj"], y, B, H, W, x1, probe=True, **ep1, **lngz)
               and (fused32 or (bool(self.hip.lib().fd_pw_dw3x3_ok(getattr(self, 'scan_dt', self.dt), Cc, D, 0, H, W)) if fused else
                                self.conv(m["in_proj"], x, B, H, W, xz, epi=L.EPI_SILU_SPLIT, split=D, probe=True, **ln1, **xonly))))
        fused32 = fused32 and zre            # (the fp32 kernel writes the depthwise half only: z has to be recomputed)
        if fused32:
            self.hip.call("fd_pw_dw3x3_f32", _p(x), Cc, 0, Cc, _p(m["n1w"]), _p(m["n1b"]), 1e-5, mp(0), mp(1), ml, _p(m["in_proj"].w_hi), _p(m["in_proj"].w_lo), D,
                   _p(m["dw_w"]), _p(m["dw_b"]), 1, _p(xc), D, 0, B, H, W, s)
        elif fused:
            # LN+modulate -> in_proj -> conv2d+SiLU (x half) / SiLU (z half) in one pass: the x half of
            # in_proj's output never exists in HBM (xz[..., :D] stays unwritten, z lands in xz[..., D:])
            self.hip.call("fd_pw_dw3x3", self.dt, _p(x), Cc, 0, Cc, _p(m["n1w"]), _p(m["n1b"]), 1e-5, mp(0), mp(1), ml,
                   _p(m["in_proj"].w), D, _p(m["dw_wm"]), _p(m["dw_b"]), 1, _p(xc), D, 0,
                   0 if zre else D, None if zre else _p(xz), 2 * D, D, B, H, W, s)
        elif zre:
            self.conv(m["in_proj"], x, B, H, W, xz, epi=L.EPI_SILU_SPLIT, split=D, **ln1, **xonly)
        elif self.conv(m["in_proj"], x, B, H, W, xz, epi=L.EPI_SILU_SPLIT, split=D, probe=True, **ln1):
            self.conv(m["in_proj"], x, B, H, W, xz, epi=L.EPI_SILU_SPLIT, split=D, **ln1)
        elif getattr(self, "f32_split", 0) and self.conv_cols(m["in_proj"], x, B, H, W, xz, 2 * D, 2, split=D, **ln1):
            pass                                          # fp32s, C = 128: the x half and the z half as two row-GEMM launches
        else:
            xm = self._b("xm", (B, H, W, Cc))
            self.hip.call("fd_ln_modulate", self.dt, _p(x), _p(m["n1w"]), _p(m["n1b"]), 1e-5, mp(0), mp(1), ml, _p(xm),
                   B, hw, Cc, s)
            self.conv(m["in_proj"], xm, B, H, W, xz, epi=L.EPI_SILU_SPLIT, split=D)
        if not fused and not fused32:
            self.hip.call("fd_dwconv3x3", self.dt, _p(xz), 2 * D, 0, _p(m["dw_w"]), _p(m["dw_b"]), 1, _p(xc), D, 0,
                   B, H, W, D, s)
        self._pr(tag + ".xc", xc)
        if not zre:
            self._pr(tag + ".z", xz[..., D:])
        # odd H / W: the four sub-grids are those of the image zero-padded to even sizes (src/emamba2.py:191-199);
        # the x_proj gather zero-fills the positions outside the image, the scan treats them as padding
        H2, W2 = (H + 1) // 2, (W + 1) // 2
        Lq = H2 * W2
        xdbl = self._b("xdbl", (4, B, Lq, CD), torch.float32)
        nws = self.hip.lib().fd_scan_ws_floats(B, H, W, D, N)
        ws = self._b("scan_ws", (nws,), torch.float32)
        if self.hip.lib().fd_selective_scan_plan(getattr(self, 'scan_dt', self.dt), D, N, R, H, W):
            # x_proj inside the scan's first phase (one workgroup per chunk at d_inner <= 256): no separate pass over xc
            self.hip.call("fd_selective_scan_xproj", getattr(self, "scan_dt", self.dt), _p(xc), _p(m["x_proj"]), _p(xdbl), _p(m["dtw"]), _p(m["dtb"]),
                   _p(m["A"]), _p(m["Ds"]), _p(y), _p(ws), B, H, W, D, N, R, s)
            self._pr(tag + ".xdbl", xdbl)
        else:
            self.conv(None, xc, B, H, W, xdbl, c0=D, weight=m["x_proj"], bias=None, Cout=CD, KH=1, KW=1, stride=2,
                      pad=0, ndir=4, w_dir_stride=CD * D, out_dir_stride=B * Lq * CD, out_f32=True,
                      OH=H2, OW=W2)
            self._pr(tag + ".xdbl", xdbl)
            self.hip.call("fd_selective_scan", getattr(self, "scan_dt", self.dt), _p(xc), _p(xdbl), _p(m["dtw"]), _p(m["dtb"]), _p(m["A"]),
                   _p(m["Ds"]), _p(y), _p(ws), B, H, W, D, N, R, s)
        self._pr(tag + ".y", y)
        if zre:
            self.conv(m["out_proj"], y, B, H, W, x1, **ep1, **lngz)
        elif self.conv(m["out_proj"], y, B, H, W, x1, probe=True, **ep1, **lng):
            self.conv(m["out_proj"], y, B, H, W, x1, **ep1, **lng)
        else:
            yz = self._b("yz", (B, H, W, D))
            self.hip.call("fd_ln_gate", self.dt, _p(y), _p(m["onw"]), _p(m["onb"]), 1e-5, _p(xz), 2 * D, D, loc,
                   self.loc_total, _p(yz), B, hw, D, s)
            self.conv(m["out_proj"], yz, B, H, W, x1, **ep1)
        self._pr(tag + ".x1", x1)
        # --- channel attention branch
        ln2 = dict(prologue=L.PRO_LN_MOD, ln_eps=1e-6, ln_shift=mp(3), ln_scale=mp(4), ln_ld=ml)
        if (f32s and m["qkv"].w_hi is not None and self.hip.lib().fd_pw_dw3x3_gram_f32_ok(self.dt, Cc, H, W)
                and self.hip.lib().fd_pw_dw3x3_proj_f32_ok(self.dt, Cc, H, W)):
            # fp32s: q, k -> depthwise -> Gram + norms in one pass over x1, then v -> depthwise -> Weff -> gated residual in
            # another: q, k, v and the attention output never reach HBM (fd_pwdw32.hip)
            nblk = self.hip.lib().fd_pw_dw3x3_gram_f32_nblk(H, W)
            part = self._b("gram", (B, m["heads"], nblk, 1024 + 64), torch.float32)
            self.hip.call("fd_pw_dw3x3_gram_f32", _p(x1), Cc, 0, Cc, None, None, 1e-6, mp(3), mp(4), ml, _p(m["qkv"].w_hi), _p(m["qkv"].w_lo), _p(m["qdw_w"]),
                   3 * Cc, _p(part), B, H, W, s)
            weff = self._b("weff", (B, Cc, Cc))
            self.hip.call("fd_chan_attn_weff", self.dt, _p(part), nblk, _p(m["temp"]), _p(m["wproj"]), _p(weff), B, Cc, s)
            self._pr(tag + ".weff", weff)
            x2 = self._b(tag + ".x2", (B, H, W, Cc))
            wvh, wvl = (C.c_void_p(t.data_ptr() + 2 * Cc * Cc * t.element_size()) for t in (m["qkv"].w_hi, m["qkv"].w_lo))
            self.hip.call("fd_pw_dw3x3_proj_f32", _p(x1), Cc, 0, Cc, None, None, 1e-6, mp(3), mp(4), ml, wvh, wvl, _p(m["qdw_w_v"]), Cc,
                   _p(weff), mp(5), ml, _p(x2), Cc, 0, B, H, W, s)
            self._pr(tag, x2)
            return x2
        if self.hip.lib().fd_pw_dw3x3_gram_ok(self.dt, Cc, H, W):
            # qkv -> qkv_dwconv -> L2 norms + q k^T in one pass: q and k never reach HBM, only v and one Gram
            # partial per workgroup do (fd_pwdw.hip: pwdw_gram_kernel)
            gdt = getattr(self, 'scan_dt', self.dt)          # carries FD_OPT_LOW_LATENCY: tiles per workgroup of the Gram kernel
            nblk = self.hip.lib().fd_pw_dw3x3_gram_nblk_opts(gdt, H, W)
            # v recomputed where it is consumed (fd_pw_dw3x3_proj: LN -> W_v -> depthwise -> Weff -> gated residual in one
            # pass over x1): the Gram kernel then runs q and k only and v never reaches HBM
            vre = getattr(self, "v_recompute", False) and bool(self.hip.lib().fd_pw_dw3x3_proj_ok(gdt, Cc, H, W))
            vbuf = None if vre else self._b("attn_v", (B, H, W, Cc))
            part = self._b("gram", (B, m["heads"], nblk, 1024 + 64), torch.float32)
            self.hip.call("fd_pw_dw3x3_gram", gdt, _p(x1), Cc, 0, Cc, None, None, 1e-6, mp(3), mp(4), ml,
                   _p(m["qkv"].w), _p(m["qdw_wm"]), _p(vbuf) if vbuf is not None else None, Cc, 0, _p(part), B, H, W, s)
            if not vre:
                self._pr(tag + ".qkv2", vbuf)
            weff = self._b("weff", (B, Cc, Cc))
            self.hip.call("fd_chan_attn_weff", self.dt, _p(part), nblk, _p(m["temp"]), _p(m["wproj"]), _p(weff), B, Cc, s)
            self._pr(tag + ".weff", weff)
            x2 = self._b(tag + ".x2", (B, H, W, Cc))
            if vre:
                wv = C.c_void_p(m["qkv"].w.data_ptr() + 2 * Cc * Cc * m["qkv"].w.element_size())
                self.hip.call("fd_pw_dw3x3_proj", gdt, _p(x1), Cc, 0, Cc, None, None, 1e-6, mp(3), mp(4), ml, wv, _p(m["qdw_wm_v"]),
                       _p(weff), mp(5), ml, _p(x2), Cc, 0, B, H, W, s)
                self._pr(tag, x2)
                return x2
            self.conv(None, vbuf, B, H, W, x2, c0=Cc, ld0=Cc, off0=0, weight=weff, w_batch_stride=Cc * Cc,
                      bias=None, Cout=Cc, KH=1, KW=1, epi=L.EPI_GATE_RES, res=x1, gate=mp(5), gate_ld=ml)
            self._pr(tag, x2)
            return x2
        qkv2 = self._b("qkv2", (B, H, W, 3 * Cc))
        if self.hip.lib().fd_pw_dw3x3_ok(self.dt, Cc, 3 * Cc, 0, H, W):
            self.hip.call("fd_pw_dw3x3", self.dt, _p(x1), Cc, 0, Cc, None, None, 1e-6, mp(3), mp(4), ml,
                   _p(m["qkv"].w), 3 * Cc, _p(m["qdw_wm"]), None, 0, _p(qkv2), 3 * Cc, 0,
                   0, None, 0, 0, B, H, W, s)
        else:
            qkv = self._b("qkv", (B, H, W, 3 * Cc))
            if self.conv(m["qkv"], x1, B, H, W, qkv, probe=True, **ln2):
                self.conv(m["qkv"], x1, B, H, W, qkv, **ln2)
            elif getattr(self, "f32_split", 0) and self.conv_cols(m["qkv"], x1, B, H, W, qkv, 3 * Cc, 3, **ln2):
                pass                                      # fp32s, C = 128: q, k, v as three row-GEMM launches
            else:
                xm2 = self._b("xm", (B, H, W, Cc))
                self.hip.call("fd_ln_modulate", self.dt, _p(x1), None, None, 1e-6, mp(3), mp(4), ml, _p(xm2), B, hw, Cc, s)
                self.conv(m["qkv"], xm2, B, H, W, qkv)
            if self.hip.lib().fd_dwconv_gram_ok(self.dt, Cc, H, W):
                # qkv_dwconv of q and k straight into the Gram (fd_pwdw.hip: dwconv_gram_kernel): only v is written
                nblk = self.hip.lib().fd_dwconv_gram_nblk(H, W)
                vbuf = self._b("attn_v", (B, H, W, Cc))
                part = self._b("gram", (B, m["heads"], nblk, 1024 + 64), torch.float32)
                self.hip.call("fd_dwconv3x3", self.dt, _p(qkv), 3 * Cc, 2 * Cc, _p(m["qdw_w_v"]), None, 0, _p(vbuf), Cc, 0,
                       B, H, W, Cc, s)
                self.hip.call("fd_dwconv_gram", self.dt, _p(qkv), 3 * Cc, Cc, _p(m["qdw_wm"]), _p(part), B, H, W, s)
                self._pr(tag + ".qkv2", vbuf)
                weff = self._b("weff", (B, Cc, Cc))
                self.hip.call("fd_chan_attn_weff", self.dt, _p(part), nblk, _p(m["temp"]), _p(m["wproj"]), _p(weff), B, Cc, s)
                self._pr(tag + ".weff", weff)
                x2 = self._b(tag + ".x2", (B, H, W, Cc))
                self.conv(None, vbuf, B, H, W, x2, c0=Cc, ld0=Cc, off0=0, weight=weff, w_batch_stride=Cc * Cc,
                          bias=None, Cout=Cc, KH=1, KW=1, epi=L.EPI_GATE_RES, res=x1, gate=mp(5), gate_ld=ml)
                self._pr(tag, x2)
                return x2
            self.hip.call("fd_dwconv3x3", self.dt, _p(qkv), 3 * Cc, 0, _p(m["qdw_w"]), None, 0, _p(qkv2), 3 * Cc, 0,
                   B, H, W, 3 * Cc, s)
        self._pr(tag + ".qkv2", qkv2)
        nblk = self.hip.lib().fd_chan_attn_nblk(hw)
        part = self._b("gram", (B, m["heads"], nblk, 1024 + 64), torch.float32)
        self.hip.call("fd_chan_attn_gram", self.dt, _p(qkv2), B, hw, Cc, _p(part), s)
        weff = self._b("weff", (B, Cc, Cc))
        self.hip.call("fd_chan_attn_weff", self.dt, _p(part), nblk, _p(m["temp"]), _p(m["wproj"]), _p(weff), B, Cc, s)
        self._pr(tag + ".weff", weff)
        x2 = self._b(tag + ".x2", (B, H, W, Cc))
        self.conv(None, qkv2, B, H, W, x2, c0=Cc, ld0=3 * Cc, off0=2 * Cc, weight=weff, w_batch_stride=Cc * Cc,
                  bias=None, Cout=Cc, KH=1, KW=1, epi=L.EPI_GATE_RES, res=x1, gate=mp(5), gate_ld=ml)
        self._pr(tag, x2)
        return x2

    # ------------------------------------------------------------------ conditioning (once per slice)
    def encode_condition(self, x_cond):
        """DA-CLIP encoder + prompt path + per-block `local` vectors; t-independent,
        so it runs once per slice instead of once per step.  x_cond (B,1,H,W) fp32 in [-1,1]."""
        B, _, H, W = x_cond.shape
        s = self.stream
        cl = self.clip
        x8 = self._b("clip_in", (B, H, W, 8))
        self.hip.call("fd_pack_planes", self.dt, _p(x_cond), None, _p(x8), B, H * W, 8, s)
        st = cl["stem"]
        h1, w1 = (H + 2 - 3) // 2 + 1, (W + 2 - 3) // 2 + 1
        a = self._b("clip_s1", (B, h1, w1, st[0].Cout))
        self.conv(st[0], x8, B, H, W, a, stride=2, epi=L.EPI_RELU)
        b_ = self._b("clip_s2", (B, h1, w1, st[1].Cout))
        self.conv(st[1], a, B, h1, w1, b_, epi=L.EPI_RELU)
        c_ = self._b("clip_s3", (B, h1, w1, st[2].Cout))
        self.conv(st[2], b_, B, h1, w1, c_, epi=L.EPI_RELU)
        h, w = h1 // 2, w1 // 2
        x = self._b("clip_p0", (B, h, w, st[2].Cout))
        self.hip.call("fd_avgpool", self.dt, _p(c_), _p(x), B, h1, w1, st[2].Cout, 2, s)
        Cx = st[2].Cout
        for i, blk in enumerate(cl["layers"]):
            stride = blk["stride"]
            o1 = self._b(f"clip_{i}_1", (B, h, w, blk["c1"].Cout))
            self.conv(blk["c1"], x, B, h, w, o1, epi=L.EPI_RELU)
            o2 = self._b(f"clip_{i}_2", (B, h, w, blk["c2"].Cout))
            self.conv(blk["c2"], o1, B, h, w, o2, epi=L.EPI_RELU)
            ho, wo = h // stride, w // stride
            if stride > 1:
                o2p = self._b(f"clip_{i}_2p", (B, ho, wo, blk["c2"].Cout))
                self.hip.call("fd_avgpool", self.dt, _p(o2), _p(o2p), B, h, w, blk["c2"].Cout, stride, s)
                o2 = o2p
            idn = x
            if blk["ds"] is not None:
                xi = x
                if stride > 1:
                    xi = self._b(f"clip_{i}_xp", (B, ho, wo, Cx))
                    self.hip.call("fd_avgpool", self.dt, _p(x), _p(xi), B, h, w, Cx, stride, s)
                idn = self._b(f"clip_{i}_id", (B, ho, wo, blk["ds"].Cout))
                self.conv(blk["ds"], xi, B, ho, wo, idn)
            o3 = self._b(f"clip_{i}_3", (B, ho, wo, blk["c3"].Cout))
            self.conv(blk["c3"], o2, B, ho, wo, o3, epi=L.EPI_RES_RELU, res=idn)
            x, Cx, h, w = o3, blk["c3"].Cout, ho, wo
        Cf, T = cl["Cf"], h * w + 1
        tok = self._b("clip_tok", (B, T, Cf))
        self.hip.call("fd_attnpool_tokens", self.dt, _p(x), _p(tok), B, h * w, Cf, s)
        qkv = self._b("clip_qkv", (B, T, 3 * Cf), torch.float32)
        self.conv(cl["qkv"], tok, B, 1, T, qkv, out_f32=True)
        pooled = self._b("clip_pool", (B, Cf), torch.float32)
        self.hip.call("fd_attnpool_core", _p(qkv), T * 3 * Cf, _p(qkv), 3 * Cf, Cf, 2 * Cf, _p(pooled), B, T, Cf,
               cl["heads"], s)
        feat = self.linear(pooled, cl["cw"], cl["cb"], self._b("clip_feat", (B, cl["cw"].shape[0]), torch.float32))
        t1 = self.linear(feat, cl["h1"][0], cl["h1"][1], self._b("h1a", (B, cl["h1"][0].shape[0]), torch.float32), L.ACT_RELU)
        t2 = self.linear(t1, cl["h1"][2], cl["h1"][3], self._b("h1b", (B, cl["h1"][2].shape[0]), torch.float32))
        dose = self._b("dose_emb", t2.shape, torch.float32)
        self.hip.call("fd_l2norm_rows", _p(t2), _p(dose), B, t2.shape[1], 0.0, s)
        t1 = self.linear(feat, cl["h2"][0], cl["h2"][1], self._b("h2a", (B, cl["h2"][0].shape[0]), torch.float32), L.ACT_RELU)
        t2 = self.linear(t1, cl["h2"][2], cl["h2"][3], self._b("h2b", (B, cl["h2"][2].shape[0]), torch.float32))
        ctx = self._b("ctx_emb", t2.shape, torch.float32)
        self.hip.call("fd_l2norm_rows", _p(t2), _p(ctx), B, t2.shape[1], 1e-12, s)
        # prompt path (src/DADiff.py:706-707)
        pr = self.prompt
        td = self.time_dim
        a1 = self.linear(dose, pr["w0"], pr["b0"], self._b("pm_a", (B, td), torch.float32), L.ACT_SILU)
        a2 = self.linear(a1, pr["w2"], pr["b2"], self._b("pm_b", (B, td), torch.float32))
        a3 = self._b("pm_c", (B, td), torch.float32)
        self.hip.call("fd_softmax_mul", _p(a2), _p(pr["p"]), _p(a3), B, td, s)
        self.prompt_emb = self.linear(a3, pr["wp"], pr["bp"], self._b("prompt_emb", (B, td), torch.float32))
        # SS2D `local` vectors of every block (src/emamba2.py:715)
        self.local_all = self.linear(ctx, self.local_w, None, self._b("local_all", (B, self.loc_total), torch.float32),
                                     L.ACT_SILU)
        self.dose_emb, self.ctx_emb = dose, ctx
        return dose, ctx

    def share_condition(self, other):
        """Take the t-independent conditioning (prompt embedding, per-block `local` vectors: fp32 in every
        mode) from another engine of the same weights instead of running the DA-CLIP tower again."""
        B = other.prompt_emb.shape[0]
        self.prompt_emb = self._b("prompt_emb", (B, self.time_dim), torch.float32)
        self.local_all = self._b("local_all", (B, self.loc_total), torch.float32)
        self.prompt_emb.copy_(other.prompt_emb)
        self.local_all.copy_(other.local_all)
        self.dose_emb, self.ctx_emb = other.dose_emb, other.ctx_emb

    # ------------------------------------------------------------------ one denoiser forward
    def time_cond(self, time):
        """time (B,) fp32 device -> adaLN vectors of all blocks (src/DADiff.py:703,709,484)."""
        B = time.shape[0]
        s = self.stream
        emb = self._b("t_emb", (B, self.dim), torch.float32)
        self.hip.call("fd_sinusoidal", _p(time), _p(emb), B, self.dim, s)
        tm = self.tm
        h = self.linear(emb, tm["w1"], tm["b1"], self._b("t_h", (B, self.time_dim), torch.float32), L.ACT_GELU)
        t = self.linear(h, tm["w2"], tm["b2"], self._b("t_t", (B, self.time_dim), torch.float32))
        tt = self._b("t_sum", (B, self.time_dim), torch.float32)
        self.hip.call("fd_add_f32", _p(t), _p(self.prompt_emb), _p(tt), B * self.time_dim, s)
        self.t_vec = tt
        self.mod_all = self.linear(tt, self.adaln_w, self.adaln_b,
                                   self._b("mod_all", (B, self.mod_total), torch.float32), pre_silu=True)

    def time_table_prepare(self, times, B):
        """Device vector of the loop's step times, every time repeated for the B slices (rows s * B + b).  A host -> device
        copy: call it BEFORE a graph capture; time_cond_table() then only launches kernels."""
        S = len(times)
        tv = self._b(f"tab_time_{S}x{B}", (S * B,), torch.float32)
        tv.copy_(torch.tensor([float(t) for t in times], dtype=torch.float32).repeat_interleave(B))
        self._tab_S, self._tab_B = S, B
        return tv

    def time_cond_table(self):
        """The adaLN vectors of ALL steps of a sampling loop in one pass: the step times are known before the loop starts
        (src/DADiff.py:1276-1300), so the six launches of time_cond per step -- sinusoidal, two linears, + prompt, the adaLN
        linear; ~75 us of latency-bound work in front of every forward -- become five launches per LOOP on S * B rows.
        Every row is computed by itself (fd_linear: one wave per output feature, rows in turn): bit for bit the vectors
        time_cond produces.  forward(..., step=s) takes its vectors from the table."""
        if getattr(self, "_tab_S", None) is None:
            raise RuntimeError("time_cond_table(): call time_table_prepare(times, B) first (it owns the host -> device copy)")
        S, B = self._tab_S, self._tab_B
        if self.prompt_emb.shape[0] != B:
            raise RuntimeError(f"time_cond_table(): the table was prepared for batch {B}, the conditioning holds {self.prompt_emb.shape[0]} slices")
        M = S * B
        s = self.stream
        tv = self._b(f"tab_time_{S}x{B}", (M,), torch.float32)
        emb = self._b("tab_emb", (M, self.dim), torch.float32)
        self.hip.call("fd_sinusoidal", _p(tv), _p(emb), M, self.dim, s)
        tm = self.tm
        h = self.linear(emb, tm["w1"], tm["b1"], self._b("tab_h", (M, self.time_dim), torch.float32), L.ACT_GELU)
        t = self.linear(h, tm["w2"], tm["b2"], self._b("tab_t", (M, self.time_dim), torch.float32))
        tt = self._b("tab_sum", (M, self.time_dim), torch.float32)
        torch.add(t.view(S, B, self.time_dim), self.prompt_emb[None], out=tt.view(S, B, self.time_dim))
        self.mod_tab = self.linear(tt, self.adaln_w, self.adaln_b, self._b("mod_tab", (M, self.mod_total), torch.float32),
                                   pre_silu=True)
        self._mod_tab_shape = (S, B)            # forward(step=) checks its batch and step against this

    def forward(self, x_t, x_in, time, out=None, x_cond2=None, sched=None, step=None):
        """x_t, x_in: (B,1,H,W) fp32 device tensors in [-1,1]; time (B,) fp32.  Returns the raw
        model output (B,1,H,W) fp32.  encode_condition(x_in) must have been called.  x_cond2: the
        third input plane of an input_condition model (src/DADiff.py:1157-1158).  `sched` = (alpha, last): also
        apply the DDIM update of src/DADiff.py:1203-1206, 1317-1318, 1344 to x_t IN PLACE (x_t <- x_t - alpha *
        clamp(out), or clamp(x_in - clamp(out)) when `last`) -- in the bf16 mode inside the last kernel of the forward."""
        if (x_cond2 is not None) != (self.in_planes == 3):
            raise ValueError(f"this Unet's init_conv takes {self.in_planes} input planes "
                             f"(input_condition={'True' if self.in_planes == 3 else 'False'})")
        B, _, H, W = x_t.shape
        nd = len(self.downs)
        div = 2 ** sum(1 for d in self.downs if d["stride"] == 2)
        if H % div or W % div:       # what the reference's own down / up-sampling needs (skip shapes must match)
            raise ValueError(f"H,W must be multiples of {div} (got {H}x{W})")
        if step is None:
            self.time_cond(time)
        else:                                   # vectors of loop step `step` from time_cond_table()
            tab = getattr(self, "_mod_tab_shape", None)
            if tab is None:
                raise RuntimeError("forward(step=...): no adaLN table -- call time_table_prepare() and time_cond_table() first")
            if tab[1] != B or not 0 <= int(step) < tab[0]:
                raise RuntimeError(f"forward(step={step}) on a batch of {B}: the adaLN table holds {tab[0]} steps x {tab[1]} slices")
            self.mod_all = self.mod_tab[step * B:(step + 1) * B]
        r = self._head(x_t, x_in, x_cond2)
        x, h, w = r, H, W
        self._skips = []
        for i in range(nd):
            x, h, w = self._down(i, x, B, h, w)
        x = self._mid(x, B, h, w)
        for i in range(len(self.ups)):
            x, h, w = self._up(i, x, B, h, w)
        return self._tail(x, r, B, H, W, out, sched, x_t, x_in)

    def forward_hybrid(self, inner, x_t, x_in, time, out=None, outer_levels=1, sched=None, down_levels=None):
        """One forward with THIS engine on the outermost resolution levels (init_conv, the first `down_levels` down
        stages, the last `outer_levels` up stages, final block) and `inner` -- another engine of the same weights, normally
        one precision class down -- on the levels in between; the activation crosses the boundary through a dtype cast
        (`fd_cast`).  The last step of a sampling loop runs this way (ResidualDiffusion: the split-bf16 fp32 engine at
        the outer levels, where most of the bf16 drift of the returned image originates, bf16 below).  `down_levels`
        (default = outer_levels) < outer_levels keeps part of the encoder on `inner`; its skips are cast on their way
        into this engine's up stages."""
        # (no `step` argument on purpose: the hybrid forward always computes its adaLN vectors from `time` on both engines --
        #  it runs once per loop, a table would save nothing)
        B, _, H, W = x_t.shape
        nd, nu = len(self.downs), len(self.ups)
        ku = outer_levels
        kd = ku if down_levels is None else max(0, min(int(down_levels), ku))
        assert 0 < ku < nd and nu == nd
        self.time_cond(time)
        inner.time_cond(time)
        r = self._head(x_t, x_in, None)
        x, h, w = r, H, W
        self._skips, inner._skips = [], []
        for i in range(kd):
            x, h, w = self._down(i, x, B, h, w)
        x = inner._cast_from(x, self, "hyb_in")
        for i in range(kd, nd):
            x, h, w = inner._down(i, x, B, h, w)
        x = inner._mid(x, B, h, w)
        for i in range(nu - ku):
            x, h, w = inner._up(i, x, B, h, w)
        x = self._cast_from(x, inner, "hyb_out")
        for lvl, (sk, hs, ws_) in enumerate(inner._skips):       # skips of levels kd .. ku - 1, shallowest first
            self._skips.append((self._cast_from(sk, inner, f"hyb_skip{kd + lvl}"), hs, ws_))
        inner._skips = []
        for i in range(nu - ku, nu):
            x, h, w = self._up(i, x, B, h, w)
        return self._tail(x, r, B, H, W, out, sched, x_t, x_in)

    def _cast_from(self, x, src, name):
        """x (a tensor of engine `src`) in this engine's storage type."""
        if src.tdt == self.tdt:
            return x
        o = self._b(name, tuple(x.shape))
        # (the build whose 16-bit type the 16-bit side is stored in does the cast: an 'fp16' engine next to the fp32s tail engine)
        (src if src.half else self).hip.call("fd_cast", src.dt, _p(x), self.dt, _p(o), x.numel(), self.stream)
        return o

    # ---- the stages of a forward (src/DADiff.py:703-740)
    def _head(self, x_t, x_in, x_cond2):
        B, _, H, W = x_t.shape
        s = self.stream
        r = self._b("r", (B, H, W, self.dim))
        if isinstance(self.init_w7, tuple) and x_cond2 is None and self.dim in (32, 64) and H % 16 == 0 and W % 16 == 0:
            self.hip.call("fd_init_conv7_f32s", _p(x_t), _p(x_in), _p(self.init_w7[0]), _p(self.init_w7[1]), _p(self.init_conv.b), _p(r),
                   B, H, W, self.dim, s)
        elif self.init_w7 is not None and not isinstance(self.init_w7, tuple) and self.hip.lib().fd_init_conv7_ok(self.dt, self.dim, H, W):
            self.hip.call("fd_init_conv7", self.dt, _p(x_t), _p(x_in), _p(x_cond2), _p(self.init_w7), _p(self.init_conv.b),
                   _p(r), B, H, W, self.dim, s)
        else:
            xin8 = self._b("unet_in", (B, H, W, 8))
            self.hip.call("fd_pack_planes3", self.dt, _p(x_t), _p(x_in), _p(x_cond2), _p(xin8), B, H * W, 8, s)
            self.conv(self.init_conv, xin8, B, H, W, r)
        self._pr("init", r)
        return r

    def _down(self, i, x, B, h, w):
        d = self.downs[i]
        x = self.mamba_block(d["mamba"], x, B, h, w, f"d{i}m")
        cw = d["samp"]
        Cx = x.shape[-1]
        if (d["stride"] == 2 and d["res"]["res"] is None and getattr(self, "down_fuse", False) and cw.KH == 4 and cw.KW == 4
                and getattr(cw, "w8", None) is None
                and self.hip.lib().fd_gn_apply_down4x4_ok(getattr(self, "scan_dt", self.dt), Cx, cw.Cout, h, w)):
            # GroupNorm apply + SiLU + residual of the block AND the 4x4 / stride-2 convolution behind it in one pass: the
            # block output (the skip) is written once and read back by nothing (fd_downfuse.hip)
            hraw, mr, sk = self.res_block(d["res"], x, Cx, None, 0, B, h, w, f"d{i}r", defer_apply=True)
            o = self._b(f"d{i}s", (B, h // 2, w // 2, cw.Cout))
            self.hip.call("fd_gn_apply_down4x4", self.dt, _p(hraw), _p(x), _p(mr), _p(d["res"]["gamma"]), _p(d["res"]["beta"]), 8,
                   _p(sk), _p(cw.w), _p(cw.b), _p(o), B, h, w, Cx, cw.Cout, self.stream)
            self._pr(f"d{i}r", sk)
            self._skips.append((sk, h, w))
            self._pr(f"d{i}s", o)
            return o, h // 2, w // 2
        x = self.res_block(d["res"], x, Cx, None, 0, B, h, w, f"d{i}r")
        self._skips.append((x, h, w))
        if d["stride"] == 2:
            o = self._b(f"d{i}s", (B, h // 2, w // 2, cw.Cout))
            self.conv(cw, x, B, h, w, o, stride=2, pad=1)
            h, w = h // 2, w // 2
        else:
            o = self._b(f"d{i}s", (B, h, w, cw.Cout))
            self.conv(cw, x, B, h, w, o)
        self._pr(f"d{i}s", o)
        return o, h, w

    def _mid(self, x, B, h, w):
        x = self.res_block(self.mid_res, x, x.shape[-1], None, 0, B, h, w, "midr")
        return self.mamba_block(self.mid_mamba, x, B, h, w, "midm")

    def _up(self, i, x, B, h, w):
        u = self.ups[i]
        sk, hs, ws_ = self._skips.pop()
        assert (hs, ws_) == (h, w)
        x = self.res_block(u["res"], x, x.shape[-1], sk, sk.shape[-1], B, h, w, f"u{i}r")
        x = self.mamba_block(u["mamba"], x, B, h, w, f"u{i}m")
        cw = u["samp"]
        if u["up"]:
            o = self._b(f"u{i}s", (B, 2 * h, 2 * w, cw.Cout))
            self.conv(cw, x, B, h, w, o, upsample=True)
            h, w = 2 * h, 2 * w
        else:
            o = self._b(f"u{i}s", (B, h, w, cw.Cout))
            self.conv(cw, x, B, h, w, o)
        self._pr(f"u{i}s", o)
        return o, h, w

    def _tail(self, x, r, B, H, W, out, sched=None, x_t=None, x_in=None):
        """final_res_block -> final_conv (src/DADiff.py:733-740) [-> the sampler's DDIM update of x_t, `sched`]."""
        if out is None:
            out = self._b("model_out", (B, 1, H, W), torch.float32)
        fr = self.final_res
        cw, hw = fr["conv"], H * W
        c0, c1 = x.shape[-1], r.shape[-1]
        fin = dict(w=self.final_w, b=self.final_b_host, out=out)
        if sched is not None:
            fin.update(mode=1, alpha=sched[0], last=int(bool(sched[1])), img=x_t, xin=x_in)
        kw = dict(c0=c0, in1=r, c1=c1)
        if (fr["res"] is not None and (self.tdt in _HALF or getattr(self, "f32_split", 0)) and not self.probe
                and _dev("FOUNDDIFF_NO_FINAL_FOLD", "") == ""):
            mt = self.hip.lib().fd_conv_mtiles(H, W)
            hraw = self._b("res_h", (B, H, W, cw.Cout))
            part = self._b("gn_part", (B, mt, cw.Cout, 2), torch.float32)
            mr = self._b("gn_mr", (B, 8, 2), torch.float32)
            ek = dict(epi=L.EPI_GNSILU_ADD_FINAL, h=hraw, gn=mr, gamma=fr["gamma"], beta=fr["beta"], groups=8, fin=fin)
            if self.conv(fr["res"], x, B, H, W, out, probe=True, **kw, **ek):
                # res_conv + GroupNorm/SiLU of the 3x3 output + final_conv (+ DDIM update) in ONE epilogue: the block's
                # 64-channel output, its read by final_conv and the separate update kernel never happen
                self.conv(cw, x, B, H, W, hraw, stats=part, **kw)
                self.hip.call("fd_gn_finalize", _p(part), B, mt, cw.Cout, 8, hw, 1e-5, _p(mr), self.stream)
                self.conv(fr["res"], x, B, H, W, out, **kw, **ek)
                return out
        x = self.res_block(fr, x, c0, r, c1, B, H, W, "finr")
        self.hip.call("fd_final_conv1", self.dt, _p(x), _p(self.final_w), _p(self.final_b), _p(out), B * hw,
               x.shape[-1], self.stream)
        self._pr("out", out)
        if sched is not None:
            self.hip.call("fd_res_ddim_step", _p(out), _p(x_t), _p(x_in), None, float(sched[0]), 0.0, int(bool(sched[1])), _p(x_t),
                   x_t.numel(), self.stream)
        return out
