/* founddiff_hip.h -- C ABI of libfounddiff_hip.so (MI355X / gfx950 only).
 *
 * The reference (hao1635/FoundDiff) is pure PyTorch with exactly ONE native boundary on the
 * sampling path:  selective_scan_cuda_core.fwd(u, delta, A, B, C, D, delta_bias,
 * delta_softplus, nrows)  at /root/reference/src/emamba2.py:154 (third-party CUDA, not
 * vendored).  Everything else bottoms out in ATen.  This library replaces that op AND the
 * ATen calls of the per-timestep denoiser with hand-written HIP kernels; each entry point
 * below cites the reference lines it replaces.
 *
 * Conventions
 *   - plain C: raw DEVICE pointers, ints, a hipStream_t passed as void*; no torch types.
 *   - activations are NHWC (channels-last) `dtype` elements: FD_F32 (parity mode, fp32
 *     storage + exact-f32 MFMA) or FD_BF16 (bf16 storage, bf16 MFMA, fp32 accumulate).
 *     Statistics, scan state, gates, biases and all "small" vectors are always fp32.
 *   - the caller owns every buffer incl. workspaces; the library allocates nothing, keeps
 *     no state, never synchronises: work is enqueued on `stream` (graph-capturable).
 *   - return 0 on success, <0 on error (fd_last_error() gives the text, thread-local).
 */
#ifndef FOUNDDIFF_HIP_H
#define FOUNDDIFF_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FD_F32 0
#define FD_BF16 1

#define FD_OK 0
#define FD_ERR_ARG (-1)
#define FD_ERR_LAUNCH (-2)

int fd_version(void);
/* Development switches (round 6).  The release build (what founddiff_amd.build.build() ships: -DFD_RELEASE) reads NO environment
 * variable: every tuning switch of the dispatch code is compiled to its default.  A development build (FOUNDDIFF_DEV_BUILD=1)
 * reads the FD_* variables of founddiff_amd/csrc/fd_common.h (FD_DEV_SWITCHES) once, on the first call that needs one.  Returns a
 * static string: "release build: ..." or the switches that are set.                                                    */
const char *fd_dev_options(void);
/* The 16-bit type behind dtype code FD_BF16 in THIS build of the library: 0 = bfloat16 (libfounddiff_hip.so, the default), 1 = IEEE
 * binary16 (libfounddiff_hip_f16.so: the same sources compiled with -DFD_HALF_F16; founddiff_amd's precision='fp16').  Every entry
 * point below that says "bf16" then reads and writes binary16 instead; fp32 arguments are unchanged.  The fp32-storage split mode
 * (FD_OPT_F32_SPLIT) and the fp8 weights mode are meant for the default build only.                                       */
int fd_half_format(void);
const char *fd_last_error(void);

/* ---- implicit-GEMM convolution / GEMM on MFMA -------------------------------------------
 * Replaces F.conv2d / nn.Linear / torch.cat / nn.Upsample(nearest) on the denoiser path:
 *   3x3 WS-conv of Block          src/DADiff.py:145-154, 221   (weights standardised at pack time)
 *   4x4-s2 Downsample, Upsample   src/DADiff.py:128-136
 *   7x7 init_conv, 1x1 res_conv   src/DADiff.py:558, 407-408, 430
 *   skip concat                   src/DADiff.py:727, 733       (two-source A operand)
 *   in_proj / out_proj / x_proj   src/emamba2.py:717, 748, 335 (x_proj = 4 stride-2 1x1 convs)
 *   qkv / project_out             src/DADiff.py:266, 283
 *   RN50 convs of DA-CLIP         src/DACLIP.py:198-211, 329-349 (BN folded at pack time)
 * out[b,oh,ow,n] = epi( bias[n] + sum_{kh,kw,c} in[b, (oh*s-ph+kh)(>>1 if upsample), ..., c] * w[n,kh,kw,c] )
 */
#define FD_EPI_NONE 0       /* acc + bias                                                      */
#define FD_EPI_SILU_SPLIT 1 /* SiLU on output channels >= epi_split (in_proj's z half)        */
#define FD_EPI_RELU 2
#define FD_EPI_GATE_RES 3   /* res[m,n] + gate[b,n] * (acc + bias)   adaLN-gated residual     */
#define FD_EPI_RES_RELU 4   /* relu(acc + bias + res[m,n])           RN50 bottleneck tail     */
#define FD_EPI_GNSILU_ADD 5 /* acc + bias + silu(GN(h[m,n]))         res_conv + Block output  */
/* GNSILU_ADD, then the denoiser's last 1x1 (final_conv, Cout -> 1) and the sampler's update in the same pass: the
 * block's Cout-channel output is never written.  o = fin_b + sum_n fin_w[n] * value[m,n] -> fin_out[b][m] (fp32);
 * fin_mode 1: + the DDIM update of src/DADiff.py:1203-1206, 1317-1318, 1344 on the fp32 image planes:
 *   pr = clamp(o, +-1);  fin_img[b][m] <- fin_last ? clamp(fin_xin - pr, +-1) : fin_img - fin_alpha * pr.
 * Streaming row-GEMM only (bf16, fd_conv_prologue_ok).                                              */
#define FD_EPI_GNSILU_ADD_FINAL 6

typedef struct fd_conv_params {
    int32_t dtype;              /* FD_F32 | FD_BF16: type of in0/in1/weight/res/h            */
    int32_t out_f32;            /* 1: `out` is fp32 regardless of dtype                       */
    const void *in0;            /* first source  [B,H,W,ld0], channels [off0, off0+c0)        */
    const void *in1;            /* optional second source (channel concat after in0) or NULL  */
    int32_t c0, ld0, off0;
    int32_t c1, ld1, off1;
    int32_t B, H, W;            /* source spatial size (before the optional x2 upsample)      */
    int32_t upsample;           /* 1: nearest x2 before the conv                              */
    int32_t KH, KW, stride, pad_h, pad_w;   /* pads may be negative (sub-grid origin)         */
    int32_t OH, OW;
    int32_t ndir;               /* 1, or 4: SS2D directions; dir k uses pad=(-(k&1), -(k>>1)),
                                   weight + k*w_dir_stride, out + k*out_dir_stride            */
    const void *weight;         /* [Cout][KH*KW*(c0+c1)], K order (kh,kw,c)                   */
    int64_t w_batch_stride;     /* elements between per-batch weight sets (0: shared)         */
    int64_t w_dir_stride;
    const float *bias;          /* [Cout] or NULL                                             */
    int32_t Cout;
    void *out;                  /* [B,OH,OW,ldo] channels [offo, offo+Cout)                   */
    int32_t ldo, offo;
    int64_t out_dir_stride;     /* elements                                                   */
    int32_t epilogue;
    int32_t epi_split;
    const void *res;            /* GATE_RES / RES_RELU: [B,OH,OW,ld_res] (dtype)              */
    int32_t ld_res, off_res;
    const float *gate;          /* GATE_RES: [B][gate_ld] fp32, entries gate[b*gate_ld + n]   */
    int32_t gate_ld;
    const void *h;              /* GNSILU_ADD: raw conv output [B,OH,OW,Cout] (dtype)          */
    const float *gn_mean_rstd;  /* GNSILU_ADD: [B][G][2]                                      */
    const float *gn_gamma, *gn_beta;
    int32_t gn_groups;
    float *stats_partial;       /* optional: per-(b, m-tile, channel) sum & sumsq of the
                                   stored values, [B][mtiles][Cout][2]; see fd_conv_mtiles    */
    /* LayerNorm over the Cin channels of in0 fused into the operand load.  Only the bf16
     * streaming row-GEMM path implements it: ask fd_conv_prologue_ok() first.
     *   LN_MOD : x' = LN(x)*(1+scale[b]) + shift[b]           src/DADiff.py:450-451,486-487
     *   LN_GATE: x' = LN(x)*z[m] + local[b]  (shift = local)  src/emamba2.py:365,747-748     */
    int32_t prologue;
    float ln_eps;
    const float *ln_gamma, *ln_beta;   /* [Cin] or NULL (LN_MOD without affine)              */
    const float *ln_shift, *ln_scale;  /* entries [b*ln_ld + c]                               */
    int32_t ln_ld;
    const void *ln_z;                  /* LN_GATE: [B,H,W,ln_ldz], channels [ln_offz, +Cin)   */
    int32_t ln_ldz, ln_offz;
    /* fp8 weights (BASELINE configs[4]): the same [Cout][KH*KW*Cin] matrix as OCP e4m3 bytes,
     * w_f8[n][k] = e4m3(weight[n][k] / w_scale[n]), one fp32 scale per output channel.  Used by the
     * 3x3 halo kernel where the K axis splits into 128-channel slabs (fd_conv_fp8_ok): the bf16 halo is
     * converted to e4m3 (x act_scale, a power of two) once per slab on its way into LDS and the product
     * runs on v_mfma_scale_f32_16x16x128_f8f6f4 (2x the bf16 rate), fp32 accumulation;
     * out = acc * w_scale[n] / act_scale + bias.  NULL: bf16 weights.  Other convs ignore these.       */
    const void *weight_f8;
    const float *w_scale;
    float act_scale;
    /* fp32 storage only: 1 = split-bf16 contraction (x = hi + lo in bf16, hi.hi + hi.lo + lo.hi on the bf16 MFMA,
     * fp32 accumulation: ~2^-16 per product) instead of the exact-f32 MFMA.  The parity mode leaves it 0.       */
    int32_t f32_split;
    /* FD_EPI_GNSILU_ADD_FINAL                                                                                  */
    const float *fin_w;                /* [Cout]                                              */
    float fin_b;
    float *fin_out;                    /* [B][H*W] fp32 model output                          */
    int32_t fin_mode, fin_last;        /* 0: fin_out only; 1: + DDIM update of fin_img        */
    float *fin_img;                    /* [B][H*W] fp32 x_t, updated in place                 */
    const float *fin_xin;              /* [B][H*W] fp32 x_input                               */
    float fin_alpha;
    /* FD_PRO_LN_GATE_ZRE (round 4): LN_GATE whose z operand is RECOMPUTED from the block input instead of read:
     *   z[m] = SiLU(zre_w . (LN(res[m]) * gamma (1 + scale[b]) + beta (1 + scale[b]) + shift[b]))
     * i.e. the z half of SS2D's in_proj (src/emamba2.py:716-719) applied to the adaLN-modulated norm1 of the Mamba
     * block's input (src/DADiff.py:450-451, 477-481), which out_proj reads anyway as its residual (`res`, GATE_RES).
     * z then never exists in HBM.  Needs Cin == 2 * Cout (d_inner = 2 * dim); ask fd_conv_prologue_ok().          */
    const void *zre_w;                 /* [Cin][Cout] (dtype): rows d_inner .. 2 d_inner - 1 of in_proj.weight    */
    const float *zre_gamma, *zre_beta; /* [Cout] norm1 affine or NULL                                             */
    const float *zre_shift, *zre_scale;/* entries [b*zre_ld + c]                                                  */
    int32_t zre_ld;
    float zre_eps;
    /* Up-sampling 3x3 convolutions (`upsample` = 1, KH = KW = 3: nn.Upsample(scale_factor=2, nearest) -> Conv2d,
     * src/DADiff.py:121-127) as four 2x2 convolutions on the source grid, one per output parity class (round 5, bf16 halo
     * kernel): weight_up2x[n][cls][r][c][ci], cls = 2 a + b for output pixel (2 i + a, 2 j + b), tap (r, c) reads source pixel
     * (i + a + r - 1, j + b + c - 1) with the weights of the 3x3 taps that fall on that pixel summed (a = 0: r = 0 <- kh 0,
     * r = 1 <- kh 1 + 2; a = 1: r = 0 <- kh 0 + 1, r = 1 <- kh 2; columns likewise), i.e. [Cout][16 * Cin] in dtype.  The same
     * convolution in exact arithmetic with 4 instead of 9 MACs per output.  NULL: the 9-tap form.  Other convs ignore it.
     * `upsample` = 2 (otherwise the same as 1) asks for one workgroup per (tile, parity class) instead of per tile: four times
     * the workgroups for a batch that does not fill the chip; the results are the same bits.                                 */
    const void *weight_up2x;
    /* fp32 storage with f32_split = 1, 3x3 / stride 1 / pad 1 (round 5): the weight matrix pre-split into its bf16 halves,
     * w = hi + lo with hi = bf16(w), lo = bf16(w - hi), each [Cout][9 * Cin] bf16.  With both set the convolution runs on the
     * halo-tiled kernel (three bf16 MFMA units per 64-channel slab: x_hi.w_hi + x_hi.w_lo + x_lo.w_hi) instead of the generic
     * split implicit GEMM; `weight` (fp32) stays the reference copy.  NULL: the generic form.                              */
    const void *weight_split_hi, *weight_split_lo;
    /* Both of the above at once (fp32 storage, f32_split = 1, `upsample` != 0, 3x3): the sub-pixel matrix of weight_up2x, built
     * from the fp32 weights and pre-split into bf16 halves, [Cout][16 * Cin] each: four 2x2 split-bf16 convolutions on the
     * source grid.  NULL: the 9-tap split form through the up-sampling index map.                                        */
    const void *weight_up2x_split_hi, *weight_up2x_split_lo;
} fd_conv_params;

/* 1 if fd_conv2d would run `p` (weight_f8 / w_scale set) on the fp8 MFMA path.                        */
int fd_conv_fp8_ok(const fd_conv_params *p);

#define FD_PRO_NONE 0
#define FD_PRO_LN_MOD 1
#define FD_PRO_LN_GATE 2
#define FD_PRO_LN_GATE_ZRE 3
/* 1 if this conv runs on the streaming row-GEMM kernel (1x1, bf16, Cin <= 256, >= 16384 pixels
 * per image, weights fit LDS) and may therefore carry a fused LN prologue.                    */
int fd_conv_prologue_ok(const fd_conv_params *p);

int fd_conv_mtiles(int OH, int OW);         /* number of m-tiles per image (for workspaces)  */
/* kernel fd_conv2d will run for p: 10 row-GEMM, 11 halo 3x3 (12: its fp8 form), 0..6 implicit-GEMM <BM,BN> =
 * <128,128>, <128,64>, <64,128>, <64,64>, <128,256>, <256,256>, <128,32> (profiling / roofline
 * bookkeeping; ids 4 and 5 depend on the batch size but give bitwise identical outputs)      */
int fd_conv_kernel_id(const fd_conv_params *p);
int fd_conv2d(const fd_conv_params *p, void *stream);

/* GroupNorm statistics from the conv's partial sums -> mean_rstd [B][G][2] (nn.GroupNorm,
 * src/DADiff.py:217,222).  */
int fd_gn_finalize(const float *stats_partial, int B, int mtiles, int C, int groups, int64_t hw,
                   float eps, float *mean_rstd, void *stream);
/* out = silu(GN(h)) (+ res)  -- Block tail + identity residual, src/DADiff.py:222-228, 430. */
int fd_gn_silu_apply(int dtype, const void *h, const float *mean_rstd, const float *gamma,
                     const float *beta, const void *res, void *out, int B, int64_t hw, int C,
                     int groups, void *stream);
/* The tail of an identity-residual ResnetBlock AND the down-sampling convolution behind it in one pass (round 4, bf16, C = 64):
 *   skip = x + SiLU(GroupNorm(h)) -- bit for bit fd_gn_silu_apply's result -- and
 *   out  = Conv2d(C, Cout, 4, stride 2, padding 1)(skip) + bias        (src/DADiff.py:128-131, 213-229, 418-430, 578-584)
 * h, x, skip [B,H,W,64]; mean_rstd [B][groups][2] from fd_gn_finalize; w [Cout][16 taps x 64] in K order (kh, kw, c) as for
 * fd_conv2d; out [B,H/2,W/2,Cout], Cout in {64, 128}; H % 16 == 0, W % 32 == 0.  The applied tensor is read back by nothing:
 * the convolution runs on the tile the apply left in LDS, with the weights resident in registers.                       */
int fd_gn_apply_down4x4_ok(int dtype_opts, int C, int Cout, int H, int W);
int fd_gn_apply_down4x4(int dtype, const void *h, const void *x, const float *mean_rstd, const float *gamma,
                        const float *beta, int groups, void *skip, const void *w, const float *bias, void *out,
                        int B, int H, int W, int C, int Cout, void *stream);

/* ---- LayerNorm family (channel-last rows) ------------------------------------------------
 * fd_ln_modulate: out = LN(x) * (1 + scale[b]) + shift[b]      src/DADiff.py:450-451, 486-487
 *   gamma/beta may be NULL (norm2: elementwise_affine=False).  */
int fd_ln_modulate(int dtype, const void *x, const float *gamma, const float *beta, float eps,
                   const float *shift, const float *scale, int mod_ld, void *out, int B,
                   int64_t hw, int C, void *stream);
/* fd_ln_gate: out = LN(y) * z + local[b]   (out_norm, y*z, +local)  src/emamba2.py:365, 747-748
 *   z has pixel stride ldz and channel offset offz (the z half of in_proj's output).         */
int fd_ln_gate(int dtype, const void *y, const float *gamma, const float *beta, float eps,
               const void *z, int ldz, int offz, const float *local, int local_ld, void *out,
               int B, int64_t hw, int C, void *stream);

/* ---- depthwise 3x3 conv, NHWC (SS2D conv2d + SiLU, qkv_dwconv)  src/emamba2.py:722,
 *      src/DADiff.py:266.  weight [9][C] fp32 (tap-major), bias [C] or NULL.                 */
int fd_dwconv3x3(int dtype, const void *in, int ld_in, int off_in, const float *weight,
                 const float *bias, int silu, void *out, int ld_out, int off_out, int B, int H,
                 int W, int C, void *stream);

/* ---- fused LayerNorm+modulate -> 1x1 conv -> depthwise 3x3 (bf16, Cin = 64, >= 32768 px/image)
 * Replaces, in one pass over the pixels, the pairs
 *   SS2D   in_proj + conv2d(3x3, groups=D)+SiLU (x half), SiLU (z half)   src/emamba2.py:716-722
 *   attn   qkv + qkv_dwconv                                               src/DADiff.py:266, 275
 * including the adaLN-modulated LayerNorm in front of them (src/DADiff.py:480-487).
 *   x      [B,H,W,ld_x] channels [off_x, +Cin)
 *   w_pw   bf16 [Cdw + Cz][Cin]: rows [0, Cdw) feed the depthwise conv, rows [Cdw, Cdw+Cz) are passed
 *          through SiLU to out_z (Cz = 0: none)
 *   w_dw   [9][Cdw/2] 32-bit words of fp16 tap weights, tap = 3*dy + dx, word j = (channel 2j low, 2j+1 high): the
 *          depthwise runs on v_pk_fma_f16 (the 1x1 output is kept in LDS as fp16 channel pairs, the 9-tap sum is
 *          accumulated in fp16 starting from the bias rounded to fp16 (toward zero); SiLU in fp32).  b_dw [Cdw] fp32 or NULL
 *   ln_*   as fd_conv_params' LN_MOD prologue (gamma/beta may be NULL)
 * fd_pw_dw3x3_ok: 1 if the shape is served (callers fall back to fd_conv2d + fd_dwconv3x3); `dtype | FD_OPT_LOW_LATENCY`
 *   answers for the one-slice kernel set (Cin = 128 stays unfused there).   */
int fd_pw_dw3x3_ok(int dtype, int Cin, int Cdw, int Cz, int H, int W);
int fd_pw_dw3x3(int dtype, const void *x, int ld_x, int off_x, int Cin, const float *ln_gamma,
                const float *ln_beta, float ln_eps, const float *ln_shift, const float *ln_scale,
                int ln_ld, const void *w_pw, int Cdw, const uint32_t *w_dw, const float *b_dw,
                int dw_silu, void *out_dw, int ld_dw, int off_dw, int Cz, void *out_z, int ld_z,
                int off_z, int B, int H, int W, void *stream);

/* ---- the qkv form of the above with the channel attention's statistics folded in (src/DADiff.py:266-276):
 * qkv (C -> 3C) -> qkv_dwconv -> per head L2 norms of q, k over all pixels and the 32x32 Gram q k^T.  q and k never
 * reach HBM: only v (out_v, [B,H,W,ld_v] channels [off_v, +64)) and one partial per workgroup do --
 * partial [B][2 heads][nblk][1024 + 64] fp32 in fd_chan_attn_gram's layout (Gram rows = q channels; then sum q^2,
 * sum k^2), nblk = fd_pw_dw3x3_gram_nblk(H, W), reduced in fixed order by fd_chan_attn_weff.  bf16, Cin = 64 (two
 * heads), w_pw [192][64] (q | k | v rows), w_dw [9][96] as for fd_pw_dw3x3, no bias; q and k are fp16 on chip (f16 MFMA):
 * their magnitudes must sit inside fp16's range -- they are L2-normalised per channel afterwards, so a per-channel
 * power-of-two scale folded into their w_pw rows / w_dw taps is free (founddiff_amd/engine.py does that at pack time). */
int fd_pw_dw3x3_gram_ok(int dtype, int Cin, int H, int W);
int fd_pw_dw3x3_gram_nblk(int H, int W);
/* ... with `dtype | FD_OPT_LOW_LATENCY` passed to fd_pw_dw3x3_gram (4 instead of 8 tiles per workgroup: the kernel set for ONE
 * slice, as for the scan) the partial count is fd_pw_dw3x3_gram_nblk_opts(dtype_opts, H, W).                         */
int fd_pw_dw3x3_gram_nblk_opts(int dtype_opts, int H, int W);
int fd_pw_dw3x3_gram(int dtype, const void *x, int ld_x, int off_x, int Cin, const float *ln_gamma,
                     const float *ln_beta, float ln_eps, const float *ln_shift, const float *ln_scale,
                     int ln_ld, const void *w_pw, const uint32_t *w_dw, void *out_v, int ld_v, int off_v,
                     float *partial, int B, int H, int W, void *stream);
/* out_v may be NULL (round 4): q and k only, for use with fd_pw_dw3x3_proj below, which recomputes v where it is
 * consumed.
 *
 * The v branch of the same attention through project_out and the block's gated residual (src/DADiff.py:266-285, 483-488):
 *   out = x + gate[b] . (w2[b] . dwconv3x3(w_pw . (LN(x) (1 + scale[b]) + shift[b])))
 * w_pw [64][64] (the v rows of qkv.weight), w_dw [9][32] fp16 channel pairs (the v taps of qkv_dwconv, fd_pw_dw3x3's
 * layout), w2 [B][64][64] = fd_chan_attn_weff's output, gate entries [b*gate_ld + n].  bf16, Cin = 64; v never reaches
 * HBM.  The depthwise output is rounded to bf16 before the second 1x1, as the stored v of the unfused sequence is.   */
int fd_pw_dw3x3_proj_ok(int dtype_opts, int Cin, int H, int W);
int fd_pw_dw3x3_proj(int dtype, const void *x, int ld_x, int off_x, int Cin, const float *ln_gamma,
                     const float *ln_beta, float ln_eps, const float *ln_shift, const float *ln_scale,
                     int ln_ld, const void *w_pw, const uint32_t *w_dw, const void *w2, const float *gate,
                     int gate_ld, void *out, int ld_o, int off_o, int B, int H, int W, void *stream);

/* ---- the three fused kernels above on FP32 STORAGE with split-bf16 contractions (round 6, fd_pwdw32.hip): the 64-channel
 * Mamba blocks of the `fp32s` engine (fp32 activations, every dense product as three bf16 MFMAs on hi / lo operand halves,
 * fp32 LayerNorm / depthwise / accumulation; the exact-f32 MFMA for the Gram).  Weights come as the checkpoint's fp32 values:
 *   w_pw_hi / w_pw_lo  bf16 [rows][64]: the checkpoint's fp32 1x1 weights split by the caller, hi = bf16(w), lo = bf16(w - hi);
 *   w_dw fp32 [9][ld_wdw] tap-major, column = w_pw row; b_dw fp32 [rows] or NULL
 * Same reference lines as the bf16 forms: src/emamba2.py:716-722 (in_proj x half + conv2d + SiLU; Cdw rows, z is recomputed by
 * fd_conv2d's FD_PRO_LN_GATE_ZRE), src/DADiff.py:266-276 (q, k rows 0..127 of qkv.weight -> qkv_dwconv -> per-head Gram and L2
 * norms; partial [B][2][nblk][1024 + 64] in fd_chan_attn_gram's layout, nblk = fd_pw_dw3x3_gram_f32_nblk(H, W)),
 * src/DADiff.py:266-285, 483-488 (v rows -> qkv_dwconv -> w2[b] = fd_chan_attn_weff's fp32 output -> x + gate . ()).
 * Cin = 64, H % 8 == 0, W % 16 == 0 (W % 8 for the Gram / project_out forms), >= 16384 px per image.               */
int fd_pw_dw3x3_f32_ok(int dtype, int Cin, int Cdw, int H, int W);
int fd_pw_dw3x3_f32(const void *x, int ld_x, int off_x, int Cin, const float *ln_gamma, const float *ln_beta, float ln_eps,
                    const float *ln_shift, const float *ln_scale, int ln_ld, const void *w_pw_hi, const void *w_pw_lo, int Cdw,
                    const float *w_dw, const float *b_dw, int dw_silu, void *out_dw, int ld_dw, int off_dw, int B, int H, int W, void *stream);
int fd_pw_dw3x3_gram_f32_ok(int dtype, int Cin, int H, int W);
int fd_pw_dw3x3_gram_f32_nblk(int H, int W);
int fd_pw_dw3x3_gram_f32(const void *x, int ld_x, int off_x, int Cin, const float *ln_gamma, const float *ln_beta, float ln_eps,
                         const float *ln_shift, const float *ln_scale, int ln_ld, const void *w_pw_hi, const void *w_pw_lo,
                         const float *w_dw, int ld_wdw, float *partial, int B, int H, int W, void *stream);
int fd_pw_dw3x3_proj_f32_ok(int dtype, int Cin, int H, int W);
int fd_pw_dw3x3_proj_f32(const void *x, int ld_x, int off_x, int Cin, const float *ln_gamma, const float *ln_beta, float ln_eps,
                         const float *ln_shift, const float *ln_scale, int ln_ld, const void *w_pw_hi, const void *w_pw_lo,
                         const float *w_dw, int ld_wdw, const float *w2, const float *gate, int gate_ld, void *out, int ld_o, int off_o, int B,
                         int H, int W, void *stream);

/* ---- qkv_dwconv + L2 norms + q k^T for the wider blocks (C >= 128, src/DADiff.py:267-276): the depthwise 3x3 of the q
 * and k channels of a qkv tensor [B,H,W,ld] (q at channel 0, k at channel C) feeding the per-head Gram directly -- q and k
 * after the depthwise conv never reach HBM.  v keeps fd_dwconv3x3.  w_dw [9][3C/2] in fd_pw_dw3x3's fp16 layout;
 * partial [B][C/32][nblk][1024 + 64] in fd_chan_attn_gram's layout, nblk = fd_dwconv_gram_nblk(H, W).            */
int fd_dwconv_gram_ok(int dtype, int C, int H, int W);
int fd_dwconv_gram_nblk(int H, int W);
int fd_dwconv_gram(int dtype, const void *qkv, int ld, int C, const uint32_t *w_dw, float *partial, int B, int H, int W,
                   void *stream);

/* ---- SS2D selective scan (replaces selective_scan_cuda_core.fwd, src/emamba2.py:154, together
 * with EfficientScan/EfficientMerge index maps 182-262, dt_proj einsum 340, softplus/bias).
 *   xc    [B,H,W,D]    (dtype)  dwconv+SiLU output, D = d_inner = 2C
 *   xdbl  [4,B,L,CD]   fp32     x_proj output per direction, CD = R + 2N, row l' = h'*ceil(W/2) + w',
 *                               L = ceil(H/2)*ceil(W/2); odd H / W: the reference's zero padding (src/emamba2.py:
 *                               191-199, 253-260) -- rows of sub-grid positions outside the image must be zero
 *   dtw   [4,D,R], dtb [4,D], A [4*D,N] (= -exp(A_logs)), Ds [4*D]   fp32
 *   y     [B,H,W,D]    (dtype)  written at the merged pixel positions
 *   ws    fp32 workspace of fd_scan_ws_floats(...) floats
 * 3 phases over chunks of the sequence (L = H*W/4 per direction): local scan, carry scan,
 * final scan + C contraction + D skip.  fp32 state throughout.  Short sequences with a wide state
 * (L <= 1024, N >= 16, R % 8 == 0: the 64x64 level of a 512x512 slice) run ONE sequential pass instead, 4 lanes
 * per channel sharing the states and the dt_proj contraction -- no chunks, no workspace traffic; which form
 * runs depends on (H, W, N, R) only, never on B.                                              */
/* OR-ed into the `dtype` argument of fd_selective_scan / fd_selective_scan_plan: keep the chunked 3-phase form at every
 * size.  The single-pass form of short sequences is the faster one once a batch fills the chip (x1.7 at 8 slices of
 * 512x512) but walks its 1024 positions with one wave per channel group: a lone slice takes 2-3x longer in it.  An
 * engine uses ONE of the two for all its calls (DAEngine low_latency), so results stay batch-invariant within it.  */
#define FD_OPT_LOW_LATENCY 0x100
/* OR-ed into the `dtype` argument of fd_selective_scan_plan / fd_selective_scan_fuses_xproj / fd_selective_scan_xproj with
 * FD_F32 (round 6): the caller's fp32-storage engine runs its contractions split-bf16 (fd_conv_params.f32_split), so the
 * x_proj einsum (src/emamba2.py:332) may run inside the scan's first phase as three bf16 MFMAs per product, x_proj_w fp32
 * [4][R + 2N][d_inner].  Without it an FD_F32 scan takes its x_dbl rows from the workspace (exact parity mode).          */
#define FD_OPT_F32_SPLIT 0x200
int64_t fd_scan_ws_floats(int B, int H, int W, int D, int N);
int fd_selective_scan(int dtype, const void *xc, const float *xdbl, const float *dtw,
                      const float *dtb, const float *A, const float *Ds, void *y, float *ws,
                      int B, int H, int W, int D, int N, int R, void *stream);
/* The same scan with the x_proj einsum (src/emamba2.py:332) folded into its first phase: x_proj_w [4][R+2N][D] in
 * dtype; xdbl is then an OUTPUT workspace (phase A writes the rows phase C reads).  Saves the separate x_proj
 * launch and its pass over xc.  Only where fd_selective_scan_fuses_xproj() says so (bf16, d_inner <= 256).     */
int fd_selective_scan_fuses_xproj(int dtype, int D, int N, int R);
int fd_selective_scan_xproj(int dtype, const void *xc, const void *x_proj_w, float *xdbl, const float *dtw,
                            const float *dtb, const float *A, const float *Ds, void *y, float *ws, int B, int H,
                            int W, int D, int N, int R, void *stream);
/* 1: call fd_selective_scan_xproj for this block; 0: run the x_proj launch, then fd_selective_scan (the single-pass
 * form takes its x_dbl rows from the workspace).                                                              */
int fd_selective_scan_plan(int dtype, int D, int N, int R, int H, int W);

/* ---- The reference's own native-op interface (the only one it has):
 *     out, x, *rest = selective_scan_cuda_core.fwd(u, delta, A, B, C, D, delta_bias, delta_softplus, nrows)
 * /root/reference/src/emamba2.py:154, layout asserted at 124-149.  All tensors fp32, contiguous in the last dim:
 *   u, delta [batch,KD,L]   A [KD,N]   B, C [batch,K,N,L] (K groups, channel d uses group d / (KD/K))
 *   D, delta_bias [KD] (either may be NULL)      out [batch,KD,L]
 *   x_last [batch,KD,N] or NULL: the state after the last position (the extension's second return value holds
 *   the chunk states its backward pass needs; this library is forward-only and returns the final state instead)
 * nrows is the extension's rows-per-block tuning knob: validated like the reference does (1..4, KD % (K*nrows)
 * == 0, src/emamba2.py:129-130) and otherwise without effect.  Python shim with the extension's module and
 * function name: founddiff_amd/selective_scan_cuda_core.py.                                                 */
int fd_selective_scan_fwd_f32(const float *u, const float *delta, const float *A, const float *B,
                              const float *C, const float *D, const float *delta_bias, int delta_softplus,
                              int nrows, int batch, int KD, int K, int N, int64_t L, float *out, float *x_last,
                              void *stream);

/* ---- channel ("transposed") attention, src/DADiff.py:263-285 ------------------------------
 * fd_chan_attn_gram: per (b, head) partial 32x32 Gram q^T k and sums of squares over pixel
 *   blocks.  qkv [B,HW,3C] (dtype).  partial: fp32 [B][heads][nblk][32*32+64].
 * fd_chan_attn_weff: reduce partials, L2-normalise, *temperature, softmax, and fold the
 *   result into project_out:  Weff[b][o][h*32+j] = sum_i Wp[o][h*32+i] * attn[b,h,i,j]
 *   so that attn@v followed by project_out is ONE GEMM over v (fd_conv2d, w_batch_stride).   */
int fd_chan_attn_nblk(int64_t hw);
int fd_chan_attn_gram(int dtype, const void *qkv, int B, int64_t hw, int C, float *partial,
                      void *stream);
/* (`partial` is reduced IN PLACE over the pixel blocks: slot 0 of each (b, head) holds the sums) */
int fd_chan_attn_weff(int dtype, float *partial, int nblk, const float *temperature,
                      const float *wproj /* [C][C] fp32 */, void *weff /* [B][C][C] dtype */,
                      int B, int C, void *stream);

/* ---- small fp32 dense layers and conditioning ---------------------------------------------
 * fd_linear: out[m,n] = act(b[n] + sum_k x[m,k] w[n,k]);  act: 0 none, 1 SiLU, 2 GELU(erf),
 *   3 ReLU.  pre_silu: apply SiLU to x first (adaLN_modulation = Linear(SiLU(t))).
 *   time MLP (src/DADiff.py:580-585), text/prompt MLPs (606-611), adaLN (463-466),
 *   SS2D `attn` (src/emamba2.py:522-525), DA-CLIP heads (src/DACLIP.py:1179-1188).           */
int fd_linear(const float *x, const float *w, const float *b, float *out, int M, int N, int K,
              int act, int pre_silu, void *stream);
/* sinusoidal embedding, src/DADiff.py:173-185: out[b] = [sin(t f_i), cos(t f_i)]             */
int fd_sinusoidal(const float *time, float *out, int B, int dim, void *stream);
/* out = softmax(x, dim=1) * p[n]   (prompt path, src/DADiff.py:706)                         */
int fd_softmax_mul(const float *x, const float *p, float *out, int M, int N, void *stream);
/* out = x / max(||x||_2, eps) per row                                                       */
int fd_l2norm_rows(const float *x, float *out, int M, int N, float eps, void *stream);
int fd_add_f32(const float *a, const float *b, float *out, int64_t n, void *stream);

/* ---- image <-> activation glue ------------------------------------------------------------
 * fd_pack_planes: NCHW fp32 planes -> NHWC `cpad` channels (dtype), zero padded
 *   (torch.cat((x, x_input), 1) src/DADiff.py:1160; x[:,1].repeat(1,3,..) 692 is folded).    */
int fd_pack_planes(int dtype, const float *p0, const float *p1, void *out, int B, int64_t hw,
                   int cpad, void *stream);
/* fd_init_conv7: the UNet's init_conv (7x7, pad 3, bias) straight from the fp32 image planes
 *   (src/DADiff.py:558, 704) -- bf16 mode, Cout in {32, 64}, H and W multiples of 16
 *   (fd_init_conv7_ok; otherwise fd_pack_planes3 + fd_conv2d).
 *   p0, p1, p2: [B,H,W] fp32 planes (x_t, x_input, x_input_condition; p1 / p2 may be NULL)
 *   w_packed: bf16 [Cout][7 kh][8 kw][4 c], zero where kw = 7 or c >= number of planes -- except
 *             that with p2 == NULL slots c = 2, 3 must repeat the weights of planes 0, 1: the kernel
 *             feeds them the bf16 rounding residuals of the two planes (hi + lo split)
 *   out: [B,H,W,Cout] bf16                                                                    */
int fd_init_conv7_ok(int dtype, int Cout, int H, int W);
int fd_init_conv7(int dtype, const float *p0, const float *p1, const float *p2, const void *w_packed,
                  const float *bias, void *out, int B, int H, int W, int Cout, void *stream);
/* The same layer on fp32 storage with the split-bf16 contraction of the `fp32s` engine (round 6), two input planes, fp32 out:
 * w_hi_packed = fd_init_conv7's packing of bf16(w) (all four channel slots: planes and their bf16 rounding residuals),
 * w_lo_packed = the same packing of bf16(w - bf16(w)) with slots 2, 3 zero: w_hi.x_hi + w_hi.x_lo + w_lo.x_hi.         */
int fd_init_conv7_f32s(const float *p0, const float *p1, const void *w_hi_packed, const void *w_lo_packed, const float *bias,
                       void *out, int B, int H, int W, int Cout, void *stream);
/* same with a third plane: torch.cat((x, x_input, x_input_condition), 1)  src/DADiff.py:1157-1158
 * (p1, p2 may be NULL)                                                                        */
int fd_pack_planes3(int dtype, const float *p0, const float *p1, const float *p2, void *out, int B,
                    int64_t hw, int cpad, void *stream);
/* final 1x1 conv to ONE channel (src/DADiff.py:683,740): out[b,p] = b0 + sum_c x[b,p,c] w[c] */
/* n elements (n % 8 == 0) of an activation tensor from one storage type to the other (FD_F32 <-> FD_BF16): the
 * level boundary of a hybrid-precision forward (engine.py: DAEngine.forward_hybrid).                       */
int fd_cast(int src_dtype, const void *in, int dst_dtype, void *out, int64_t n, void *stream);
int fd_final_conv1(int dtype, const void *x, const float *w, const float *b, float *out,
                   int64_t npix, int C, void *stream);
/* avg-pool k x k stride k, NHWC (src/DACLIP.py:181,193,282)                                  */
int fd_avgpool(int dtype, const void *in, void *out, int B, int H, int W, int C, int k,
               void *stream);
/* attention-pool helpers (src/DACLIP.py:226-259): tokens = [mean; x]  ->  [B, HW+1, C]      */
int fd_attnpool_tokens(int dtype, const void *x, void *tok, int B, int64_t hw, int C,
                       void *stream);
/* one-query multi-head attention: q rows of stride q_ld, k/v [B,T,ld] fp32 -> out [B,C] fp32 */
int fd_attnpool_core(const float *q, int q_ld, const float *kv, int ld, int koff, int voff, float *out,
                     int B, int T, int C, int heads, void *stream);

/* ---- scheduler math (fp32 images, shape [B, npix]) ----------------------------------------
 * fd_res_predictions: pred_res = clamp(out); pred_noise = (x_t - x_in - (ac-1) pred_res)/bc;
 *   x_start = clamp(x_in - pred_res)          src/DADiff.py:1202-1207, 1120-1124
 *   ac, bc: per-batch alphas_cumsum[t], betas_cumsum[t] (device, [B]).                       */
int fd_res_predictions(const float *model_out, const float *x_t, const float *x_in,
                       const float *ac, const float *bc, float *pred_res, float *pred_noise,
                       float *x_start, int B, int64_t npix, void *stream);
/* fd_res_ddim_step: img' = last ? clamp(x_in - clamp(out)) : img - alpha*clamp(out) + sigma*noise
 *   src/DADiff.py:1317-1318, 1344 (type "use_pred_noise").  noise may be NULL (eta = 0).     */
int fd_res_ddim_step(const float *model_out, const float *img, const float *x_in,
                     const float *noise, float alpha, float sigma, int last, float *img_out,
                     int64_t n, void *stream);
/* fd_res_step_obj: the same three operations for every objective of model_predictions
 *   (src/DADiff.py:1168-1207) and the dual-UNet model (817-836).
 *   mode 0: o0 = residual (pred_res; pred_res_noise tested as "res")
 *        1: o1 = noise    (pred_noise; pred_res_noise tested as "noise"): x_start =
 *           clamp((x_t - ac x_in - bc o1)/omac), pred_res = clamp(x_in - x_start)      1126-1130
 *        2: o0 = residual, o1 = noise: x_start = clamp(x_t - ac clamp(o0) - bc o1)     1132-1136
 *        3: o0 = x_0, o1 = noise (pred_x0_noise): pred_res = clamp(x_in - o0)          1191-1195
 *   par [B][8] = {ac, bc, omac, k0, k1, k2, k3, flag} (alphas_cumsum[t], betas_cumsum[t],
 *        one_minus_alphas_cumsum[t] per batch element)
 *   step 0: predictions only (any of pred_res / pred_noise / x_start may be NULL)
 *        1: DDIM: img_out = flag ? x_start : x_t - k0 pred_res + k1 noise             1317-1318, 1344
 *        2: posterior: img_out = k0 x_t + k1 pred_res + k2 x_start + exp(k3/2) noise  1142-1151, 1226-1229
 *   noise may be NULL.                                                                        */
int fd_res_step_obj(int mode, int step, const float *o0, const float *o1, const float *x_t,
                    const float *x_in, const float *noise, const float *par, float *pred_res,
                    float *pred_noise, float *x_start, float *img_out, int B, int64_t npix,
                    void *stream);
/* fd_res_posterior_step: mean = c1 x_t + c2 pred_res + c3 x_start; out = mean + exp(.5 lv) noise
 *   src/DADiff.py:1142-1151, 1226-1229.  coef: [B][4] = c1,c2,c3,logvar.  noise NULL at t=0. */
int fd_res_posterior_step(const float *model_out, const float *x_t, const float *x_in,
                          const float *noise, const float *coef, float *img_out,
                          float *x_start_out, int B, int64_t npix, void *stream);
/* ---- ancestral sampling without a host in the loop, step noise keyed per slice (fd_sched.hip)
 * The reference draws randn_like(x) from the device generator at every step (src/DADiff.py:1228): a slice's noise
 * depends on its place in the batch.  Here noise(slice, t, pixel) = Box-Muller(Philox4x32-10(key = seeds[b] (64 bit),
 * counter = (pixel / 4, t, 0x46444e5a, 0))): the same slice gets the same stream on any rank / batch / stream
 * (BASELINE configs[3]: a volume sharded over 8 GPUs).  oracle/keyed_noise.py restates it.
 *   fd_keyed_normal               out[b][i] = that noise (x_T draws with t = 0x7fffffff, tests)
 *   fd_ancestral_begin            *t_dev -= 1;  time_buf[b] = times[*t_dev]   (times = alphas_cumsum * T, device)
 *   fd_res_posterior_step_keyed   fd_res_posterior_step with t = *t_dev, coefficients coef_table[t][4] = c1, c2, c3,
 *                                 logvar (device, [T][4]) and the keyed noise (none at t = 0)
 * All three read the step from DEVICE memory: a chunk of steps captures into one HIP graph.                   */
int fd_keyed_normal(const int64_t *seeds, int t, float *out, int B, int64_t npix, void *stream);
int fd_ancestral_begin(int *t_dev, const float *times, float *time_buf, int B, void *stream);
int fd_res_posterior_step_keyed(const float *model_out, const float *x_t, const float *x_in,
                                const float *coef_table, const int *t_dev, const int64_t *seeds,
                                float *img_out, float *x_start_out, int B, int64_t npix, void *stream);
/* ---- vanilla DDPM U-Net extras (src/denoising_diffusion_pytorch.py) -----------------------
 * fd_gn_film_silu_apply: silu(GN(h)*(1+scale[b]) + shift[b])   Block w/ scale_shift, 190-199, 213-221
 * fd_chan_ln:            LN over channels * g (+ res)           LayerNorm/PreNorm/Residual, 95-101,127-146
 * fd_linear_attention:   LinearAttention core (227-255) on qkv [B,hw,3*hidden] (to_qkv output):
 *     k softmax over pixels, context = k v^T / hw folded with to_out's weight into
 *     wtot [B][C][hidden] (dtype); q is replaced IN PLACE by softmax_d(q)*scale, so that
 *     to_out(context^T q) is one fd_conv2d over the q slice with per-batch weights.
 *     kstats: fp32 [B][hidden][2], ctx: fp32 [B][hidden/32][32][32] workspaces.
 * fd_attention:          softmax attention, dim_head 32 (257-279): qkv [B,n,3*hidden] -> out [B,n,hidden]
 * fd_lincomb3:           out = ca*a + cb*b + cc*c (b, c may be NULL), optional clamp to [-1,1]:
 *     GaussianDiffusion.predict_* / q_posterior / ddim update (523-554, 633-643)              */
int fd_gn_film_silu_apply(int dtype, const void *h, const float *mean_rstd, const float *gamma,
                          const float *beta, const float *film_scale, const float *film_shift,
                          int film_ld, void *out, int B, int64_t hw, int C, int groups, void *stream);
int fd_chan_ln(int dtype, const void *x, const float *g, const void *res, void *out, int64_t nrows,
               int C, void *stream);
int fd_linear_attention(int dtype, void *qkv, int B, int64_t hw, int hidden, const float *wout,
                        float *kstats, float *ctx, void *wtot, int C, void *stream);
int fd_attention(int dtype, const void *qkv, void *out, int B, int64_t n, int hidden, void *stream);
int fd_lincomb3(const float *a, const float *b, const float *c, float ca, float cb, float cc,
                int clamp, float *out, int64_t n, void *stream);

/* ---- evaluation metrics on device (src/util.py:188-236, used by Trainer.test src/DADiff.py:1883-1885)
 * pred/target [B,H,W] fp32 in [0,1].  partial: fp32 workspace [B][fd_metrics_nblk(H,W)][2].
 * out [B][3] = PSNR (max_val 1), SSIM (11x11 Gaussian sigma 1.5, reflect pad, clamp, mean), RMSE. */
int fd_metrics_nblk(int H, int W);
int fd_metrics(const float *pred, const float *target, int B, int H, int W, float *partial,
               float *out, void *stream);

/* out = a*x + b  (normalize_to_neg_one_to_one / unnormalize, src/DADiff.py:109-120)          */
int fd_affine_f32(const float *x, float a, float b, float *out, int64_t n, void *stream);
/* out = x + s*noise   (x_T = x_input + sqrt(sum_scale) eps, src/DADiff.py:1294)              */
int fd_axpy_f32(const float *x, const float *noise, float s, float *out, int64_t n, void *stream);

#ifdef __cplusplus
}
#endif
#endif
