// fd_conv.hip -- implicit-GEMM convolution / GEMM on MFMA for gfx950.
//
// One kernel serves every dense contraction of the denoiser (see include/founddiff_hip.h):
// M = output pixels of one image, N = Cout, K = KH*KW*Cin with (kh,kw,c) ordering so that a
// 16-byte chunk of the K axis is 8 (bf16) / 4 (f32) consecutive channels of ONE input pixel:
// coalesced NHWC loads, no im2col buffer.  The A operand may come from two tensors (skip
// concat), through a nearest x2 upsample, at stride 2 with a sub-grid origin (SS2D
// directions), and with a pixel stride / channel offset (slices of wider tensors).
//
// Tile: 128 (pixels) x BN (64|128 channels) x 128 bytes of K per step, 256 threads = 4 waves
// in 2x2, each wave 64 x BN/2 as 16x16 MFMA tiles (bf16: v_mfma_f32_16x16x32_bf16; f32 parity
// mode: v_mfma_f32_16x16x4_f32, exact f32).  Global -> registers -> LDS double buffer, one
// barrier per K step; LDS rows are 128 B with a 16-byte-chunk XOR swizzle
// (chunk ^= (row>>1)&7) that makes the bf16 ds_read_b128 fragment reads conflict-free.
// The accumulators are staged through LDS so the epilogue works on row-contiguous 8-channel
// vectors: 16/32-byte stores, vector loads of the residual / GroupNorm operand, and
// per-channel partial sums for the next GroupNorm.
#include <stdlib.h>
#include "fd_common.h"

namespace {

constexpr int ROWB = 128;  // bytes of K per tile row
// pixel-tile height: 128 rows for big images, 64 when an image has few pixels (low-resolution
// levels at B=1 would otherwise launch fewer workgroups than there are CUs).  Depends on the
// image size ONLY, so workspaces sized with fd_conv_mtiles() and results are batch-invariant.
static inline int conv_bm(int64_t ohw) { return ohw <= 16384 ? 64 : 128; }

template <typename T> struct Frag;
template <> struct Frag<bf16> {
    static constexpr int KSTEPS = 2;  // 64 bf16 per row = 2 x K32
};
template <> struct Frag<float> {
    static constexpr int KSTEPS = 1;  // 32 f32 per row
};

__device__ __forceinline__ int swz(int row, int chunk) { return (chunk ^ ((row >> 1) & 7)) << 4; }

// WM x WN waves per workgroup, each owning a (BM/WM) x (BN/WN) sub-tile.  Configurations built:
//   <128,128,2,2> <128,64,2,2>   big images (> 16384 pixels)
//   <64,128,2,2>  <64,64,2,2>    small images, Cout < 256
//   <128,256,2,4>                small images, Cout >= 256: 8 waves, 64x64 per wave -- the dense layers of the
//                                64x64 / 128x128 levels re-read their operands from beyond L2 (weights up to
//                                7 MB, A up to 450 MB per launch at batch 8): bytes per FLOP scale with
//                                1/BM + 1/BN, so the larger tile halves that traffic.
// GroupNorm partial sums are always written per 64-row band (one epilogue pass per wave row), so the
// workspace layout [B][ceil(OHW/64)..] does not depend on the tile height.
// PW = pointwise fast path (1x1, stride 1, no padding / upsampling / directions, Cin and c0 multiples of
// the K tile): the gather degenerates to "row m, channels k..k+63", so every per-lane offset is computed
// once and a K step only advances a wave-uniform base pointer (SGPR base + 32-bit VGPR offset loads) --
// the general decode costs ~80 VALU instructions per K step against 16-32 MFMAs.
// SPL (fp32 storage only): split-bf16 contraction.  Each f32 operand is split into hi = bf16(x) and lo = bf16(x - hi)
// on its way into LDS (the 32 floats of a tile row become 32 hi + 32 lo bf16 in the same 128 bytes) and the product
// is hi.hi + hi.lo + lo.hi on the bf16 MFMA with fp32 accumulation: ~2^-16 relative per product instead of the
// exact-f32 MFMA's 2^-24, at 3 x 16 instead of 8 x 32 matrix-pipe cycles per 32-deep step.  It serves the
// fp32-storage levels of a sampling loop's LAST step in the bf16 production mode (DAEngine mode 'fp32s'); the parity
// mode ('fp32', the 1e-3 gate) never sets it.
template <typename T, int BM, int BN, int WM, int WN, bool PW = false, bool SPL = false>
__global__ __launch_bounds__(64 * WM * WN) void conv_igemm_kernel(const fd_conv_params p) {
    static_assert(!SPL || (sizeof(T) == 4 && !PW), "split-bf16 contraction: fp32 storage, general loader");
    constexpr int NTHR = 64 * WM * WN, NWAVE = WM * WN;
    constexpr int RPL = NTHR / 8;            // tile rows covered by one loader pass (8 chunks per row)
    constexpr int AR = BM / RPL, NB = BN / RPL;
    constexpr int TMW = BM / WM, TNW = BN / WN;
    constexpr int MT = TMW / 16, NT = TNW / 16;
    constexpr bool PWE = PW && (NT % 2 == 0);     // transposed issue + direct epilogue need pairs of 16-channel tiles
    constexpr int CH = TT<T>::CH;
    constexpr int BK = ROWB / (int)sizeof(T);
    constexpr int AB_BYTES = 2 * (BM + BN) * ROWB;
    constexpr int C_BYTES = TMW * BN * 4;    // accumulators are staged one wave-row (TMW rows) at a time
    constexpr int SM_BYTES = AB_BYTES > C_BYTES ? AB_BYTES : C_BYTES;
    static_assert(TMW == 64 || (BM == 64 && TMW == 32), "GroupNorm partials are written per 64-row band");
    __shared__ __attribute__((aligned(16))) unsigned char smem[SM_BYTES];
    __shared__ float s_stat[NWAVE][BN][2];
    // pointwise epilogue: the per-channel operands (bias; gate | GroupNorm gamma, beta, mean, rstd) of the channel tile,
    // fetched once at kernel start.  Loaded inside the epilogue they were 4-8 dependent global round trips per workgroup
    // AFTER its last MFMA (one per 32-channel group), sitting behind whatever else was in flight -- with the K loop
    // removed the kernel still took 40 % of its time
    __shared__ __attribute__((aligned(16))) float s_ep[PWE ? 5 : 1][PWE ? BN : 1];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int mt = blockIdx.x, nt = blockIdx.y;
    const int b = blockIdx.z / p.ndir, dir = blockIdx.z % p.ndir;
    if constexpr (PWE) {
        for (int c = tid; c < BN; c += NTHR) {
            const int n = nt * BN + c;
            const bool ok = n < p.Cout;
            s_ep[0][c] = (p.bias && ok) ? p.bias[n] : 0.f;
            float e0 = 0.f, e1 = 0.f, gm = 0.f, gr = 0.f;
            if (ok && p.epilogue == FD_EPI_GATE_RES) e0 = p.gate[(int64_t)b * p.gate_ld + n];
            else if (ok && p.epilogue == FD_EPI_GNSILU_ADD) {
                e0 = p.gn_gamma[n];
                e1 = p.gn_beta[n];
                const int g = n / (p.Cout / p.gn_groups);
                gm = p.gn_mean_rstd[((int64_t)b * p.gn_groups + g) * 2];
                gr = p.gn_mean_rstd[((int64_t)b * p.gn_groups + g) * 2 + 1];
            }
            s_ep[1][c] = e0; s_ep[2][c] = e1; s_ep[3][c] = gm; s_ep[4][c] = gr;
        }
        // (published by the barriers of the K loop: every launch has at least one K step)
    }
    const int Cin = p.c0 + p.c1;
    const int K = p.KH * p.KW * Cin;
    const int OHW = p.OH * p.OW;
    const int pad_h = p.ndir > 1 ? -(dir & 1) : p.pad_h;
    const int pad_w = p.ndir > 1 ? -(dir >> 1) : p.pad_w;
    const T *__restrict__ in0 = (const T *)p.in0;
    const T *__restrict__ in1 = (const T *)p.in1;
    const T *__restrict__ wgt = (const T *)p.weight + (int64_t)b * p.w_batch_stride + (int64_t)dir * p.w_dir_stride;
    const int Hs = p.upsample ? 2 * p.H : p.H, Ws = p.upsample ? 2 * p.W : p.W;

    // ---- loader roles
    const int chunk = tid & 7, rbase = tid >> 3;
    int ihb[AR], iwb[AR];
#pragma unroll
    for (int i = 0; i < AR; ++i) {
        int m = mt * BM + rbase + RPL * i;
        if (m < OHW) {
            int oh = m / p.OW, ow = m - oh * p.OW;
            ihb[i] = oh * p.stride - pad_h;
            iwb[i] = ow * p.stride - pad_w;
        } else {
            ihb[i] = -(1 << 28);
            iwb[i] = 0;
        }
    }
    u32x4 ra[AR], rb[NB];
    const int nkt = (K + BK - 1) / BK;

    // (kh, kw, channel base) of the next K tile, kept as wave-uniform counters: when the tile
    // lies inside one filter tap (Cin % BK == 0: every large layer) the per-tile address math
    // has no integer division; tiles that straddle taps (tiny Cin) take the general decode.
    int t_kh = 0, t_kw = 0, t_cb = 0;
    const int64_t img_px = (int64_t)p.H * p.W;
    const T *in0b = in0 + (int64_t)b * img_px * p.ld0;
    const T *in1b = in1 ? in1 + (int64_t)b * img_px * p.ld1 : nullptr;
    unsigned aoff0[AR], adelta[AR], woff[NB];   // adelta = offset in source 1 minus offset in source 0
    bool aok[AR], wok[NB];
    if constexpr (PW) {
#pragma unroll
        for (int i = 0; i < AR; ++i) {
            const int m = mt * BM + rbase + RPL * i;
            aok[i] = m < OHW;
            const int mm = aok[i] ? m : 0;
            aoff0[i] = (unsigned)(mm * p.ld0 + chunk * CH);
            adelta[i] = (unsigned)(mm * p.ld1) - (unsigned)(mm * p.ld0);
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int n = nt * BN + rbase + RPL * i;
            wok[i] = n < p.Cout;
            woff[i] = (unsigned)((wok[i] ? n : 0) * K + chunk * CH);
        }
    }
    auto gload = [&](int kt) {
        if constexpr (PW) {
            const int kb = kt * BK;                    // wave-uniform
            const bool from0 = kb < p.c0;
            const unsigned sel = from0 ? 0u : ~0u;     // uniform mask instead of a select between arrays
            const T *src = from0 ? in0b + p.off0 + kb : in1b + p.off1 + (kb - p.c0);
            const T *wsrc = wgt + kb;
#pragma unroll
            for (int i = 0; i < AR; ++i) {
                u32x4 v = {0, 0, 0, 0};
                if (aok[i]) v = *(const u32x4 *)(src + (aoff0[i] + (adelta[i] & sel)));
                ra[i] = v;
            }
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                u32x4 v = {0, 0, 0, 0};
                if (wok[i]) v = *(const u32x4 *)(wsrc + woff[i]);
                rb[i] = v;
            }
            return;
        }
        int kh, kw, c;
        bool kv;
        if (t_cb + BK <= Cin) {
            kh = t_kh; kw = t_kw; c = t_cb + chunk * CH;
            kv = t_kh < p.KH;
        } else {
            const int k = (t_kh * p.KW + t_kw) * Cin + t_cb + chunk * CH;
            kv = k < K;
            const int tap = k / Cin;
            c = k - tap * Cin;
            kh = tap / p.KW;
            kw = tap - kh * p.KW;
        }
        const int k = (kh * p.KW + kw) * Cin + c;
        const T *src;
        int ld, coff;
        if (c < p.c0) { src = in0b; ld = p.ld0; coff = p.off0 + c; }
        else { src = in1b; ld = p.ld1; coff = p.off1 + c - p.c0; }
#pragma unroll
        for (int i = 0; i < AR; ++i) {
            int ih = ihb[i] + kh, iw = iwb[i] + kw;
            bool ok = kv && ih >= 0 && ih < Hs && iw >= 0 && iw < Ws;
            if (p.upsample) { ih >>= 1; iw >>= 1; }
            u32x4 v = {0, 0, 0, 0};
            if (ok) v = *(const u32x4 *)(src + ((ih * p.W + iw) * ld + coff));
            ra[i] = v;
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            int n = nt * BN + rbase + RPL * i;
            u32x4 v = {0, 0, 0, 0};
            if (kv && n < p.Cout) v = *(const u32x4 *)(wgt + (int64_t)n * K + k);
            rb[i] = v;
        }
        t_cb += BK;
        while (t_cb >= Cin) {
            t_cb -= Cin;
            if (++t_kw == p.KW) { t_kw = 0; ++t_kh; }
        }
    };
    auto lstore = [&](int buf) {
        unsigned char *sA = smem + buf * (BM + BN) * ROWB;
        unsigned char *sB = sA + BM * ROWB;
        if constexpr (SPL) {
            // 4 floats (k = 4*chunk .. +3) -> 4 hi (8 bytes at k-chunk chunk/2, half chunk&1) + 4 lo (k-chunk 4 + chunk/2)
            auto split_store = [&](unsigned char *row, int r, const u32x4 v) {
                const uint32_t raw[4] = {v.x, v.y, v.z, v.w};
                uint32_t hw[2], lw[2];
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const float f0 = __builtin_bit_cast(float, raw[2 * e]), f1 = __builtin_bit_cast(float, raw[2 * e + 1]);
                    const bf16 h0 = (bf16)f0, h1 = (bf16)f1;
                    const bf16 l0 = (bf16)(f0 - (float)h0), l1 = (bf16)(f1 - (float)h1);
                    hw[e] = (uint32_t)__builtin_bit_cast(uint16_t, h0) | ((uint32_t)__builtin_bit_cast(uint16_t, h1) << 16);
                    lw[e] = (uint32_t)__builtin_bit_cast(uint16_t, l0) | ((uint32_t)__builtin_bit_cast(uint16_t, l1) << 16);
                }
                *(uint2 *)(row + swz(r, chunk >> 1) + (chunk & 1) * 8) = uint2{hw[0], hw[1]};
                *(uint2 *)(row + swz(r, 4 + (chunk >> 1)) + (chunk & 1) * 8) = uint2{lw[0], lw[1]};
            };
#pragma unroll
            for (int i = 0; i < AR; ++i) { const int r = rbase + RPL * i; split_store(sA + r * ROWB, r, ra[i]); }
#pragma unroll
            for (int i = 0; i < NB; ++i) { const int r = rbase + RPL * i; split_store(sB + r * ROWB, r, rb[i]); }
            return;
        }
#pragma unroll
        for (int i = 0; i < AR; ++i) {
            int r = rbase + RPL * i;
            *(u32x4 *)(sA + r * ROWB + swz(r, chunk)) = ra[i];
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            int r = rbase + RPL * i;
            // PW: channel c of a 32-channel group sits at row (c&4 ? 16 : 0) + 4*(c>>3) + (c&3), so that the
            // transposed MFMA below leaves 8 CONSECUTIVE channels of one pixel in a lane (fd_gemm_rows.hip)
            if constexpr (PWE) r = (r & ~31) | ((r & 4) << 2) | (((r & 31) >> 3) << 2) | (r & 3);
            *(u32x4 *)(sB + r * ROWB + swz(r, chunk)) = rb[i];
        }
    };

    // ---- compute roles
    const int wm = wave / WN, wn = wave % WN;
    const int fr = lane & 15, fg = lane >> 4;
    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    gload(0);
    lstore(0);
    __syncthreads();
    for (int kt = 0; kt < nkt; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nkt) gload(kt + 1);
        const unsigned char *sA = smem + buf * (BM + BN) * ROWB;
        const unsigned char *sB = sA + BM * ROWB;
#pragma unroll
        for (int ks = 0; ks < Frag<T>::KSTEPS; ++ks) {
            if constexpr (sizeof(T) == 2) {
                bf16x8 af[MT], bfr[NT];
#pragma unroll
                for (int i = 0; i < MT; ++i) {
                    int r = TMW * wm + 16 * i + fr;
                    af[i] = *(const bf16x8 *)(sA + r * ROWB + swz(r, ks * 4 + fg));
                }
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    int r = TNW * wn + 16 * j + fr;
                    bfr[j] = *(const bf16x8 *)(sB + r * ROWB + swz(r, ks * 4 + fg));
                }
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j)
                        acc[i][j] = PWE ? FD_MFMA16(bfr[j], af[i], acc[i][j], 0, 0, 0)
                                       : FD_MFMA16(af[i], bfr[j], acc[i][j], 0, 0, 0);
            } else if constexpr (SPL) {
                // split-bf16: hi fragment = k-chunk fg, lo fragment = k-chunk 4 + fg of the row; lane group fg owns
                // k = 8*fg .. 8*fg+7 in both, which is the K32 bf16 MFMA's own operand layout
                bf16x8 ah[MT], al[MT], bh[NT], bl[NT];
#pragma unroll
                for (int i = 0; i < MT; ++i) {
                    int r = TMW * wm + 16 * i + fr;
                    ah[i] = *(const bf16x8 *)(sA + r * ROWB + swz(r, fg));
                    al[i] = *(const bf16x8 *)(sA + r * ROWB + swz(r, 4 + fg));
                }
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    int r = TNW * wn + 16 * j + fr;
                    bh[j] = *(const bf16x8 *)(sB + r * ROWB + swz(r, fg));
                    bl[j] = *(const bf16x8 *)(sB + r * ROWB + swz(r, 4 + fg));
                }
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j) {
                        acc[i][j] = FD_MFMA16(al[i], bh[j], acc[i][j], 0, 0, 0);
                        acc[i][j] = FD_MFMA16(ah[i], bl[j], acc[i][j], 0, 0, 0);
                        acc[i][j] = FD_MFMA16(ah[i], bh[j], acc[i][j], 0, 0, 0);
                    }
            } else {
                // f32: lane group fg owns k = 8*fg .. 8*fg+7 of the 32-wide step; MFMA step e
                // contracts the k-set {8g + e}: any consistent A/B k-permutation is a valid sum.
                f32x4 a0[MT], a1[MT], b0[NT], b1[NT];
#pragma unroll
                for (int i = 0; i < MT; ++i) {
                    int r = TMW * wm + 16 * i + fr;
                    a0[i] = *(const f32x4 *)(sA + r * ROWB + swz(r, 2 * fg));
                    a1[i] = *(const f32x4 *)(sA + r * ROWB + swz(r, 2 * fg + 1));
                }
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    int r = TNW * wn + 16 * j + fr;
                    b0[j] = *(const f32x4 *)(sB + r * ROWB + swz(r, 2 * fg));
                    b1[j] = *(const f32x4 *)(sB + r * ROWB + swz(r, 2 * fg + 1));
                }
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int i = 0; i < MT; ++i)
#pragma unroll
                        for (int j = 0; j < NT; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[i][e], b0[j][e], acc[i][j], 0, 0, 0);
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int i = 0; i < MT; ++i)
#pragma unroll
                        for (int j = 0; j < NT; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[i][e], b1[j][e], acc[i][j], 0, 0, 0);
            }
        }
        if (kt + 1 < nkt) lstore(buf ^ 1);
        __syncthreads();
    }

    if constexpr (PWE) {
        // ---- pointwise epilogue straight from the accumulators (D^T: lane = pixel fr, 4 channels per tile,
        // two permuted tiles = 8 consecutive channels): no LDS staging, no barrier.  The dispatcher only
        // selects PW when Cout, strides and offsets are multiples of 8 and no GroupNorm sums are wanted.
        const int64_t obase_pw = 0;
        (void)obase_pw;
#pragma unroll
        for (int jp = 0; jp < NT / 2; ++jp) {
            const int n0 = nt * BN + TNW * wn + 32 * jp + 8 * fg;
            if (n0 >= p.Cout) continue;
            float bias8[8], ev0[8], ev1[8], gm = 0.f, gr = 0.f;
            const int cl = TNW * wn + 32 * jp + 8 * fg;          // channel inside the tile
            load8(&s_ep[0][cl], bias8);
            if (p.epilogue == FD_EPI_GATE_RES) load8(&s_ep[1][cl], ev0);
            else if (p.epilogue == FD_EPI_GNSILU_ADD) {
                load8(&s_ep[1][cl], ev0);
                load8(&s_ep[2][cl], ev1);
                gm = s_ep[3][cl];                               // channels-per-group is a multiple of 8 (dispatcher)
                gr = s_ep[4][cl];
            }
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const int m = mt * BM + TMW * wm + 16 * i + fr;
                if (m >= OHW) continue;
                const int64_t pix = (int64_t)b * OHW + m;
                float val[8];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    val[e] = acc[i][2 * jp][e] + bias8[e];
                    val[4 + e] = acc[i][2 * jp + 1][e] + bias8[4 + e];
                }
                if (p.epilogue == FD_EPI_SILU_SPLIT) {
                    if (n0 >= p.epi_split) fd_silu8(val);
                } else if (p.epilogue == FD_EPI_RELU) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) val[e] = fmaxf(val[e], 0.f);
                } else if (p.epilogue == FD_EPI_GATE_RES || p.epilogue == FD_EPI_RES_RELU) {
                    float rs[8];
                    load8((const T *)p.res + pix * p.ld_res + p.off_res + n0, rs);
                    if (p.epilogue == FD_EPI_GATE_RES) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) val[e] = __builtin_fmaf(ev0[e], val[e], rs[e]);   // (explicitly ONE rounding: see fd_pwgemm.hip)
                    } else {
#pragma unroll
                        for (int e = 0; e < 8; ++e) val[e] = fmaxf(val[e] + rs[e], 0.f);
                    }
                } else if (p.epilogue == FD_EPI_GNSILU_ADD) {
                    float hv[8];
                    load8((const T *)p.h + pix * p.Cout + n0, hv);
                    fd_gn_silu_add8(val, hv, gm, gr, ev0, ev1);
                }
                if (p.out_f32) store8((float *)p.out + pix * p.ldo + p.offo + n0, val);
                else store8((T *)p.out + pix * p.ldo + p.offo + n0, val);
            }
        }
        return;
    }
    // ---- epilogue on 8-channel vectors, one wave row (TMW tile rows) per pass through LDS
    constexpr int VPR = BN / 8;          // vectors per row
    constexpr int RPP = NTHR / VPR;      // rows per pass step
    static_assert(VPR <= 64 && (VPR & (VPR - 1)) == 0, "vector column groups must tile a wave");
    const int v = tid % VPR, r0 = tid / VPR;
    const int n0 = nt * BN + v * 8;
    const bool vec_ok = (p.Cout % 8 == 0) && (p.ldo % 8 == 0) && (p.offo % 8 == 0) &&
                        (p.res == nullptr || (p.ld_res % 8 == 0 && p.off_res % 8 == 0));
    float bias[8], gate[8], gam[8], bet[8], gmean[8], grstd[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        int n = n0 + e;
        bool ok = n < p.Cout;
        bias[e] = (p.bias && ok) ? p.bias[n] : 0.f;
        gate[e] = (p.epilogue == FD_EPI_GATE_RES && ok) ? p.gate[(int64_t)b * p.gate_ld + n] : 0.f;
        if (p.epilogue == FD_EPI_GNSILU_ADD && ok) {
            int g = n / (p.Cout / p.gn_groups);
            gam[e] = p.gn_gamma[n];
            bet[e] = p.gn_beta[n];
            gmean[e] = p.gn_mean_rstd[((int64_t)b * p.gn_groups + g) * 2];
            grstd[e] = p.gn_mean_rstd[((int64_t)b * p.gn_groups + g) * 2 + 1];
        } else {
            gam[e] = bet[e] = gmean[e] = grstd[e] = 0.f;
        }
    }
    float *sC = (float *)smem;
    float ssum[8], ssq[8];
    void *outp = p.out;
    const int64_t obase = (int64_t)dir * p.out_dir_stride;
    const int bands = (OHW + 63) / 64;       // 64-row bands per image = stats_partial tiles
    for (int h = 0; h < WM; ++h) {
        if (h > 0) __syncthreads();
        if (wm == h) {
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        int r = 16 * i + fg * 4 + e;
                        int cc = TNW * wn + 16 * j + fr;
                        sC[r * BN + cc] = acc[i][j][e];
                    }
        }
        __syncthreads();
        if ((h * TMW) % 64 == 0) {       // first pass of a 64-row band
#pragma unroll
            for (int e = 0; e < 8; ++e) ssum[e] = ssq[e] = 0.f;
        }
        for (int r = r0; r < TMW; r += RPP) {
            const int m = mt * BM + h * TMW + r;
            if (m >= OHW || n0 >= p.Cout) continue;
            const int64_t pix = (int64_t)b * OHW + m;
            float val[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) val[e] = sC[r * BN + v * 8 + e] + bias[e];
            if (p.epilogue == FD_EPI_SILU_SPLIT) {
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    if (n0 + e >= p.epi_split) val[e] = fd_silu(val[e]);
            } else if (p.epilogue == FD_EPI_RELU) {
#pragma unroll
                for (int e = 0; e < 8; ++e) val[e] = fmaxf(val[e], 0.f);
            } else if (p.epilogue == FD_EPI_GATE_RES || p.epilogue == FD_EPI_RES_RELU) {
                float rs[8];
                const T *rp = (const T *)p.res + pix * p.ld_res + p.off_res + n0;
                if (vec_ok) load8(rp, rs);
                else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) rs[e] = (n0 + e < p.Cout) ? ld1(rp + e) : 0.f;
                }
                if (p.epilogue == FD_EPI_GATE_RES) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) val[e] = __builtin_fmaf(gate[e], val[e], rs[e]);
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) val[e] = fmaxf(val[e] + rs[e], 0.f);
                }
            } else if (p.epilogue == FD_EPI_GNSILU_ADD) {
                float hv[8];
                const T *hp = (const T *)p.h + pix * p.Cout + n0;
                if (vec_ok) load8(hp, hv);
                else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) hv[e] = (n0 + e < p.Cout) ? ld1(hp + e) : 0.f;
                }
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    val[e] += fd_silu((hv[e] - gmean[e]) * grstd[e] * gam[e] + bet[e]);
            }
            if (p.stats_partial) {
#pragma unroll
                for (int e = 0; e < 8; ++e) { ssum[e] += val[e]; ssq[e] += val[e] * val[e]; }
            }
            if (p.out_f32) {
                float *op = (float *)outp + obase + pix * p.ldo + p.offo + n0;
                if (vec_ok) store8(op, val);
                else {
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        if (n0 + e < p.Cout) op[e] = val[e];
                }
            } else {
                T *op = (T *)outp + obase + pix * p.ldo + p.offo + n0;
                if (vec_ok) store8(op, val);
                else {
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        if (n0 + e < p.Cout) st1(op + e, val[e]);
                }
            }
        }
        if (p.stats_partial && ((h + 1) * TMW) % 64 == 0) {      // last pass of the band
            // lanes with equal (tid % VPR) hold the same 8 columns: reduce inside the wave, then
            // across the waves through LDS (fixed order -> deterministic).
#pragma unroll
            for (int e = 0; e < 8; ++e) {
#pragma unroll
                for (int o = VPR; o < 64; o <<= 1) {
                    ssum[e] += __shfl_xor(ssum[e], o, 64);
                    ssq[e] += __shfl_xor(ssq[e], o, 64);
                }
            }
            if (lane < VPR) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    s_stat[wave][lane * 8 + e][0] = ssum[e];
                    s_stat[wave][lane * 8 + e][1] = ssq[e];
                }
            }
            __syncthreads();
            const int band = (mt * BM + h * TMW) / 64;      // (TMW = 32: both passes belong to band mt)
            if (tid < BN && band < bands) {
                int n = nt * BN + tid;
                if (n < p.Cout) {
                    float s = 0.f, q = 0.f;
#pragma unroll
                    for (int w = 0; w < NWAVE; ++w) { s += s_stat[w][tid][0]; q += s_stat[w][tid][1]; }
                    float *sp = p.stats_partial + (((int64_t)b * bands + band) * p.Cout + n) * 2;
                    sp[0] = s;
                    sp[1] = q;
                }
            }
        }
    }
}

__global__ __launch_bounds__(1024) void gn_finalize_kernel(const float *__restrict__ part, int mtiles, int C, int groups,
                                                        double inv_cnt, float eps, float *__restrict__ mean_rstd) {
    // one workgroup of 1024 threads per (batch, group); float2 loads, fixed reduction order
    const int b = blockIdx.x / groups, g = blockIdx.x % groups;
    const int cpg = C / groups;
    double s = 0.0, q = 0.0;
    const int total = mtiles * cpg;
    for (int i = threadIdx.x; i < total; i += 1024) {
        int t = i / cpg, c = g * cpg + i % cpg;
        const float2 pp = *(const float2 *)(part + (((int64_t)b * mtiles + t) * C + c) * 2);
        s += pp.x;
        q += pp.y;
    }
    __shared__ double sh[2][1024];
    sh[0][threadIdx.x] = s;
    sh[1][threadIdx.x] = q;
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) {
            sh[0][threadIdx.x] += sh[0][threadIdx.x + o];
            sh[1][threadIdx.x] += sh[1][threadIdx.x + o];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        double mean = sh[0][0] * inv_cnt;
        double var = sh[1][0] * inv_cnt - mean * mean;
        if (var < 0) var = 0;
        mean_rstd[blockIdx.x * 2] = (float)mean;
        mean_rstd[blockIdx.x * 2 + 1] = (float)(1.0 / sqrt(var + (double)eps));
    }
}

template <typename T>
__global__ void gn_silu_apply_kernel(const T *__restrict__ h, const float *__restrict__ mean_rstd,
                                     const float *__restrict__ gamma, const float *__restrict__ beta,
                                     const T *__restrict__ res, T *__restrict__ out, int64_t hw, int C,
                                     int groups, int64_t nvec_per_img) {
    // A thread's 8-channel vector index advances by gridDim.x*blockDim.x per iteration; both are multiples
    // of C/8 on the host side, so the channel vector (and with it gamma / beta / the group statistics) is
    // loop-invariant: scale = rstd*gamma, shift = beta - mean*rstd*gamma are computed ONCE per thread.
    // (Per element per iteration -- an integer division and four scalar loads each -- this kernel ran
    // VALU-bound: PMC VALU busy 82 % at 4.2 TB/s.)
    const int b = blockIdx.y;
    const int cpg = C / groups;
    const int vpr = C / 8;
    const int64_t i0 = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    const int c0 = (int)(i0 % vpr) * 8;
    float sc[8], sh[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int c = c0 + e, g = c / cpg;
        const float mean = mean_rstd[((int64_t)b * groups + g) * 2];
        const float rstd = mean_rstd[((int64_t)b * groups + g) * 2 + 1];
        sc[e] = rstd * gamma[c];
        sh[e] = beta[c] - mean * sc[e];
    }
    for (int64_t i = i0; i < nvec_per_img; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t off = (int64_t)b * hw * C + i * 8;
        float hv[8], rv[8], o[8];
        load8(h + off, hv);
        if (res) load8(res + off, rv);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            f32x2 y = fd_silu2(f32x2{hv[2 * j], hv[2 * j + 1]} * f32x2{sc[2 * j], sc[2 * j + 1]} + f32x2{sh[2 * j], sh[2 * j + 1]});
            if (res) y = y + f32x2{rv[2 * j], rv[2 * j + 1]};
            o[2 * j] = y.x;
            o[2 * j + 1] = y.y;
        }
        store8(out + off, o);
    }
}

}  // namespace

// GroupNorm partial-sum bands per image (64 output pixels each, whatever tile the conv runs with)
extern "C" int fd_conv_mtiles(int OH, int OW) { return cdiv((int64_t)OH * OW, 64); }

int fd_gemm_rows_launch(const fd_conv_params &p, hipStream_t s);
int fd_conv3x3_ok(const fd_conv_params &p);
int fd_conv3x3_fp8_ok(const fd_conv_params &p);
int fd_conv3x3_launch(const fd_conv_params &p, hipStream_t s);
int fd_conv3x3_up2x_ok(const fd_conv_params &p);
int fd_conv3x3_split_ok(const fd_conv_params &p);
int fd_conv3x3_up2x_split_ok(const fd_conv_params &p);
int fd_conv3x3_rw_ok(const fd_conv_params &p);
int fd_conv3x3_rw_launch(const fd_conv_params &p, hipStream_t s);
int fd_pwgemm_ok(const fd_conv_params &p);
int fd_pwgemm_launch(const fd_conv_params &p, hipStream_t s);

// Which kernel fd_conv2d dispatches `p` to: 10 streaming row-GEMM (17: its fp32-storage split-bf16 form, fd_gemm_rows32.hip), 11 halo-tiled 3x3 (12: its fp8 form; 14: an up-sampling 3x3
// as four 2x2 convolutions on the source grid, weight_up2x; 15: its split-bf16 form on fp32 storage; 16: both at once), 13 the 64 -> 64 3x3 with
// the weights resident in registers (fd_conv3x3_rw.hip), else the
// implicit-GEMM tile variant 0 <128,128>, 1 <128,64>, 2 <64,128>, 3 <64,64>, 4 <128,256>, 5 <256,256>, 6 <128,32> (BM, BN);
// 7: the persistent 256x256 pointwise GEMM with deferred stores (fd_pwgemm.hip) where 5 would run and it applies.
extern "C" int fd_conv_fp8_ok(const fd_conv_params *pp) { return pp && !fd_conv_prologue_ok(pp) && fd_conv3x3_fp8_ok(*pp); }

extern "C" int fd_conv_kernel_id(const fd_conv_params *pp) {
    if (fd_conv_prologue_ok(pp)) return pp->dtype == FD_F32 ? 17 : 10;
    if (fd_conv3x3_rw_ok(*pp)) return 13;
    if (fd_conv3x3_ok(*pp)) return fd_conv3x3_fp8_ok(*pp) ? 12 : (fd_conv3x3_up2x_ok(*pp) ? 14 : 11);
    if (fd_conv3x3_up2x_split_ok(*pp)) return 16;
    if (fd_conv3x3_split_ok(*pp)) return 15;
    const bool wide = pp->Cout > 64, tall = conv_bm((int64_t)pp->OH * pp->OW) == 128;
    const int force = fd_dev(FD_DEV_CONV_KID);   // development: tile experiments
    if (force >= 0 && force <= 6 && !pp->stats_partial && pp->KH == 1 && pp->Cout >= 256) return force;
    // 8-wave 128x256 tile (id 4) for the dense layers of the 64x64 / 128x128 levels: it halves the operand
    // traffic from beyond L2 (+15..30 % at batch 8) but launches 4x fewer workgroups, so it is chosen only
    // when the batch fills the chip.  This choice may depend on the batch size without breaking batch
    // invariance: every output element accumulates its K tiles in the same order whatever the tile shape
    // (bitwise identical results); only the GroupNorm partial sums depend on the tiling, so convolutions
    // that emit them keep the batch-independent configuration.
    // (measured at batch 8: wins 12-25 % for Cout 256..512; loses 5-10 % for Cout >= 1024, where the small
    //  tile already launches plenty of workgroups per A tile)
    // 256x256 (id 5, 8 waves of 64x128): a CU sustains only ~20 B/clk of L2-hit loads, so the MFMA rate of
    // these K-streaming tiles is set by FLOP per loaded byte = BM*BN/(BM+BN): 600-680 TFLOP/s against 560
    // for every smaller tile (measured, batch 8).  Round 3: also at K = 256 (the in_proj / qkv of the 256-channel blocks at
    // 128x128: 176 -> 140 us and 120 -> 99 us at batch 8 against the 64x128 tile, `FD_CONV_KID` experiments).
    if (!tall && !pp->stats_partial && pp->Cout >= 256 && pp->KH * pp->KW * (pp->c0 + pp->c1) >= 256 &&
        (int64_t)pp->B * pp->ndir * cdiv((int64_t)pp->OH * pp->OW, 256) * cdiv(pp->Cout, 256) >= 192)
        return fd_pwgemm_ok(*pp) ? 7 : 5;
    if (!tall && pp->Cout >= 256 && pp->Cout <= 512 && !pp->stats_partial && pp->KH * pp->KW * (pp->c0 + pp->c1) >= 256) {
        const int64_t wgs = (int64_t)pp->B * pp->ndir * cdiv((int64_t)pp->OH * pp->OW, 128) * cdiv(pp->Cout, 256);
        if (wgs >= 192 || fd_dev(FD_DEV_CONV_BIG_TILE)) return 4;
    }
    // very narrow outputs (x_proj: Cout = dt_rank + 2 d_state = 12..40 at the high-resolution levels): a
    // 32-column tile halves the wasted B-side work of the 64-column one and fits more workgroups per CU
    if (tall && pp->Cout <= 32) return 6;
    return (tall ? 0 : 2) + (wide ? 0 : 1);
}

extern "C" int fd_conv2d(const fd_conv_params *pp, void *stream) {
    const fd_conv_params &p = *pp;
    const int CH = p.dtype == FD_BF16 ? 8 : 4;
    FD_REQUIRE(p.dtype == FD_F32 || p.dtype == FD_BF16, "fd_conv2d: bad dtype %d", p.dtype);
    FD_REQUIRE(p.in0 && p.weight && p.out, "fd_conv2d: null pointer");
    FD_REQUIRE(p.c0 > 0 && p.c0 % CH == 0 && p.ld0 % CH == 0 && p.off0 % CH == 0,
               "fd_conv2d: source 0 channels/stride/offset (%d,%d,%d) must be multiples of %d", p.c0, p.ld0, p.off0, CH);
    FD_REQUIRE(p.c1 == 0 || (p.in1 && p.c1 % CH == 0 && p.ld1 % CH == 0 && p.off1 % CH == 0),
               "fd_conv2d: source 1 channels/stride/offset must be multiples of %d", CH);
    FD_REQUIRE(p.B > 0 && p.H > 0 && p.W > 0 && p.OH > 0 && p.OW > 0 && p.Cout > 0, "fd_conv2d: bad sizes");
    FD_REQUIRE(p.KH > 0 && p.KW > 0 && p.stride > 0, "fd_conv2d: bad kernel/stride");
    FD_REQUIRE(p.ndir == 1 || p.ndir == 4, "fd_conv2d: ndir must be 1 or 4");
    FD_REQUIRE(p.epilogue >= FD_EPI_NONE && p.epilogue <= FD_EPI_GNSILU_ADD_FINAL, "fd_conv2d: bad epilogue %d", p.epilogue);
    FD_REQUIRE(p.epilogue != FD_EPI_GNSILU_ADD_FINAL || fd_conv_prologue_ok(pp),
               "fd_conv2d: GNSILU_ADD_FINAL runs on the streaming row-GEMM only (bf16 or fp32s, 1x1, >= 16384 px, fin_* set)");
    if (p.epilogue == FD_EPI_GATE_RES) FD_REQUIRE(p.res && p.gate, "fd_conv2d: GATE_RES needs res and gate");
    if (p.epilogue == FD_EPI_RES_RELU) FD_REQUIRE(p.res, "fd_conv2d: RES_RELU needs res");
    if (p.epilogue == FD_EPI_GNSILU_ADD || p.epilogue == FD_EPI_GNSILU_ADD_FINAL)
        FD_REQUIRE(p.h && p.gn_mean_rstd && p.gn_gamma && p.gn_beta && p.gn_groups > 0 && p.Cout % p.gn_groups == 0,
                   "fd_conv2d: GNSILU_ADD needs h, statistics, affine and groups | Cout");
    if (fd_conv_prologue_ok(pp)) {
        FD_REQUIRE(fd_gemm_rows_launch(p, (hipStream_t)stream) == 0, "fd_conv2d: row-GEMM dispatch failed");
        FD_LAUNCH_OK("fd_conv2d(row-gemm)");
        return FD_OK;
    }
    if (fd_conv3x3_rw_ok(p)) {
        fd_conv3x3_rw_launch(p, (hipStream_t)stream);
        FD_LAUNCH_OK("fd_conv2d(3x3, weights in registers)");
        return FD_OK;
    }
    if (fd_conv3x3_ok(p) || fd_conv3x3_split_ok(p) || fd_conv3x3_up2x_split_ok(p)) {
        fd_conv3x3_launch(p, (hipStream_t)stream);
        FD_LAUNCH_OK("fd_conv2d(3x3 halo)");
        return FD_OK;
    }
    FD_REQUIRE(p.prologue == FD_PRO_NONE, "fd_conv2d: fused LN prologue requested for a conv the row-GEMM path "
                                          "cannot run (check fd_conv_prologue_ok first)");
    FD_REQUIRE((int64_t)p.H * p.W * (p.ld0 > p.ld1 ? p.ld0 : p.ld1) < (1ll << 31),
               "fd_conv2d: one image of a source must hold < 2^31 elements");
    const int kid = fd_conv_kernel_id(pp);
    if (kid == 7) {
        fd_pwgemm_launch(p, (hipStream_t)stream);
        FD_LAUNCH_OK("fd_conv2d(persistent pointwise GEMM)");
        return FD_OK;
    }
    const int BMs[7] = {128, 128, 64, 64, 128, 256, 128}, BNs[7] = {128, 64, 128, 64, 256, 256, 32};
    dim3 grid(cdiv((int64_t)p.OH * p.OW, BMs[kid]), cdiv(p.Cout, BNs[kid]), p.B * p.ndir), block(kid == 4 || kid == 5 ? 512 : 256);
    hipStream_t s = (hipStream_t)stream;
    // pointwise fast path (bf16): same tiles, same K order, same results -- only the address math differs
    const bool pw = p.dtype == FD_BF16 && p.KH == 1 && p.KW == 1 && p.stride == 1 && p.pad_h == 0 && p.pad_w == 0 &&
                    !p.upsample && p.ndir == 1 && p.OH == p.H && p.OW == p.W && (p.c0 + p.c1) % 64 == 0 &&
                    p.c0 % 64 == 0 && (int64_t)p.Cout * (p.c0 + p.c1) < (1ll << 31) &&
                    // its epilogue stores 8-channel vectors straight from the accumulators
                    !p.stats_partial && p.Cout % 8 == 0 && p.ldo % 8 == 0 && p.offo % 8 == 0 &&
                    (!p.res || (p.ld_res % 8 == 0 && p.off_res % 8 == 0)) && (!p.bias || ((uintptr_t)p.bias & 15) == 0) &&
                    (p.epilogue != FD_EPI_SILU_SPLIT || p.epi_split % 8 == 0) &&
                    (p.epilogue != FD_EPI_GNSILU_ADD || (p.Cout / p.gn_groups) % 8 == 0);
#define FD_CONV_LAUNCH(T_, BM_, BN_, WM_, WN_, PW_) \
    hipLaunchKernelGGL((conv_igemm_kernel<T_, BM_, BN_, WM_, WN_, PW_>), grid, block, 0, s, p)
#define FD_CONV_DISPATCH(T_, PW_)                                     \
    switch (kid) {                                                    \
    case 0: FD_CONV_LAUNCH(T_, 128, 128, 2, 2, PW_); break;           \
    case 1: FD_CONV_LAUNCH(T_, 128, 64, 2, 2, PW_); break;            \
    case 2: FD_CONV_LAUNCH(T_, 64, 128, 2, 2, PW_); break;            \
    case 3: FD_CONV_LAUNCH(T_, 64, 64, 2, 2, PW_); break;             \
    case 5: FD_CONV_LAUNCH(T_, 256, 256, 4, 2, PW_); break;           \
    case 6: FD_CONV_LAUNCH(T_, 128, 32, 2, 2, PW_); break;            \
    default: FD_CONV_LAUNCH(T_, 128, 256, 2, 4, PW_); break;          \
    }
    if (p.dtype == FD_BF16) {
        if (pw) { FD_CONV_DISPATCH(bf16, true) } else { FD_CONV_DISPATCH(bf16, false) }
    } else if (p.f32_split) {
#define FD_CONV_LAUNCH_S(BM_, BN_, WM_, WN_) \
    hipLaunchKernelGGL((conv_igemm_kernel<float, BM_, BN_, WM_, WN_, false, true>), grid, block, 0, s, p)
        switch (kid) {
        case 0: FD_CONV_LAUNCH_S(128, 128, 2, 2); break;
        case 1: FD_CONV_LAUNCH_S(128, 64, 2, 2); break;
        case 2: FD_CONV_LAUNCH_S(64, 128, 2, 2); break;
        case 3: FD_CONV_LAUNCH_S(64, 64, 2, 2); break;
        case 5: FD_CONV_LAUNCH_S(256, 256, 4, 2); break;
        case 6: FD_CONV_LAUNCH_S(128, 32, 2, 2); break;
        default: FD_CONV_LAUNCH_S(128, 256, 2, 4); break;
        }
#undef FD_CONV_LAUNCH_S
    } else { FD_CONV_DISPATCH(float, false) }
#undef FD_CONV_DISPATCH
#undef FD_CONV_LAUNCH
    FD_LAUNCH_OK("fd_conv2d");
    return FD_OK;
}

extern "C" int fd_gn_finalize(const float *stats_partial, int B, int mtiles, int C, int groups,
                              int64_t hw, float eps, float *mean_rstd, void *stream) {
    FD_REQUIRE(stats_partial && mean_rstd && groups > 0 && C % groups == 0, "fd_gn_finalize: bad args");
    double inv = 1.0 / ((double)hw * (C / groups));
    hipLaunchKernelGGL(gn_finalize_kernel, dim3(B * groups), dim3(1024), 0, (hipStream_t)stream,
                       stats_partial, mtiles, C, groups, inv, eps, mean_rstd);
    FD_LAUNCH_OK("fd_gn_finalize");
    return FD_OK;
}

extern "C" int fd_gn_silu_apply(int dtype, const void *h, const float *mean_rstd, const float *gamma,
                                const float *beta, const void *res, void *out, int B, int64_t hw,
                                int C, int groups, void *stream) {
    FD_REQUIRE(C % 8 == 0 && C % groups == 0, "fd_gn_silu_apply: C=%d must be a multiple of 8 and of groups", C);
    int64_t nvec = hw * C / 8;
    // grid stride (gridDim.x * 256 threads) must be a multiple of the vectors per pixel row (C/8) so that a
    // thread keeps its channel vector: round the block count to a multiple of C/8 / gcd(C/8, 256)
    const int vpr = C / 8;
    int gcd = vpr, t = 256;
    while (t) { int r = gcd % t; gcd = t; t = r; }
    const int step = vpr / gcd;
    int64_t nb = (nvec + 255) / 256;
    if (nb > 4096) nb = 4096;
    nb = (nb + step - 1) / step * step;
    dim3 grid((unsigned)nb, B), block(256);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == FD_BF16)
        hipLaunchKernelGGL(gn_silu_apply_kernel<bf16>, grid, block, 0, s, (const bf16 *)h, mean_rstd, gamma, beta,
                           (const bf16 *)res, (bf16 *)out, hw, C, groups, nvec);
    else
        hipLaunchKernelGGL(gn_silu_apply_kernel<float>, grid, block, 0, s, (const float *)h, mean_rstd, gamma, beta,
                           (const float *)res, (float *)out, hw, C, groups, nvec);
    FD_LAUNCH_OK("fd_gn_silu_apply");
    return FD_OK;
}
