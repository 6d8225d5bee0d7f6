// fd_gemm_rows.hip -- streaming row-GEMM for the 1x1 convolutions / linears of the high-resolution
// levels (bf16): out[m, :] = epi( W . pro(x[m, :]) ), M = B*H*W huge, K = Cin <= 256, weights tiny.
//
// These layers are HBM-bound (arithmetic intensity < 100 FLOP/B), so the kernel is organised
// around the byte stream, not around a C tile:
//   * the whole weight matrix lives in LDS for the lifetime of a (persistent) workgroup;
//   * there is NO LDS staging of activations and NO barrier in the pixel loop: the MFMA is issued
//     transposed (D^T = W . X^T), whose B operand is "8 consecutive channels of one pixel per
//     lane" -- exactly a 16-byte global load of the NHWC row -- and whose D layout gives every
//     lane 4 consecutive output channels of one pixel; two 16-row weight tiles are row-permuted
//     so that a lane ends up with 8 CONSECUTIVE channels: one 16-byte store, with the residual /
//     GroupNorm operand fetched by the matching 16-byte load;
//   * a pixel's K channels sit in 4 lanes (l, l^16, l^32, l^48): LayerNorm statistics for the
//     fused prologues are two xor-shuffles, so LN+adaLN-modulate (in_proj, qkv) and
//     out_norm*z+local (out_proj) never round-trip through HBM.
#include "fd_common.h"

namespace {

__device__ __forceinline__ void unpack8(const bf16x8 &v, float f[8]) {
#pragma unroll
    for (int i = 0; i < 8; ++i) f[i] = (float)v[i];
}
__device__ __forceinline__ bf16x8 pack8(const float f[8]) {
    bf16x8 v;
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = fd_cvt_h(f[i]);
    return v;
}

// LDS image of W: row n at n*RS bytes, 16-byte chunks XOR-swizzled so that the 16 rows a wave
// reads per fragment ({8a+b} or {8a+4+b}, a,b in 0..3) fall on distinct bank groups.
__device__ __forceinline__ int w_off(int row, int chunk, int RS) {
    const int sw = RS == 128 ? (((row >> 1) & 1) | (((row >> 3) & 3) << 1))
                             : ((row & 3) | (((row >> 3) & 3) << 2));
    return row * RS + ((chunk ^ sw) << 4);
}

// S = 16-pixel subtiles per wave iteration: 2 for K <= 64 (each weight fragment read from LDS feeds
// two MFMAs), 1 for wider K where two subtiles' operands would halve the occupancy.
// NT = threads per workgroup: 256 when >= 3 workgroups fit a CU's LDS, 512 / 768 when the weight image is
// so large (Cout*RS up to 128 KiB) that only 2 / 1 fit -- the waves that hide each other's HBM latency
// then have to live in ONE workgroup sharing the weights (768 = 12 waves: 3 per SIMD, 170 VGPRs each).
template <int KS, int PRO, int EPI, int NT, int S = (KS <= 2 ? 2 : 1)>
__global__ __launch_bounds__(NT) void gemm_rows_kernel(const fd_conv_params p, int wtiles, int RS) {
    constexpr int NW = NT / 64;
    constexpr int K = 32 * KS;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int N = p.Cout;
    unsigned char *sW = smem;
    float *sV = (float *)(smem + (size_t)N * RS);          // prologue vectors: [3][K]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fg = lane >> 4;
    const int b = blockIdx.y;
    const int64_t hw = (int64_t)p.H * p.W;
    const bf16 *wg = (const bf16 *)p.weight + (int64_t)b * p.w_batch_stride;
    // ---- stage W (and the per-channel prologue vectors) once per workgroup
    for (int idx = tid; idx < N * (K / 8); idx += NT) {
        const int row = idx / (K / 8), ch = idx - row * (K / 8);
        *(u32x4 *)(sW + w_off(row, ch, RS)) = *(const u32x4 *)(wg + (int64_t)row * K + ch * 8);
    }
    if (PRO == 1) {          // G = gamma (1+scale), Bc = beta (1+scale) + shift
        for (int c = tid; c < K; c += NT) {
            const float g = p.ln_gamma ? p.ln_gamma[c] : 1.f, be = p.ln_beta ? p.ln_beta[c] : 0.f;
            const float sc = 1.f + p.ln_scale[(int64_t)b * p.ln_ld + c], sh = p.ln_shift[(int64_t)b * p.ln_ld + c];
            sV[c] = g * sc;
            sV[K + c] = be * sc + sh;
        }
    } else if (PRO == 2) {   // gamma, beta, local
        for (int c = tid; c < K; c += NT) {
            sV[c] = p.ln_gamma[c];
            sV[K + c] = p.ln_beta[c];
            sV[2 * K + c] = p.ln_shift[(int64_t)b * p.ln_ld + c];
        }
    }
    __syncthreads();

    const bf16 *in0 = (const bf16 *)p.in0 + (int64_t)b * hw * p.ld0 + p.off0;
    const bf16 *in1 = p.in1 ? (const bf16 *)p.in1 + (int64_t)b * hw * p.ld1 + p.off1 : nullptr;
    const bf16 *zin = PRO == 2 ? (const bf16 *)p.ln_z + (int64_t)b * hw * p.ln_ldz + p.ln_offz : nullptr;
    bf16 *outp = (bf16 *)p.out + (int64_t)b * hw * p.ldo + p.offo;
    const bf16 *resp = p.res ? (const bf16 *)p.res + (int64_t)b * hw * p.ld_res + p.off_res : nullptr;
    const bf16 *hp = p.h ? (const bf16 *)p.h + (int64_t)b * hw * N : nullptr;
    const int cpg = p.gn_groups > 0 ? N / p.gn_groups : 1;

    auto load_tile = [&](int wt, bf16x8 (&xb)[S][KS]) {
#pragma unroll
        for (int s = 0; s < S; ++s) {
            // rows beyond the image (ragged last tile) read the last row instead: their results are never stored, and
            // unconditional loads keep the tile body one basic block (the scheduling fences below need that)
            const int64_t m = min((int64_t)wt * (16 * S) + 16 * s + fr, hw - 1);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const int c = ks * 32 + fg * 8;
                const bf16 *src = c < p.c0 ? in0 + m * p.ld0 + c : in1 + m * p.ld1 + (c - p.c0);
                xb[s][ks] = *(const bf16x8 *)src;
            }
        }
    };

    // No software prefetch: the kernel keeps its register footprint small enough for 3-4 waves
    // per SIMD and lets the other waves' MFMA/store phases cover this wave's load latency.  (Round 4: the next tile's
    // rows requested one tile ahead in the no-prologue K >= 96 variants, 72-118 VGPRs: 12.65 -> 12.68 ms per batch-8
    // forward, alternated -- not kept.)
    const int wstride = gridDim.x * NW;
    for (int wt = blockIdx.x * NW + wave; wt < wtiles; wt += wstride) {
        bf16x8 xb[S][KS];
        load_tile(wt, xb);
        if (PRO != 0) {
            // the prologue vectors of a lane (its 8 channels of every K32 step) do not depend on the tile: hoisted out of
            // the pixel loop they were 2-3 x KS x 8 permanently live registers (out_proj: 182 VGPRs, 2 waves per SIMD;
            // K = 256: 256 + 116 spilled to AGPRs, ONE wave per SIMD) on kernels that live on waves hiding each other's
            // HBM latency.  An opaque copy of fg keeps the (cheap, broadcast-free) LDS reads inside the loop.
            int fgo = fg;
            asm volatile("" : "+v"(fgo));
#pragma unroll
            for (int s = 0; s < S; ++s) {
                // three cheap unpack passes instead of one fp32 copy of the row kept live across the
                // shuffles: the fp32 temporaries are 8 values at a time (register diet -> occupancy)
                float sum = 0.f;
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    float f[8];
                    unpack8(xb[s][ks], f);
#pragma unroll
                    for (int e = 0; e < 8; ++e) sum += f[e];
                }
                sum += __shfl_xor(sum, 16, 64);
                sum += __shfl_xor(sum, 32, 64);
                const float mean = sum * (1.f / K);
                float q = 0.f;
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    float f[8];
                    unpack8(xb[s][ks], f);
#pragma unroll
                    for (int e = 0; e < 8; ++e) { const float d = f[e] - mean; q += d * d; }
                }
                q += __shfl_xor(q, 16, 64);
                q += __shfl_xor(q, 32, 64);
                const float rstd = rsqrtf(q * (1.f / K) + p.ln_eps);
                const int64_t m = min((int64_t)wt * (16 * S) + 16 * s + fr, hw - 1);
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const int c = ks * 32 + fgo * 8;
                    float f[8], g8[8], b8[8];
                    unpack8(xb[s][ks], f);
                    load8(sV + c, g8);
                    load8(sV + K + c, b8);
                    if (PRO == 1) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) f[e] = (f[e] - mean) * rstd * g8[e] + b8[e];
                    } else {
                        float z8[8], l8[8];
                        const bf16x8 z = *(const bf16x8 *)(zin + m * p.ln_ldz + c);
                        unpack8(z, z8);
                        load8(sV + 2 * K + c, l8);
#pragma unroll
                        for (int e = 0; e < 8; ++e) f[e] = ((f[e] - mean) * rstd * g8[e] + b8[e]) * z8[e] + l8[e];
                    }
                    xb[s][ks] = pack8(f);
                    // one K32 step at a time: left to itself the compiler issues the gamma / beta / local / z loads of
                    // ALL steps first and sinks the arithmetic to the MFMA loop: ~100 more live registers (2 instead
                    // of 4 waves per SIMD on an HBM-bound kernel).  The empty asm pins this step's result here.
                    if (PRO == 2) {
                        u32x4 pin = __builtin_bit_cast(u32x4, xb[s][ks]);
                        asm volatile("" : "+v"(pin));
                        xb[s][ks] = __builtin_bit_cast(bf16x8, pin);
                    }
                }
            }
        }
        const int64_t m0 = (int64_t)wt * (16 * S) + fr;
        float fdot[S];                       // GNSILU_ADD_FINAL: this lane's share of the final 1x1 (Cout -> 1)
#pragma unroll
        for (int s = 0; s < S; ++s) fdot[s] = 0.f;
#pragma unroll 1
        for (int ng = 0; ng < N / 32; ++ng) {
            f32x4 acc[2][S];
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int s = 0; s < S; ++s) acc[t][s] = (f32x4){0.f, 0.f, 0.f, 0.f};
            const int rowa = 32 * ng + 8 * (fr >> 2) + (fr & 3);
            // LN_GATE prologue (out_proj): the weight fragments, too, stay LDS reads per tile instead of N*K/256 hoisted
            // registers per lane -- this variant needs its registers for waves (HBM latency), not for a 16-64 KB matrix
            int fgw = fg;
            if (PRO == 2) asm volatile("" : "+v"(fgw));
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const bf16x8 wa = *(const bf16x8 *)(sW + w_off(rowa, ks * 4 + fgw, RS));
                const bf16x8 wb = *(const bf16x8 *)(sW + w_off(rowa + 4, ks * 4 + fgw, RS));
#pragma unroll
                for (int s = 0; s < S; ++s) {
                    acc[0][s] = FD_MFMA16(wa, xb[s][ks], acc[0][s], 0, 0, 0);
                    acc[1][s] = FD_MFMA16(wb, xb[s][ks], acc[1][s], 0, 0, 0);
                }
            }
            // lane (pixel fr of subtile s, group fg) holds channels n0 .. n0+7
            const int n0 = 32 * ng + 8 * fg;
            float bias[8], ev0[8], ev1[8], ev2[8], ev3[8];
            if (p.bias) load8(p.bias + n0, bias);
            else {
#pragma unroll
                for (int e = 0; e < 8; ++e) bias[e] = 0.f;
            }
            if (EPI == FD_EPI_GATE_RES) load8(p.gate + (int64_t)b * p.gate_ld + n0, ev0);
            if (EPI == FD_EPI_GNSILU_ADD || EPI == FD_EPI_GNSILU_ADD_FINAL) {
                load8(p.gn_gamma + n0, ev0);
                load8(p.gn_beta + n0, ev1);
                // channels-per-group is a multiple of 8 (checked on the host): one group per vector
                const int g = n0 / cpg;
                const float gm = p.gn_mean_rstd[((int64_t)b * p.gn_groups + g) * 2];
                const float gr = p.gn_mean_rstd[((int64_t)b * p.gn_groups + g) * 2 + 1];
#pragma unroll
                for (int e = 0; e < 8; ++e) { ev2[e] = gm; ev3[e] = gr; }
            }
#pragma unroll
            for (int s = 0; s < S; ++s) {
                const int64_t m = m0 + 16 * s;
                if (m >= hw) continue;
                float val[8];
#pragma unroll
                for (int e = 0; e < 4; ++e) { val[e] = acc[0][s][e] + bias[e]; val[4 + e] = acc[1][s][e] + bias[4 + e]; }
                if (EPI == FD_EPI_SILU_SPLIT) {
                    if (n0 >= p.epi_split) fd_silu8(val);
                } else if (EPI == FD_EPI_RELU) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) val[e] = fmaxf(val[e], 0.f);
                } else if (EPI == FD_EPI_GATE_RES || EPI == FD_EPI_RES_RELU) {
                    float rs[8];
                    load8(resp + m * p.ld_res + n0, rs);
                    if (EPI == FD_EPI_GATE_RES) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) val[e] = __builtin_fmaf(ev0[e], val[e], rs[e]);
                    } else {
#pragma unroll
                        for (int e = 0; e < 8; ++e) val[e] = fmaxf(val[e] + rs[e], 0.f);
                    }
                } else if (EPI == FD_EPI_GNSILU_ADD || EPI == FD_EPI_GNSILU_ADD_FINAL) {
                    float hv[8];
                    load8(hp + m * N + n0, hv);
                    fd_gn_silu_add8(val, hv, ev2[0], ev3[0], ev0, ev1);
                }
                if (EPI == FD_EPI_GNSILU_ADD_FINAL) {
                    // the block output is rounded to the storage type before final_conv reads it in the unfused
                    // sequence: same rounding point here
                    float fw[8];
                    load8(p.fin_w + n0, fw);
#pragma unroll
                    for (int e = 0; e < 8; ++e) fdot[s] += (float)fd_cvt_h(val[e]) * fw[e];
                } else {
                    store8(outp + m * p.ldo + n0, val);
                }
            }
        }
        if (EPI == FD_EPI_GNSILU_ADD_FINAL) {
            // a pixel's Cout channels sit in the 4 lanes fr, fr + 16, fr + 32, fr + 48: two xor-shuffles
#pragma unroll
            for (int s = 0; s < S; ++s) {
                float o = fdot[s];
                o += __shfl_xor(o, 16, 64);
                o += __shfl_xor(o, 32, 64);
                o += p.fin_b;
                const int64_t m = m0 + 16 * s;
                if (fg == 0 && m < hw) {
                    const int64_t j = (int64_t)b * hw + m;
                    p.fin_out[j] = o;
                    if (p.fin_mode == 1) {
                        const float pr = fd_clamp1(o);
                        p.fin_img[j] = p.fin_last ? fd_clamp1(p.fin_xin[j] - pr) : p.fin_img[j] - p.fin_alpha * pr;
                    }
                }
            }
        }
    }
}

int row_stride(int K) { return K <= 64 ? 128 : (K <= 128 ? 256 : 512); }

// ---------------------------------------------------------------------------------------------------------------
// out_proj with the z gate RECOMPUTED (FD_PRO_LN_GATE_ZRE, round 4).  The fused in_proj kernel (fd_pwdw.hip) used to
// write z = SiLU(W_z . LNmod(x)) -- d_inner channels per pixel, twice the block's own width -- only for this kernel to
// read it back as the gate of out_norm(y): 1.07 GB of the 6.2 GB a level-0 Mamba block moves at batch 8.  out_proj
// reads the block input x anyway (the residual of its GATE_RES epilogue), the z rows of in_proj are 16 KB, and the
// transposed MFMA issue of this file hands a lane EXACTLY the 8 consecutive z channels of its pixel that the LN_GATE
// prologue multiplies into its 8 y channels (K32 step ks of y <-> 32-row group ks of W_z).  So per 16-pixel tile:
//   x row (CX = K/2 channels) -> LayerNorm + adaLN modulate (one pass, packed; the arithmetic of fd_ln_mod_chunk)
//   -> bf16 -> K/32 x (K/64) x 2 MFMAs against W_z in LDS -> SiLU -> bf16 (the rounding point of the stored z)
//   -> LN_GATE prologue on y with z from registers -> out_proj MFMAs -> x + gate . (acc + bias) with x from the
//   registers the prologue loaded.  The matrix pipe of this HBM-bound kernel was 4 % busy; bytes per pixel fall from
//   (K + K + CX + CX) x 2 to (K + CX + CX) x 2.
__device__ __forceinline__ float zre_dot2(uint32_t a, uint32_t b, float c) {
    return fd_dot2(__builtin_bit_cast(bf16x2, a), __builtin_bit_cast(bf16x2, b), c);
}

template <int KS, int NT, bool PF>
__global__ __launch_bounds__(NT, NT == 256 ? 3 : 2) void gemm_rows_zre_kernel(const fd_conv_params p, int wtiles) {
    constexpr int NW = NT / 64, K = 32 * KS, KX = KS / 2, CX = 16 * KS;
    constexpr int RS = K <= 128 ? 256 : 512, RX = CX <= 64 ? 128 : 256;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char *sW = smem;                                  // out_proj  [CX rows][RS]
    unsigned char *sZ = smem + CX * RS;                        // W_z       [K rows][RX]
    float *sV = (float *)(sZ + K * RX);                        // out_norm gamma, beta, local: [3][K]
    float *sX = sV + 3 * K;                                    // norm1 G = gamma (1 + scale), Bc = beta (1 + scale) + shift: [2][CX]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fg = lane >> 4;
    const int b = blockIdx.y;
    const int64_t hw = (int64_t)p.H * p.W;
    {
        const bf16 *wg = (const bf16 *)p.weight, *wz = (const bf16 *)p.zre_w;
        for (int idx = tid; idx < CX * (K / 8); idx += NT) {
            const int row = idx / (K / 8), ch = idx - row * (K / 8);
            *(u32x4 *)(sW + w_off(row, ch, RS)) = *(const u32x4 *)(wg + (int64_t)row * K + ch * 8);
        }
        for (int idx = tid; idx < K * (CX / 8); idx += NT) {
            const int row = idx / (CX / 8), ch = idx - row * (CX / 8);
            *(u32x4 *)(sZ + w_off(row, ch, RX)) = *(const u32x4 *)(wz + (int64_t)row * CX + ch * 8);
        }
        for (int c = tid; c < K; c += NT) {
            sV[c] = p.ln_gamma[c];
            sV[K + c] = p.ln_beta[c];
            sV[2 * K + c] = p.ln_shift[(int64_t)b * p.ln_ld + c];
        }
        for (int c = tid; c < CX; c += NT) {
            const float g = p.zre_gamma ? p.zre_gamma[c] : 1.f, be = p.zre_beta ? p.zre_beta[c] : 0.f;
            const float sc = 1.f + p.zre_scale[(int64_t)b * p.zre_ld + c], sh = p.zre_shift[(int64_t)b * p.zre_ld + c];
            sX[c] = g * sc;
            sX[CX + c] = be * sc + sh;
        }
    }
    __syncthreads();
    const bf16 *yin = (const bf16 *)p.in0 + (int64_t)b * hw * p.ld0 + p.off0;
    const bf16 *xin = (const bf16 *)p.res + (int64_t)b * hw * p.ld_res + p.off_res;
    bf16 *outp = (bf16 *)p.out + (int64_t)b * hw * p.ldo + p.offo;
    const int rperm = 8 * (fr >> 2) + (fr & 3);       // weight rows permuted so that a lane ends up with 8 consecutive channels
    const int wstride = gridDim.x * NW;
    // K = 128: the next tile's rows are requested before this tile's arithmetic (a tile is ~550 VALU slots per lane here,
    // against ~250 in gemm_rows_kernel, which leaves the overlap to its 3-4 waves per SIMD); K = 256 has no registers for it
    u32x4 yn[PF ? KS : 1], xnx[PF ? KX : 1];
    auto fetch = [&](int wt, u32x4 *yd, u32x4 *xd) {
        const int64_t mm = min((int64_t)wt * 16 + fr, hw - 1);    // ragged last tile: rows beyond the image read the last row
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) yd[ks] = *(const u32x4 *)(yin + mm * p.ld0 + ks * 32 + fg * 8);
#pragma unroll
        for (int kx = 0; kx < KX; ++kx) xd[kx] = *(const u32x4 *)(xin + mm * p.ld_res + kx * 32 + fg * 8);
    };
    if (PF && blockIdx.x * NW + wave < wtiles) fetch(blockIdx.x * NW + wave, yn, xnx);
    for (int wt = blockIdx.x * NW + wave; wt < wtiles; wt += wstride) {
        const int64_t m = min((int64_t)wt * 16 + fr, hw - 1);
        u32x4 yr[KS], xr[KX];
        if (PF) {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) yr[ks] = yn[ks];
#pragma unroll
            for (int kx = 0; kx < KX; ++kx) xr[kx] = xnx[kx];
            fetch(min(wt + wstride, wtiles - 1), yn, xnx);
        } else {
            fetch(wt, yr, xr);
        }
        // the per-channel vectors of a lane do not depend on the tile; an opaque copy of fg keeps their (cheap) LDS reads
        // inside the loop instead of 5 x K/4 permanently live registers (see gemm_rows_kernel)
        int fgo = fg;
        asm volatile("" : "+v"(fgo));
        // ---- z = SiLU(W_z . LNmod(x)): a pixel's CX channels sit in the 4 lanes fr, fr + 16, fr + 32, fr + 48
        u32x4 zb[KS];
        {
            // sum and sum of squares straight from the packed pairs: v_dot2_f32_bf16 against (1, 1) and against itself
            // (exact products, fp32 accumulation) -- no unpack for the statistics
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int kx = 0; kx < KX; ++kx) {
                const uint32_t rw[4] = {xr[kx].x, xr[kx].y, xr[kx].z, xr[kx].w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    s1 = zre_dot2(rw[j], FD_H_ONES, s1);
                    s2 = zre_dot2(rw[j], rw[j], s2);
                }
            }
            s1 += __shfl_xor(s1, 16, 64);
            s2 += __shfl_xor(s2, 16, 64);
            s1 += __shfl_xor(s1, 32, 64);
            s2 += __shfl_xor(s2, 32, 64);
            const float mean = s1 * (1.f / CX);
            const float var = fmaxf(s2 * (1.f / CX) - mean * mean, 0.f);
            const float rstd = __builtin_amdgcn_rsqf(var + p.zre_eps), nm = -mean * rstd;
            bf16x8 xn[KX];
#pragma unroll
            for (int kx = 0; kx < KX; ++kx) {
                const int c = kx * 32 + fgo * 8;
                const uint32_t rw[4] = {xr[kx].x, xr[kx].y, xr[kx].z, xr[kx].w};
                uint32_t o[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const f32x2 t = fd_unpack_bf16(rw[j]) * rstd + nm;
                    o[j] = fd_pack_bf16(t * *(const f32x2 *)&sX[c + 2 * j] + *(const f32x2 *)&sX[CX + c + 2 * j]);
                }
                xn[kx] = __builtin_bit_cast(bf16x8, (u32x4){o[0], o[1], o[2], o[3]});
            }
#pragma unroll
            for (int ng = 0; ng < KS; ++ng) {
                f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kx = 0; kx < KX; ++kx) {
                    const bf16x8 wa = *(const bf16x8 *)(sZ + w_off(32 * ng + rperm, kx * 4 + fgo, RX));
                    const bf16x8 wb = *(const bf16x8 *)(sZ + w_off(32 * ng + rperm + 4, kx * 4 + fgo, RX));
                    a0 = FD_MFMA16(wa, xn[kx], a0, 0, 0, 0);
                    a1 = FD_MFMA16(wb, xn[kx], a1, 0, 0, 0);
                }
                float val[8] = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
                fd_silu8(val);
                zb[ng] = (u32x4){fd_pack_bf16(f32x2{val[0], val[1]}), fd_pack_bf16(f32x2{val[2], val[3]}),
                                 fd_pack_bf16(f32x2{val[4], val[5]}), fd_pack_bf16(f32x2{val[6], val[7]})};
            }
        }
        // ---- out_norm(y) * z + local (one-pass statistics like the x row's)
        bf16x8 xb[KS];
        {
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const uint32_t rw[4] = {yr[ks].x, yr[ks].y, yr[ks].z, yr[ks].w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    s1 = zre_dot2(rw[j], FD_H_ONES, s1);
                    s2 = zre_dot2(rw[j], rw[j], s2);
                }
            }
            s1 += __shfl_xor(s1, 16, 64);
            s2 += __shfl_xor(s2, 16, 64);
            s1 += __shfl_xor(s1, 32, 64);
            s2 += __shfl_xor(s2, 32, 64);
            const float mean = s1 * (1.f / K);
            const float var = fmaxf(s2 * (1.f / K) - mean * mean, 0.f);
            const float rstd = __builtin_amdgcn_rsqf(var + p.ln_eps), nm = -mean * rstd;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const int c = ks * 32 + fgo * 8;
                const uint32_t rw[4] = {yr[ks].x, yr[ks].y, yr[ks].z, yr[ks].w};
                const uint32_t zw[4] = {zb[ks].x, zb[ks].y, zb[ks].z, zb[ks].w};
                uint32_t o[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const f32x2 t = fd_unpack_bf16(rw[j]) * rstd + nm;
                    const f32x2 u = t * *(const f32x2 *)&sV[c + 2 * j] + *(const f32x2 *)&sV[K + c + 2 * j];
                    o[j] = fd_pack_bf16(u * fd_unpack_bf16(zw[j]) + *(const f32x2 *)&sV[2 * K + c + 2 * j]);
                }
                u32x4 pin = {o[0], o[1], o[2], o[3]};
                asm volatile("" : "+v"(pin));            // one K32 step at a time (register diet, see gemm_rows_kernel)
                xb[ks] = __builtin_bit_cast(bf16x8, pin);
            }
        }
        // ---- out_proj + gate . () + x
        const bool live = (int64_t)wt * 16 + fr < hw;
#pragma unroll
        for (int ng = 0; ng < KX; ++ng) {
            f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const bf16x8 wa = *(const bf16x8 *)(sW + w_off(32 * ng + rperm, ks * 4 + fgo, RS));
                const bf16x8 wb = *(const bf16x8 *)(sW + w_off(32 * ng + rperm + 4, ks * 4 + fgo, RS));
                a0 = FD_MFMA16(wa, xb[ks], a0, 0, 0, 0);
                a1 = FD_MFMA16(wb, xb[ks], a1, 0, 0, 0);
            }
            const int n0 = 32 * ng + 8 * fg;
            float bias[8], gt[8];
            if (p.bias) load8(p.bias + n0, bias);
            else {
#pragma unroll
                for (int e = 0; e < 8; ++e) bias[e] = 0.f;
            }
            load8(p.gate + (int64_t)b * p.gate_ld + n0, gt);
            const uint32_t rw[4] = {xr[ng].x, xr[ng].y, xr[ng].z, xr[ng].w};
            float val[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) { val[e] = a0[e] + bias[e]; val[4 + e] = a1[e] + bias[4 + e]; }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f32x2 r = fd_unpack_bf16(rw[j]);
                val[2 * j] = r.x + gt[2 * j] * val[2 * j];
                val[2 * j + 1] = r.y + gt[2 * j + 1] * val[2 * j + 1];
            }
            if (live) store8(outp + m * p.ldo + n0, val);
        }
    }
}

// 12 waves per workgroup leave 170 VGPRs per lane: enough for every variant without a prologue, for
// LN+modulate up to K=128 and for LN*z+local up to K=96; the wider ones stay at 8 waves (256 VGPRs).
constexpr bool fits_768(int KS, int PRO) { return PRO == 0 || (PRO == 1 && KS <= 4) || KS <= 3; }
// ... and the widest fused-LayerNorm variants need more than the 256 VGPRs of an 8-wave workgroup
constexpr bool fits_512(int KS, int PRO) { return !(PRO == 1 && KS >= 8) && !(PRO == 2 && KS >= 6); }

template <int KS, int PRO, int EPI, int NT>
void launch_rows_nt(const fd_conv_params &p, dim3 grid, size_t lds, int wtiles, int RS, hipStream_t s) {
    if (lds > 64 * 1024)
        (void)hipFuncSetAttribute((const void *)gemm_rows_kernel<KS, PRO, EPI, NT>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((gemm_rows_kernel<KS, PRO, EPI, NT>), grid, dim3(NT), lds, s, p, wtiles, RS);
}

template <int KS, int PRO, int EPI>
void launch_rows(const fd_conv_params &p, int nt, dim3 grid, size_t lds, int wtiles, int RS, hipStream_t s) {
    if (nt == 256) launch_rows_nt<KS, PRO, EPI, 256>(p, grid, lds, wtiles, RS, s);
    else if (nt == 512) {
        if constexpr (fits_512(KS, PRO)) launch_rows_nt<KS, PRO, EPI, 512>(p, grid, lds, wtiles, RS, s);
    } else if constexpr (fits_768(KS, PRO)) launch_rows_nt<KS, PRO, EPI, 768>(p, grid, lds, wtiles, RS, s);
}

}  // namespace

int fd_rows32_ok(const fd_conv_params &p);                          // fd_gemm_rows32.hip: fp32 storage, split-bf16 contractions
int fd_gemm_rows32_launch(const fd_conv_params &p, hipStream_t s);

// 1 if `p` can run on the streaming row-GEMM (and therefore may carry a fused LN prologue).
extern "C" int fd_conv_prologue_ok(const fd_conv_params *pp) {
    const fd_conv_params &p = *pp;
    const int K = p.c0 + p.c1;
    if (p.dtype == FD_F32) return fd_rows32_ok(p);
    if (p.dtype != FD_BF16 || p.out_f32) return 0;
    if (p.KH != 1 || p.KW != 1 || p.stride != 1 || p.pad_h != 0 || p.pad_w != 0 || p.upsample || p.ndir != 1) return 0;
    if (p.OH != p.H || p.OW != p.W) return 0;
    if (K % 32 || K > 256 || K == 160 || K == 224 || p.c0 % 8 || p.c1 % 8) return 0;
    if (p.Cout % 32 || (size_t)p.Cout * row_stride(K) > 128 * 1024) return 0;
    if (p.stats_partial) return 0;
    if (p.ldo % 8 || p.offo % 8 || p.ld0 % 8 || p.off0 % 8 || (p.in1 && (p.ld1 % 8 || p.off1 % 8))) return 0;
    if (p.res && (p.ld_res % 8 || p.off_res % 8)) return 0;
    if (p.epilogue == FD_EPI_GATE_RES && (p.gate_ld % 4 || ((uintptr_t)p.gate & 15))) return 0;
    if (p.epilogue == FD_EPI_SILU_SPLIT && p.epi_split % 8) return 0;
    if (p.bias && ((uintptr_t)p.bias & 15)) return 0;
    if ((p.epilogue == FD_EPI_GNSILU_ADD || p.epilogue == FD_EPI_GNSILU_ADD_FINAL) && (p.gn_groups <= 0 || (p.Cout / p.gn_groups) % 8)) return 0;
    if (p.epilogue == FD_EPI_GNSILU_ADD_FINAL && (!p.fin_w || !p.fin_out || ((uintptr_t)p.fin_w & 15) ||
                                                  (p.fin_mode == 1 && (!p.fin_img || !p.fin_xin)))) return 0;
    if (p.prologue < FD_PRO_NONE || p.prologue > FD_PRO_LN_GATE_ZRE) return 0;
    if (p.prologue == FD_PRO_LN_MOD && p.epilogue != FD_EPI_NONE && p.epilogue != FD_EPI_SILU_SPLIT) return 0;
    if ((p.prologue == FD_PRO_LN_GATE || p.prologue == FD_PRO_LN_GATE_ZRE) && p.epilogue != FD_EPI_GATE_RES) return 0;
    if (p.prologue != FD_PRO_NONE) {
        if (p.c1 != 0) return 0;
        if (p.prologue == FD_PRO_LN_GATE && (!p.ln_z || p.ln_ldz % 8 || p.ln_offz % 8 || !p.ln_gamma || !p.ln_beta)) return 0;
        if (!p.ln_shift || (p.prologue == FD_PRO_LN_MOD && !p.ln_scale)) return 0;
        if (p.prologue == FD_PRO_LN_GATE_ZRE) {
            // z recomputed from the residual operand: d_inner = 2 dim, shared (not per-batch) weights, the K = 128 instance
            if ((K != 128 && K != 256) || 2 * p.Cout != K || p.w_batch_stride != 0 || !p.res || !p.ln_gamma || !p.ln_beta) return 0;
            if (!p.zre_w || ((uintptr_t)p.zre_w & 15) || !p.zre_shift || !p.zre_scale) return 0;
        }
    }
    if ((int64_t)p.H * p.W < 16384) return 0;     // few pixels: the tiled kernel parallelises better
    return 1;
}

constexpr int ZRE8_NT = 512;

int fd_gemm_rows_launch(const fd_conv_params &p, hipStream_t s) {
    if (p.dtype == FD_F32) return fd_gemm_rows32_launch(p, s);
    const int K = p.c0 + p.c1, KS = K / 32, RS = row_stride(K);
    const int64_t hw = (int64_t)p.H * p.W;
    const int px = KS <= 2 ? 32 : 16;      // pixels per wave iteration (see template parameter S)
    const bool zre = p.prologue == FD_PRO_LN_GATE_ZRE;
    const int wtiles = (int)((hw + (zre ? 16 : px) - 1) / (zre ? 16 : px));
    const size_t lds = (size_t)p.Cout * RS + 3 * (size_t)K * sizeof(float) +
                       (zre ? (size_t)K * row_stride(K / 2) + (size_t)K * sizeof(float) : 0);
    int per_cu = (int)(150 * 1024 / lds);
    // (round 4: 3 persistent workgroups per CU instead of 4 once the batch fills the chip: -0.1..-0.3 % per batch-8 forward,
    //  2: +0.8 %; 4 for small batches: +0.7 ms per 50-step slice at batch 1 otherwise.  The grid size does not touch the
    //  results: every pixel row is computed by itself)
    const int pcmax = fd_dev(FD_DEV_ROWS_PER_CU_MAX);    // development
    const int pclim = pcmax > 0 ? pcmax : (p.B >= 4 ? 3 : 4);
    if (per_cu > pclim) per_cu = pclim;
    if (zre && per_cu > 3) per_cu = 3;       // 168 VGPRs with the next tile's rows in flight: 3 waves per SIMD
    if (per_cu < 1) per_cu = 1;
    int nt = per_cu >= 3 ? 256 : ((per_cu == 2 || !fits_768(KS, p.prologue)) ? 512 : 768);
    if (nt == 512 && !fits_512(KS, p.prologue)) nt = 256;
    if (zre && K == 256) nt = ZRE8_NT;
    const int cap = fd_dev(FD_DEV_ROWS_PER_CU);   // development: see fd_occ_pad
    int gx = (256 * ((cap > 0 && per_cu > cap) ? cap : per_cu) + p.B - 1) / p.B;
    const int need = (wtiles + nt / 64 - 1) / (nt / 64);
    if (gx > need) gx = need;
    dim3 grid(gx, p.B);
    if (zre) {
        const bool nopf = fd_dev(FD_DEV_ZRE_NOPF);      // development
        if (K == 128 && nopf) hipLaunchKernelGGL((gemm_rows_zre_kernel<4, 256, false>), grid, dim3(256), lds, s, p, wtiles);
        else if (K == 128) hipLaunchKernelGGL((gemm_rows_zre_kernel<4, 256, true>), grid, dim3(256), lds, s, p, wtiles);
        else {
            // K = 256: 64 KB of out_proj + 64 KB of W_z rows -- one workgroup per CU, so its waves are the CU's waves
            (void)hipFuncSetAttribute((const void *)gemm_rows_zre_kernel<8, ZRE8_NT, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipLaunchKernelGGL((gemm_rows_zre_kernel<8, ZRE8_NT, false>), grid, dim3(ZRE8_NT), lds, s, p, wtiles);
        }
        return 0;
    }
#define FD_GR(KS_, PRO_, EPI_) launch_rows<KS_, PRO_, EPI_>(p, nt, grid, lds, wtiles, RS, s)
#define FD_GR_K(PRO_, EPI_)                            \
    switch (KS) {                                      \
    case 1: FD_GR(1, PRO_, EPI_); break;               \
    case 2: FD_GR(2, PRO_, EPI_); break;               \
    case 3: FD_GR(3, PRO_, EPI_); break;               \
    case 4: FD_GR(4, PRO_, EPI_); break;               \
    case 6: FD_GR(6, PRO_, EPI_); break;               \
    case 8: FD_GR(8, PRO_, EPI_); break;               \
    default: return -1;                                \
    }
    if (p.prologue == FD_PRO_LN_MOD) {
        if (p.epilogue == FD_EPI_SILU_SPLIT) { FD_GR_K(1, FD_EPI_SILU_SPLIT) } else { FD_GR_K(1, FD_EPI_NONE) }
    } else if (p.prologue == FD_PRO_LN_GATE) {
        FD_GR_K(2, FD_EPI_GATE_RES)
    } else {
        switch (p.epilogue) {
        case FD_EPI_NONE: FD_GR_K(0, FD_EPI_NONE) break;
        case FD_EPI_SILU_SPLIT: FD_GR_K(0, FD_EPI_SILU_SPLIT) break;
        case FD_EPI_RELU: FD_GR_K(0, FD_EPI_RELU) break;
        case FD_EPI_GATE_RES: FD_GR_K(0, FD_EPI_GATE_RES) break;
        case FD_EPI_RES_RELU: FD_GR_K(0, FD_EPI_RES_RELU) break;
        case FD_EPI_GNSILU_ADD: FD_GR_K(0, FD_EPI_GNSILU_ADD) break;
        case FD_EPI_GNSILU_ADD_FINAL: FD_GR_K(0, FD_EPI_GNSILU_ADD_FINAL) break;
        default: return -1;
        }
    }
#undef FD_GR_K
#undef FD_GR
    return 0;
}
