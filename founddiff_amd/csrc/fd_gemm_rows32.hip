// fd_gemm_rows32.hip -- the streaming row-GEMM of fd_gemm_rows.hip on FP32 STORAGE with split-bf16 contractions: the
// 1x1 convolutions of the `fp32s` engine (precision='fp32s': the mode that carries the north star's 1e-3 gate over
// the whole sampling loop, src/DADiff.py:1276-1365) at >= 16384 pixels per image.
//
// Until round 6 the fp32s engine ran these layers on the generic split implicit GEMM with the LayerNorm producers as
// separate row kernels (fd_ln_modulate, fd_ln_gate) and in_proj writing a z half that out_proj read back: 12 of the
// 34 ms of a batch-8 forward were 1x1 layers moving fp32 tensors the bf16 engine never stores.  Same dataflow here as
// in the bf16 kernels:
//   * the weight matrix lives in LDS for the lifetime of a persistent workgroup -- as TWO bf16 images, w = hi + lo
//     (hi = bf16(w), lo = bf16(w - hi), split while staging: the caller passes the fp32 matrix);
//   * transposed issue D^T = W . X^T: a lane's B operand is 8 consecutive channels of one pixel = two 16-byte loads
//     of the fp32 NHWC row, split into x_hi / x_lo in registers; every product is x_hi.w_hi + x_lo.w_hi + x_hi.w_lo
//     on v_mfma_f32_16x16x32_bf16 (the lo.lo term is below 2^-16 of the product and dropped, as in
//     conv_igemm_kernel<float, ..., SPL> and the split halo kernel);
//   * LayerNorm statistics of a pixel row = two xor-shuffles (its K channels sit in lanes l, l^16, l^32, l^48), in
//     fp32 and two passes (centred: these rows are fp32, fd_common.h explains why the one-pass form is bf16-only):
//     LN + adaLN-modulate (in_proj, qkv: src/DADiff.py:450-451, 477-488), out_norm . z + local (out_proj:
//     src/emamba2.py:747-748) and the z gate recomputed from the block input (FD_PRO_LN_GATE_ZRE) are prologues;
//   * a lane ends up with 8 consecutive output channels of its pixel: two 16-byte stores, residual / GroupNorm
//     operands by the matching loads.
#include "fd_common.h"

namespace {

// LDS image of one bf16 half of W: row n at n*RS bytes, 16-byte chunks XOR-swizzled (fd_gemm_rows.hip: w_off)
__device__ __forceinline__ int w_off32(int row, int chunk, int RS) {
    const int sw = RS == 128 ? (((row >> 1) & 1) | (((row >> 3) & 3) << 1))
                             : ((row & 3) | (((row >> 3) & 3) << 2));
    return row * RS + ((chunk ^ sw) << 4);
}

__device__ __forceinline__ void split8(const float (&f)[8], bf16x8 &hi, bf16x8 &lo) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const bf16 h = (bf16)f[e];
        hi[e] = h;
        lo[e] = (bf16)(f[e] - (float)h);
    }
}

// rows [0, nrows) of an fp32 matrix [nrows][K] -> two swizzled bf16 images
__device__ __forceinline__ void stage_split(const float *wg, int nrows, int K, int RS, unsigned char *sHi, unsigned char *sLo,
                                            int tid, int nt) {
    const int cpr = K / 8;
    for (int idx = tid; idx < nrows * cpr; idx += nt) {
        const int row = idx / cpr, ch = idx - row * cpr;
        float f[8];
        load8(wg + (int64_t)row * K + ch * 8, f);
        bf16x8 hi, lo;
        split8(f, hi, lo);
        *(bf16x8 *)(sHi + w_off32(row, ch, RS)) = hi;
        *(bf16x8 *)(sLo + w_off32(row, ch, RS)) = lo;
    }
}

// acc += W[rows of this lane's fragment] . x, three bf16 MFMAs per K32 step
#define FD_MFMA3(acc, wh, wl, xh, xl)                                                  \
    do {                                                                               \
        acc = FD_MFMA16(wl, xh, acc, 0, 0, 0);           \
        acc = FD_MFMA16(wh, xl, acc, 0, 0, 0);           \
        acc = FD_MFMA16(wh, xh, acc, 0, 0, 0);           \
    } while (0)

template <int KS, int PRO, int EPI, int NT>
__global__ __launch_bounds__(NT) void gemm_rows32_kernel(const fd_conv_params p, int wtiles, int RS) {
    constexpr int NW = NT / 64;
    constexpr int K = 32 * KS;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int N = p.Cout;
    unsigned char *sWh = smem, *sWl = smem + (size_t)N * RS;
    float *sV = (float *)(smem + 2 * (size_t)N * RS);          // prologue vectors: [3][K]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fg = lane >> 4;
    const int b = blockIdx.y;
    const int64_t hw = (int64_t)p.H * p.W;
    stage_split((const float *)p.weight + (int64_t)b * p.w_batch_stride, N, K, RS, sWh, sWl, tid, NT);
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

    const float *in0 = (const float *)p.in0 + (int64_t)b * hw * p.ld0 + p.off0;
    const float *in1 = p.in1 ? (const float *)p.in1 + (int64_t)b * hw * p.ld1 + p.off1 : nullptr;
    const float *zin = PRO == 2 ? (const float *)p.ln_z + (int64_t)b * hw * p.ln_ldz + p.ln_offz : nullptr;
    float *outp = (float *)p.out + (int64_t)b * hw * p.ldo + p.offo;
    const float *resp = p.res ? (const float *)p.res + (int64_t)b * hw * p.ld_res + p.off_res : nullptr;
    const float *hp = p.h ? (const float *)p.h + (int64_t)b * hw * N : nullptr;
    const int cpg = p.gn_groups > 0 ? N / p.gn_groups : 1;
    const int rperm = 8 * (fr >> 2) + (fr & 3);

    const int wstride = gridDim.x * NW;
    for (int wt = blockIdx.x * NW + wave; wt < wtiles; wt += wstride) {
        // rows beyond the image (ragged last tile) read the last row instead: their results are never stored
        const int64_t m = min((int64_t)wt * 16 + fr, hw - 1);
        const bool live = (int64_t)wt * 16 + fr < hw;
        bf16x8 xh[KS], xl[KS];
        {
            float xf[KS][8];
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const int c = ks * 32 + fg * 8;
                const float *src = c < p.c0 ? in0 + m * p.ld0 + c : in1 + m * p.ld1 + (c - p.c0);
                load8(src, xf[ks]);
            }
            if (PRO != 0) {
                // (an opaque copy of fg keeps the per-channel vectors LDS reads inside the pixel loop: fd_gemm_rows.hip)
                int fgo = fg;
                asm volatile("" : "+v"(fgo));
                float sum = 0.f;
#pragma unroll
                for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                    for (int e = 0; e < 8; ++e) sum += xf[ks][e];
                sum += __shfl_xor(sum, 16, 64);
                sum += __shfl_xor(sum, 32, 64);
                const float mean = sum * (1.f / K);
                float q = 0.f;
#pragma unroll
                for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                    for (int e = 0; e < 8; ++e) { const float d = xf[ks][e] - mean; q += d * d; }
                q += __shfl_xor(q, 16, 64);
                q += __shfl_xor(q, 32, 64);
                const float rstd = rsqrtf(q * (1.f / K) + p.ln_eps);
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const int c = ks * 32 + fgo * 8;
                    float g8[8], b8[8];
                    load8(sV + c, g8);
                    load8(sV + K + c, b8);
                    if (PRO == 1) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) xf[ks][e] = (xf[ks][e] - mean) * rstd * g8[e] + b8[e];
                    } else {
                        float z8[8], l8[8];
                        load8(zin + m * p.ln_ldz + c, z8);
                        load8(sV + 2 * K + c, l8);
#pragma unroll
                        for (int e = 0; e < 8; ++e) xf[ks][e] = ((xf[ks][e] - mean) * rstd * g8[e] + b8[e]) * z8[e] + l8[e];
                    }
                }
            }
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) split8(xf[ks], xh[ks], xl[ks]);
        }
        float fdot = 0.f;                    // GNSILU_ADD_FINAL: this lane's share of the final 1x1 (Cout -> 1)
#pragma unroll 1
        for (int ng = 0; ng < N / 32; ++ng) {
            f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
            const int rowa = 32 * ng + rperm;
            int fgw = fg;
            asm volatile("" : "+v"(fgw));    // weight fragments stay LDS reads per tile (register diet, fd_gemm_rows.hip)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const int oa = w_off32(rowa, ks * 4 + fgw, RS), ob = w_off32(rowa + 4, ks * 4 + fgw, RS);
                const bf16x8 wah = *(const bf16x8 *)(sWh + oa), wal = *(const bf16x8 *)(sWl + oa);
                const bf16x8 wbh = *(const bf16x8 *)(sWh + ob), wbl = *(const bf16x8 *)(sWl + ob);
                FD_MFMA3(a0, wah, wal, xh[ks], xl[ks]);
                FD_MFMA3(a1, wbh, wbl, xh[ks], xl[ks]);
            }
            // lane (pixel fr, group fg) holds channels n0 .. n0+7
            const int n0 = 32 * ng + 8 * fg;
            float val[8];
            if (p.bias) {
                float bias[8];
                load8(p.bias + n0, bias);
#pragma unroll
                for (int e = 0; e < 4; ++e) { val[e] = a0[e] + bias[e]; val[4 + e] = a1[e] + bias[4 + e]; }
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) { val[e] = a0[e]; val[4 + e] = a1[e]; }
            }
            if (EPI == FD_EPI_SILU_SPLIT) {
                if (n0 >= p.epi_split) fd_silu8(val);
            } else if (EPI == FD_EPI_RELU) {
#pragma unroll
                for (int e = 0; e < 8; ++e) val[e] = fmaxf(val[e], 0.f);
            } else if (EPI == FD_EPI_GATE_RES || EPI == FD_EPI_RES_RELU) {
                float rs[8];
                load8(resp + m * p.ld_res + n0, rs);
                if (EPI == FD_EPI_GATE_RES) {
                    float gt[8];
                    load8(p.gate + (int64_t)b * p.gate_ld + n0, gt);
#pragma unroll
                    for (int e = 0; e < 8; ++e) val[e] = __builtin_fmaf(gt[e], val[e], rs[e]);
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) val[e] = fmaxf(val[e] + rs[e], 0.f);
                }
            } else if (EPI == FD_EPI_GNSILU_ADD || EPI == FD_EPI_GNSILU_ADD_FINAL) {
                float hv[8], ga[8], be[8];
                load8(hp + m * N + n0, hv);
                load8(p.gn_gamma + n0, ga);
                load8(p.gn_beta + n0, be);
                // channels-per-group is a multiple of 8 (checked on the host): one group per vector
                const int g = n0 / cpg;
                const float gm = p.gn_mean_rstd[((int64_t)b * p.gn_groups + g) * 2];
                const float gr = p.gn_mean_rstd[((int64_t)b * p.gn_groups + g) * 2 + 1];
                fd_gn_silu_add8(val, hv, gm, gr, ga, be);
            }
            if (EPI == FD_EPI_GNSILU_ADD_FINAL) {
                float fw[8];
                load8(p.fin_w + n0, fw);
#pragma unroll
                for (int e = 0; e < 8; ++e) fdot += val[e] * fw[e];
            } else if (live) {
                store8(outp + m * p.ldo + n0, val);
            }
        }
        if (EPI == FD_EPI_GNSILU_ADD_FINAL) {
            // a pixel's Cout channels sit in the 4 lanes fr, fr + 16, fr + 32, fr + 48: two xor-shuffles
            float o = fdot;
            o += __shfl_xor(o, 16, 64);
            o += __shfl_xor(o, 32, 64);
            o += p.fin_b;
            if (fg == 0 && live) {
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

// ---------------------------------------------------------------------------------------------------------------
// out_proj with the z gate recomputed (FD_PRO_LN_GATE_ZRE) on fp32 storage, K = d_inner = 128 (the 64-channel Mamba
// blocks at 512x512 / 256x256): gemm_rows_zre_kernel<4> of fd_gemm_rows.hip with fp32 rows and split contractions.
//   x row (CX = 64 channels) -> LayerNorm + adaLN modulate -> split -> W_z (hi, lo in LDS) -> SiLU = z (fp32, registers)
//   y row (K = 128) -> out_norm -> . z + local -> split -> out_proj (hi, lo in LDS) -> x + gate . (acc + bias).
// Per pixel: (K + CX + CX) x 4 bytes instead of the (K + K + CX + CX) x 4 of the stored-z form, and in_proj writes its
// x half only (no 2 d_inner-wide fp32 tensor at all).
template <int NT>
__global__ __launch_bounds__(NT, 2) void gemm_rows32_zre_kernel(const fd_conv_params p, int wtiles) {
    constexpr int KS = 4, NW = NT / 64, K = 128, KX = 2, CX = 64, RS = 256, RX = 128;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char *sWh = smem, *sWl = smem + CX * RS;          // out_proj  [CX rows][RS] x 2
    unsigned char *sZh = smem + 2 * CX * RS, *sZl = sZh + K * RX;   // W_z  [K rows][RX] x 2
    float *sV = (float *)(sZl + K * RX);                       // out_norm gamma, beta, local: [3][K]
    float *sX = sV + 3 * K;                                    // norm1 G = gamma (1 + scale), Bc = beta (1 + scale) + shift: [2][CX]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fg = lane >> 4;
    const int b = blockIdx.y;
    const int64_t hw = (int64_t)p.H * p.W;
    stage_split((const float *)p.weight, CX, K, RS, sWh, sWl, tid, NT);
    stage_split((const float *)p.zre_w, K, CX, RX, sZh, sZl, tid, NT);
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
    __syncthreads();
    const float *yin = (const float *)p.in0 + (int64_t)b * hw * p.ld0 + p.off0;
    const float *xin = (const float *)p.res + (int64_t)b * hw * p.ld_res + p.off_res;
    float *outp = (float *)p.out + (int64_t)b * hw * p.ldo + p.offo;
    const int rperm = 8 * (fr >> 2) + (fr & 3);
    const int wstride = gridDim.x * NW;
    for (int wt = blockIdx.x * NW + wave; wt < wtiles; wt += wstride) {
        const int64_t m = min((int64_t)wt * 16 + fr, hw - 1);
        const bool live = (int64_t)wt * 16 + fr < hw;
        float yf[KS][8], xr[KX][8];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) load8(yin + m * p.ld0 + ks * 32 + fg * 8, yf[ks]);
#pragma unroll
        for (int kx = 0; kx < KX; ++kx) load8(xin + m * p.ld_res + kx * 32 + fg * 8, xr[kx]);
        int fgo = fg;
        asm volatile("" : "+v"(fgo));
        // ---- z = SiLU(W_z . LNmod(x))
        float zf[KS][8];
        {
            float sum = 0.f;
#pragma unroll
            for (int kx = 0; kx < KX; ++kx)
#pragma unroll
                for (int e = 0; e < 8; ++e) sum += xr[kx][e];
            sum += __shfl_xor(sum, 16, 64);
            sum += __shfl_xor(sum, 32, 64);
            const float mean = sum * (1.f / CX);
            float q = 0.f;
#pragma unroll
            for (int kx = 0; kx < KX; ++kx)
#pragma unroll
                for (int e = 0; e < 8; ++e) { const float d = xr[kx][e] - mean; q += d * d; }
            q += __shfl_xor(q, 16, 64);
            q += __shfl_xor(q, 32, 64);
            const float rstd = rsqrtf(q * (1.f / CX) + p.zre_eps);
            bf16x8 nh[KX], nl[KX];
#pragma unroll
            for (int kx = 0; kx < KX; ++kx) {
                const int c = kx * 32 + fgo * 8;
                float g8[8], b8[8], t[8];
                load8(sX + c, g8);
                load8(sX + CX + c, b8);
#pragma unroll
                for (int e = 0; e < 8; ++e) t[e] = (xr[kx][e] - mean) * rstd * g8[e] + b8[e];
                split8(t, nh[kx], nl[kx]);
            }
#pragma unroll
            for (int ng = 0; ng < KS; ++ng) {
                f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kx = 0; kx < KX; ++kx) {
                    const int oa = w_off32(32 * ng + rperm, kx * 4 + fgo, RX), ob = w_off32(32 * ng + rperm + 4, kx * 4 + fgo, RX);
                    const bf16x8 wah = *(const bf16x8 *)(sZh + oa), wal = *(const bf16x8 *)(sZl + oa);
                    const bf16x8 wbh = *(const bf16x8 *)(sZh + ob), wbl = *(const bf16x8 *)(sZl + ob);
                    FD_MFMA3(a0, wah, wal, nh[kx], nl[kx]);
                    FD_MFMA3(a1, wbh, wbl, nh[kx], nl[kx]);
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) { zf[ng][e] = a0[e]; zf[ng][4 + e] = a1[e]; }
                fd_silu8(zf[ng]);
            }
        }
        // ---- out_norm(y) * z + local
        bf16x8 xh[KS], xl[KS];
        {
            float sum = 0.f;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int e = 0; e < 8; ++e) sum += yf[ks][e];
            sum += __shfl_xor(sum, 16, 64);
            sum += __shfl_xor(sum, 32, 64);
            const float mean = sum * (1.f / K);
            float q = 0.f;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int e = 0; e < 8; ++e) { const float d = yf[ks][e] - mean; q += d * d; }
            q += __shfl_xor(q, 16, 64);
            q += __shfl_xor(q, 32, 64);
            const float rstd = rsqrtf(q * (1.f / K) + p.ln_eps);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const int c = ks * 32 + fgo * 8;
                float g8[8], b8[8], l8[8], t[8];
                load8(sV + c, g8);
                load8(sV + K + c, b8);
                load8(sV + 2 * K + c, l8);
#pragma unroll
                for (int e = 0; e < 8; ++e) t[e] = ((yf[ks][e] - mean) * rstd * g8[e] + b8[e]) * zf[ks][e] + l8[e];
                split8(t, xh[ks], xl[ks]);
            }
        }
        // ---- out_proj + gate . () + x
#pragma unroll
        for (int ng = 0; ng < KX; ++ng) {
            f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const int oa = w_off32(32 * ng + rperm, ks * 4 + fgo, RS), ob = w_off32(32 * ng + rperm + 4, ks * 4 + fgo, RS);
                const bf16x8 wah = *(const bf16x8 *)(sWh + oa), wal = *(const bf16x8 *)(sWl + oa);
                const bf16x8 wbh = *(const bf16x8 *)(sWh + ob), wbl = *(const bf16x8 *)(sWl + ob);
                FD_MFMA3(a0, wah, wal, xh[ks], xl[ks]);
                FD_MFMA3(a1, wbh, wbl, xh[ks], xl[ks]);
            }
            const int n0 = 32 * ng + 8 * fg;
            float bias[8], gt[8], val[8];
            if (p.bias) load8(p.bias + n0, bias);
            else {
#pragma unroll
                for (int e = 0; e < 8; ++e) bias[e] = 0.f;
            }
            load8(p.gate + (int64_t)b * p.gate_ld + n0, gt);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                val[e] = xr[ng][e] + gt[e] * (a0[e] + bias[e]);
                val[4 + e] = xr[ng][4 + e] + gt[4 + e] * (a1[e] + bias[4 + e]);
            }
            if (live) store8(outp + m * p.ldo + n0, val);
        }
    }
}

int row_stride32(int K) { return K <= 64 ? 128 : (K <= 128 ? 256 : 512); }

size_t rows32_lds(const fd_conv_params &p) {
    const int K = p.c0 + p.c1;
    if (p.prologue == FD_PRO_LN_GATE_ZRE) return 2 * (size_t)64 * 256 + 2 * (size_t)128 * 128 + (3 * 128 + 2 * 64) * sizeof(float);
    return 2 * (size_t)p.Cout * row_stride32(K) + 3 * (size_t)K * sizeof(float);
}

template <int KS, int PRO, int EPI>
void launch_rows32(const fd_conv_params &p, int nt, dim3 grid, size_t lds, int wtiles, int RS, hipStream_t s) {
    if (nt == 256) {
        if (lds > 64 * 1024)
            (void)hipFuncSetAttribute((const void *)gemm_rows32_kernel<KS, PRO, EPI, 256>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((gemm_rows32_kernel<KS, PRO, EPI, 256>), grid, dim3(256), lds, s, p, wtiles, RS);
    } else {
        if (lds > 64 * 1024)
            (void)hipFuncSetAttribute((const void *)gemm_rows32_kernel<KS, PRO, EPI, 512>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((gemm_rows32_kernel<KS, PRO, EPI, 512>), grid, dim3(512), lds, s, p, wtiles, RS);
    }
}

}  // namespace

// 1 if `p` (fp32 storage, f32_split) can run on the fp32 streaming row-GEMM, and therefore may carry a fused LN prologue.
int fd_rows32_ok(const fd_conv_params &p) {
    const int K = p.c0 + p.c1;
    if (p.dtype != FD_F32 || !p.f32_split || p.out_f32) return 0;
    if (p.KH != 1 || p.KW != 1 || p.stride != 1 || p.pad_h != 0 || p.pad_w != 0 || p.upsample || p.ndir != 1) return 0;
    if (p.OH != p.H || p.OW != p.W) return 0;
    if (K % 32 || K > 256 || K == 160 || K == 224 || p.c0 % 8 || p.c1 % 8) return 0;
    if (p.Cout % 32) return 0;
    if (p.stats_partial) return 0;
    if (p.ldo % 4 || p.offo % 4 || p.ld0 % 4 || p.off0 % 4 || (p.in1 && (p.ld1 % 4 || p.off1 % 4))) return 0;
    if (((uintptr_t)p.in0 & 15) || ((uintptr_t)p.out & 15) || ((uintptr_t)p.weight & 15) || (p.in1 && ((uintptr_t)p.in1 & 15))) return 0;
    if (p.w_batch_stride % 4) return 0;
    if (p.res && (p.ld_res % 4 || p.off_res % 4 || ((uintptr_t)p.res & 15))) return 0;
    if (p.epilogue == FD_EPI_GATE_RES && (p.gate_ld % 4 || ((uintptr_t)p.gate & 15))) return 0;
    if (p.epilogue == FD_EPI_SILU_SPLIT && p.epi_split % 8) return 0;
    if (p.bias && ((uintptr_t)p.bias & 15)) return 0;
    if ((p.epilogue == FD_EPI_GNSILU_ADD || p.epilogue == FD_EPI_GNSILU_ADD_FINAL) &&
        (p.gn_groups <= 0 || (p.Cout / p.gn_groups) % 8 || !p.h || ((uintptr_t)p.h & 15))) return 0;
    if (p.epilogue == FD_EPI_GNSILU_ADD_FINAL && (!p.fin_w || !p.fin_out || ((uintptr_t)p.fin_w & 15) ||
                                                  (p.fin_mode == 1 && (!p.fin_img || !p.fin_xin)))) return 0;
    if (p.prologue < FD_PRO_NONE || p.prologue > FD_PRO_LN_GATE_ZRE) return 0;
    if (p.prologue == FD_PRO_LN_MOD && p.epilogue != FD_EPI_NONE && p.epilogue != FD_EPI_SILU_SPLIT) return 0;
    if ((p.prologue == FD_PRO_LN_GATE || p.prologue == FD_PRO_LN_GATE_ZRE) && p.epilogue != FD_EPI_GATE_RES) return 0;
    if (p.prologue != FD_PRO_NONE) {
        if (p.c1 != 0) return 0;
        if (p.prologue == FD_PRO_LN_GATE && (!p.ln_z || p.ln_ldz % 4 || p.ln_offz % 4 || ((uintptr_t)p.ln_z & 15) || !p.ln_gamma || !p.ln_beta)) return 0;
        if (!p.ln_shift || (p.prologue == FD_PRO_LN_MOD && !p.ln_scale)) return 0;
        if (p.prologue == FD_PRO_LN_GATE_ZRE) {
            if (K != 128 || p.Cout != 64 || p.w_batch_stride != 0 || !p.res || !p.ln_gamma || !p.ln_beta) return 0;
            if (!p.zre_w || ((uintptr_t)p.zre_w & 15) || !p.zre_shift || !p.zre_scale) return 0;
        }
    }
    if (rows32_lds(p) > 150 * 1024) return 0;
    if ((int64_t)p.H * p.W < 16384) return 0;     // few pixels: the tiled kernel parallelises better
    return 1;
}

int fd_gemm_rows32_launch(const fd_conv_params &p, hipStream_t s) {
    const int K = p.c0 + p.c1, KS = K / 32, RS = row_stride32(K);
    const int64_t hw = (int64_t)p.H * p.W;
    const int wtiles = (int)((hw + 15) / 16);
    const size_t lds = rows32_lds(p);
    int per_cu = (int)(150 * 1024 / lds);
    const int pcmax = fd_dev(FD_DEV_ROWS32_PER_CU);
    if (per_cu > pcmax) per_cu = pcmax;
    if (per_cu < 1) per_cu = 1;
    // one resident workgroup per CU: its waves are the CU's waves (8 of them); otherwise 4-wave workgroups
    const int nt = per_cu >= 2 ? 256 : 512;
    int gx = (256 * per_cu + p.B - 1) / p.B;
    const int need = (wtiles + nt / 64 - 1) / (nt / 64);
    if (gx > need) gx = need;
    dim3 grid(gx, p.B);
    if (p.prologue == FD_PRO_LN_GATE_ZRE) {
        if (lds > 64 * 1024)
            (void)hipFuncSetAttribute((const void *)gemm_rows32_zre_kernel<256>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((gemm_rows32_zre_kernel<256>), grid, dim3(256), lds, s, p, wtiles);
        return 0;
    }
#define FD_GR(KS_, PRO_, EPI_) launch_rows32<KS_, PRO_, EPI_>(p, nt, grid, lds, wtiles, RS, s)
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
