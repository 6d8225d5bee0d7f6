// fd_vanilla.hip -- kernels only the vanilla DDPM U-Net (src/denoising_diffusion_pytorch.py) needs:
// FiLM-modulated GroupNorm tail, channel LayerNorm with gain + residual, LinearAttention
// (softmax over d for q, over n for k, 32x32 context per head) and the bottleneck softmax
// attention.  Correctness-first (config 1 is the reference's CPU-runnable plumbing case).
#include "fd_common.h"

namespace {

// out = silu( GN(h) * (1 + scale[b]) + shift[b] )           (Block with scale_shift, 190-199)
template <typename T>
__global__ void gn_film_silu_kernel(const T *__restrict__ h, const float *__restrict__ mean_rstd,
                                    const float *__restrict__ gamma, const float *__restrict__ beta,
                                    const float *__restrict__ fscale, const float *__restrict__ fshift, int film_ld,
                                    T *__restrict__ out, int64_t hw, int C, int groups, int64_t nvec_per_img) {
    const int b = blockIdx.y, cpg = C / groups, vpr = C / 8;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < nvec_per_img;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int c0 = (int)(i % vpr) * 8;
        const int64_t off = (int64_t)b * hw * C + i * 8;
        float hv[8], o[8];
        load8(h + off, hv);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int c = c0 + e, g = c / cpg;
            const float mean = mean_rstd[((int64_t)b * groups + g) * 2], rstd = mean_rstd[((int64_t)b * groups + g) * 2 + 1];
            float y = (hv[e] - mean) * rstd * gamma[c] + beta[c];
            if (fscale) y = y * (1.f + fscale[(int64_t)b * film_ld + c]) + fshift[(int64_t)b * film_ld + c];
            o[e] = fd_silu(y);
        }
        store8(out + off, o);
    }
}

// out = LN_c(x) * g (+ res): reference LayerNorm(dim) with gain only (127-136), eps 1e-5
template <typename T>
__global__ __launch_bounds__(256) void chan_ln_kernel(const T *__restrict__ x, const float *__restrict__ g,
                                                     const T *__restrict__ res, T *__restrict__ out, int C,
                                                     int64_t nrows) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= nrows) return;
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s += ld1(x + row * C + c);
    const float mean = wave_sum(s) / C;
    float q = 0.f;
    for (int c = lane; c < C; c += 64) { const float d = ld1(x + row * C + c) - mean; q += d * d; }
    const float rstd = rsqrtf(wave_sum(q) / C + 1e-5f);
    for (int c = lane; c < C; c += 64) {
        float y = (ld1(x + row * C + c) - mean) * rstd * g[c];
        if (res) y += ld1(res + row * C + c);
        st1(out + row * C + c, y);
    }
}

// per (batch, channel) over pixels: max and sum exp(k - max)  (k.softmax(dim=-1), 246)
template <typename T>
__global__ __launch_bounds__(256) void col_softmax_stats_kernel(const T *__restrict__ x, int ld, int off, int64_t hw,
                                                               int Cn, float *__restrict__ stats) {
    __shared__ float red[4];
    const int b = blockIdx.y, c = blockIdx.x, tid = threadIdx.x;
    const T *p = x + (int64_t)b * hw * ld + off + c;
    float mx = -3.4e38f;
    for (int64_t n = tid; n < hw; n += 256) mx = fmaxf(mx, ld1(p + n * ld));
    mx = wave_max(mx);
    if ((tid & 63) == 0) red[tid >> 6] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float s = 0.f;
    for (int64_t n = tid; n < hw; n += 256) s += __expf(ld1(p + n * ld) - mx);
    s = wave_sum(s);
    if ((tid & 63) == 0) red[tid >> 6] = s;
    __syncthreads();
    if (tid == 0) {
        stats[((int64_t)b * Cn + c) * 2] = mx;
        stats[((int64_t)b * Cn + c) * 2 + 1] = red[0] + red[1] + red[2] + red[3];
    }
}

// context[b,h,d,e] = sum_n softmax_n(k)[d,n] * v[e,n] / hw, one workgroup per (b, head): fixed order
template <typename T>
__global__ __launch_bounds__(256) void linattn_context_kernel(const T *__restrict__ qkv, int hidden, int64_t hw,
                                                             const float *__restrict__ kstats,
                                                             float *__restrict__ ctx) {
    __shared__ float sk[64][33], sv[64][33];
    const int head = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const int ld = 3 * hidden;
    const T *base = qkv + (int64_t)b * hw * ld;
    const int d0 = tid >> 5, e = tid & 31;     // thread owns ctx[d0 + 8*i][e], i = 0..3
    float acc[4] = {0, 0, 0, 0};
    for (int64_t n0 = 0; n0 < hw; n0 += 64) {
        for (int i = tid; i < 64 * 32; i += 256) {
            const int r = i >> 5, c = i & 31;
            const int64_t n = n0 + r;
            float kv = 0.f, vv = 0.f;
            if (n < hw) {
                const int kc = head * 32 + c;
                const float mx = kstats[((int64_t)b * hidden + kc) * 2], sm = kstats[((int64_t)b * hidden + kc) * 2 + 1];
                kv = __expf(ld1(base + n * ld + hidden + kc) - mx) / sm;
                vv = ld1(base + n * ld + 2 * hidden + kc);
            }
            sk[r][c] = kv;
            sv[r][c] = vv;
        }
        __syncthreads();
        for (int r = 0; r < 64; ++r) {
            const float vv = sv[r][e];
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] += sk[r][d0 + 8 * i] * vv;
        }
        __syncthreads();
    }
    const float inv = 1.f / (float)hw;
#pragma unroll
    for (int i = 0; i < 4; ++i)
        ctx[(((int64_t)b * gridDim.x + head) * 32 + d0 + 8 * i) * 32 + e] = acc[i] * inv;
}

// Wtot[b][o][h*32+d] = sum_e Wout[o][h*32+e] * ctx[b,h,d,e]   (context^T q, then to_out conv: one GEMM)
template <typename T>
__global__ void linattn_weff_kernel(const float *__restrict__ ctx, const float *__restrict__ wout, T *__restrict__ wtot,
                                    int C, int hidden) {
    const int b = blockIdx.y;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= C * hidden) return;
    const int o = idx / hidden, hd = idx - o * hidden, head = hd >> 5, d = hd & 31;
    const float *cx = ctx + (((int64_t)b * (hidden / 32) + head) * 32 + d) * 32;
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < 32; ++e) s += wout[(int64_t)o * hidden + head * 32 + e] * cx[e];
    st1(wtot + ((int64_t)b * C + o) * hidden + hd, s);
}

// in place on channels [off, off+heads*32): per pixel, per head softmax over the 32 channels, * scale
template <typename T>
__global__ void softmax_heads_kernel(T *__restrict__ x, int ld, int off, int heads, float scale, int64_t npix) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= npix * heads) return;
    const int64_t pix = i / heads;
    const int h = (int)(i - pix * heads);
    T *p = x + pix * ld + off + h * 32;
    float v[32], mx = -3.4e38f;
#pragma unroll
    for (int c = 0; c < 32; ++c) { v[c] = ld1(p + c); mx = fmaxf(mx, v[c]); }
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < 32; ++c) { v[c] = __expf(v[c] - mx); s += v[c]; }
    const float inv = scale / s;
#pragma unroll
    for (int c = 0; c < 32; ++c) st1(p + c, v[c] * inv);
}

// Softmax attention, dim_head 32 (src/denoising_diffusion_pytorch.py:257-279), flash-style on the matrix cores:
//   * a workgroup owns 128 queries of one (batch, head): 4 waves x 2 blocks of 16 queries; the scaled Q fragments
//     stay in registers for the whole kernel;
//   * K / V stream through LDS in tiles of 64 keys shared by the 4 waves: K as [key][32 d] rows (the A operand of
//     S^T = K Q^T is a plain 16-byte row read), V TRANSPOSED as [d][key] (the A operand of O^T = V^T P^T is two
//     8-byte reads of 4 consecutive keys);
//   * everything is computed transposed -- S^T[key][query] and O^T[d][query] -- so that a lane's accumulator
//     column is ONE query (l % 16) in both products: the online-softmax statistics (running max, sum, rescale)
//     are per-lane scalars, the 4 lane groups of a query are combined with two xor-shuffles, and the
//     exponentiated scores are already laid out as the B operand of the second product (the key order inside a
//     32-key step is the accumulator order, applied to the V^T reads as well): P never moves between lanes;
//   * bf16: v_mfma_f32_16x16x32_bf16, P rounded to bf16, fp32 accumulation and statistics;
//     fp32 (parity mode): v_mfma_f32_16x16x4_f32, exact fp32 products.
// qkv [B, n, 3*hidden]; out [B, n, hidden].
constexpr int AT_KT = 64, AT_QW = 32, AT_QB = AT_QW / 16;           // key tile, queries per wave, 16-query blocks

template <typename T>
__global__ __launch_bounds__(256) void attention_mfma_kernel(const T *__restrict__ qkv, T *__restrict__ out, int hidden,
                                                            int64_t n, float scale) {
    constexpr bool BF = sizeof(T) == 2;
    constexpr int KLD = BF ? 32 : 33;                     // K tile row stride (elements): fp32 rows padded against
    constexpr int VLD = BF ? AT_KT + 8 : AT_KT + 1;       // bank conflicts of the scalar fragment reads
    __shared__ __attribute__((aligned(16))) T sK[AT_KT * KLD];
    __shared__ __attribute__((aligned(16))) T sV[32 * VLD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lq = lane & 15, g = lane >> 4;
    const int head = blockIdx.y, b = blockIdx.z;
    const int ld = 3 * hidden;
    const T *base = qkv + (int64_t)b * n * ld;
    const int64_t q0 = (int64_t)blockIdx.x * (4 * AT_QW) + wave * AT_QW;
    // Q^T fragments (B operand): lane = (query lq, d group g)
    bf16x8 qb16[AT_QB];
    float qf[AT_QB][8];
#pragma unroll
    for (int qb = 0; qb < AT_QB; ++qb) {
        const int64_t qi = min(q0 + qb * 16 + lq, n - 1);
        const T *qp = base + qi * ld + head * 32;
        if constexpr (BF) {
            float v[8];
            load8(qp + 8 * g, v);
#pragma unroll
            for (int e = 0; e < 8; ++e) qb16[qb][e] = fd_cvt_h(v[e] * scale);
        } else {
#pragma unroll
            for (int s = 0; s < 8; ++s) qf[qb][s] = (float)qp[4 * s + g] * scale;      // k-step s: d = 4 s + g
        }
    }
    f32x4 o[AT_QB][2];
    float m[AT_QB], l[AT_QB];
#pragma unroll
    for (int qb = 0; qb < AT_QB; ++qb) {
        o[qb][0] = o[qb][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
        m[qb] = -3.0e38f;
        l[qb] = 0.f;
    }
    for (int64_t k0 = 0; k0 < n; k0 += AT_KT) {
        // ---- stage the tile: thread = (key tid / 4, 8 d values tid % 4); keys past n are zero rows
        {
            const int key = tid >> 2, c = tid & 3;
            float kv[8], vv[8];
            if (k0 + key < n) {
                load8(base + (k0 + key) * ld + hidden + head * 32 + 8 * c, kv);
                load8(base + (k0 + key) * ld + 2 * hidden + head * 32 + 8 * c, vv);
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) kv[e] = vv[e] = 0.f;
            }
            __syncthreads();                               // the previous tile's reads are done
            if constexpr (BF) store8(&sK[key * KLD + 8 * c], kv);
            else {
#pragma unroll
                for (int e = 0; e < 8; ++e) sK[key * KLD + 8 * c + e] = kv[e];
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) sV[(8 * c + e) * VLD + key] = (T)vv[e];
            __syncthreads();
        }
#pragma unroll
        for (int kb = 0; kb < AT_KT / 32; ++kb) {
            // ---- S^T for two 16-key blocks: rows = keys (4 g + i), column = query lq
            f32x4 st[AT_QB][2];
#pragma unroll
            for (int blk = 0; blk < 2; ++blk) {
                const int krow = kb * 32 + blk * 16 + lq;
                if constexpr (BF) {
                    const bf16x8 ka = *(const bf16x8 *)&sK[krow * KLD + 8 * g];
#pragma unroll
                    for (int qb = 0; qb < AT_QB; ++qb)
                        st[qb][blk] = FD_MFMA16(ka, qb16[qb], (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                } else {
#pragma unroll
                    for (int qb = 0; qb < AT_QB; ++qb) st[qb][blk] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int s = 0; s < 8; ++s) {
                        const float ka = sK[krow * KLD + 4 * s + g];
#pragma unroll
                        for (int qb = 0; qb < AT_QB; ++qb)
                            st[qb][blk] = __builtin_amdgcn_mfma_f32_16x16x4f32(ka, qf[qb][s], st[qb][blk], 0, 0, 0);
                    }
                }
            }
            // ---- online softmax per query (= per lane column), keys past n masked
            bf16x8 pb[AT_QB];
            float pf[AT_QB][8];
#pragma unroll
            for (int qb = 0; qb < AT_QB; ++qb) {
                float sv[8], mx = -3.0e38f;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int64_t key = k0 + kb * 32 + (e >> 2) * 16 + 4 * g + (e & 3);
                    sv[e] = key < n ? st[qb][e >> 2][e & 3] : -3.0e38f;
                    mx = fmaxf(mx, sv[e]);
                }
                mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
                mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
                const float mn = fmaxf(m[qb], mx);
                const float corr = __expf(m[qb] - mn);
                float rs = 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float pe = __expf(sv[e] - mn);
                    pf[qb][e] = pe;
                    rs += pe;
                }
                rs += __shfl_xor(rs, 16, 64);
                rs += __shfl_xor(rs, 32, 64);
                l[qb] = l[qb] * corr + rs;
                m[qb] = mn;
#pragma unroll
                for (int dblk = 0; dblk < 2; ++dblk) o[qb][dblk] *= corr;
                if constexpr (BF) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) pb[qb][e] = fd_cvt_h(pf[qb][e]);
                }
            }
            // ---- O^T += V^T P^T: rows = d, k = the 32 keys in accumulator order (e < 4: key 4 g + e; e >= 4: 16 + 4 g + e - 4)
#pragma unroll
            for (int dblk = 0; dblk < 2; ++dblk) {
                const T *vrow = &sV[(dblk * 16 + lq) * VLD + kb * 32 + 4 * g];
                if constexpr (BF) {
                    typedef __attribute__((ext_vector_type(4))) bf16 bf16x4;
                    const bf16x4 v0 = *(const bf16x4 *)vrow, v1 = *(const bf16x4 *)(vrow + 16);
                    const bf16x8 va = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
#pragma unroll
                    for (int qb = 0; qb < AT_QB; ++qb)
                        o[qb][dblk] = FD_MFMA16(va, pb[qb], o[qb][dblk], 0, 0, 0);
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) {          // k-step e: lane group g contributes key (e < 4 ? 4 g + e : 16 + 4 g + e - 4)
                        const float va = vrow[(e >> 2) * 16 + (e & 3)];
#pragma unroll
                        for (int qb = 0; qb < AT_QB; ++qb)
                            o[qb][dblk] = __builtin_amdgcn_mfma_f32_16x16x4f32(va, pf[qb][e], o[qb][dblk], 0, 0, 0);
                    }
                }
            }
        }
    }
    // ---- O^T[d = dblk*16 + 4 g + i][query lq] / l -> out[query][head*32 + d]: 4 consecutive d per lane
#pragma unroll
    for (int qb = 0; qb < AT_QB; ++qb) {
        const int64_t qi = q0 + qb * 16 + lq;
        if (qi >= n) continue;
        const float inv = 1.f / l[qb];
#pragma unroll
        for (int dblk = 0; dblk < 2; ++dblk) {
            T *op = out + ((int64_t)b * n + qi) * hidden + head * 32 + dblk * 16 + 4 * g;
#pragma unroll
            for (int i = 0; i < 4; ++i) st1(op + i, o[qb][dblk][i] * inv);
        }
    }
}

__global__ void lincomb3_kernel(const float *a, const float *b, const float *c, float ca, float cb, float cc, int clamp,
                                float *out, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float v = ca * a[i];
        if (b) v += cb * b[i];
        if (c) v += cc * c[i];
        if (clamp) v = fd_clamp1(v);
        out[i] = v;
    }
}

unsigned g1v(int64_t n) {
    int64_t g = (n + 255) / 256;
    return (unsigned)(g > 4096 ? 4096 : (g < 1 ? 1 : g));
}

}  // namespace

#define FD_DISPATCH_T(dtype, KERNEL, grid, block, lds, stream, ...)                                      \
    do {                                                                                                 \
        if ((dtype) == FD_BF16) hipLaunchKernelGGL(KERNEL<bf16>, grid, block, lds, (hipStream_t)stream, __VA_ARGS__); \
        else hipLaunchKernelGGL(KERNEL<float>, grid, block, lds, (hipStream_t)stream, __VA_ARGS__);      \
    } while (0)

extern "C" int fd_gn_film_silu_apply(int dtype, const void *h, const float *mean_rstd, const float *gamma,
                                     const float *beta, const float *film_scale, const float *film_shift, int film_ld,
                                     void *out, int B, int64_t hw, int C, int groups, void *stream) {
    FD_REQUIRE(h && out && mean_rstd && gamma && beta && C % 8 == 0 && C % groups == 0, "fd_gn_film_silu_apply: bad args");
    FD_REQUIRE((film_scale == nullptr) == (film_shift == nullptr), "fd_gn_film_silu_apply: scale/shift both or none");
    const int64_t nvec = hw * C / 8;
    dim3 grid(g1v(nvec), B), block(256);
    if (dtype == FD_BF16)
        hipLaunchKernelGGL(gn_film_silu_kernel<bf16>, grid, block, 0, (hipStream_t)stream, (const bf16 *)h, mean_rstd, gamma,
                           beta, film_scale, film_shift, film_ld, (bf16 *)out, hw, C, groups, nvec);
    else
        hipLaunchKernelGGL(gn_film_silu_kernel<float>, grid, block, 0, (hipStream_t)stream, (const float *)h, mean_rstd,
                           gamma, beta, film_scale, film_shift, film_ld, (float *)out, hw, C, groups, nvec);
    FD_LAUNCH_OK("fd_gn_film_silu_apply");
    return FD_OK;
}

extern "C" int fd_chan_ln(int dtype, const void *x, const float *g, const void *res, void *out, int64_t nrows, int C,
                          void *stream) {
    FD_REQUIRE(x && g && out, "fd_chan_ln: null pointer");
    dim3 grid((unsigned)((nrows + 3) / 4)), block(256);
    if (dtype == FD_BF16)
        hipLaunchKernelGGL(chan_ln_kernel<bf16>, grid, block, 0, (hipStream_t)stream, (const bf16 *)x, g, (const bf16 *)res,
                           (bf16 *)out, C, nrows);
    else
        hipLaunchKernelGGL(chan_ln_kernel<float>, grid, block, 0, (hipStream_t)stream, (const float *)x, g,
                           (const float *)res, (float *)out, C, nrows);
    FD_LAUNCH_OK("fd_chan_ln");
    return FD_OK;
}

extern "C" int fd_linear_attention(int dtype, void *qkv, int B, int64_t hw, int hidden, const float *wout,
                                   float *kstats, float *ctx, void *wtot, int C, void *stream) {
    FD_REQUIRE(qkv && wout && kstats && ctx && wtot && hidden % 32 == 0, "fd_linear_attention: bad args");
    const int heads = hidden / 32;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == FD_BF16) {
        hipLaunchKernelGGL(col_softmax_stats_kernel<bf16>, dim3(hidden, B), dim3(256), 0, s, (const bf16 *)qkv, 3 * hidden,
                           hidden, hw, hidden, kstats);
        hipLaunchKernelGGL(linattn_context_kernel<bf16>, dim3(heads, B), dim3(256), 0, s, (const bf16 *)qkv, hidden, hw,
                           kstats, ctx);
        hipLaunchKernelGGL(linattn_weff_kernel<bf16>, dim3((C * hidden + 255) / 256, B), dim3(256), 0, s, ctx, wout,
                           (bf16 *)wtot, C, hidden);
        hipLaunchKernelGGL(softmax_heads_kernel<bf16>, dim3((unsigned)((B * hw * heads + 255) / 256)), dim3(256), 0, s,
                           (bf16 *)qkv, 3 * hidden, 0, heads, 0.17677669529663687f, (int64_t)B * hw);
    } else {
        hipLaunchKernelGGL(col_softmax_stats_kernel<float>, dim3(hidden, B), dim3(256), 0, s, (const float *)qkv,
                           3 * hidden, hidden, hw, hidden, kstats);
        hipLaunchKernelGGL(linattn_context_kernel<float>, dim3(heads, B), dim3(256), 0, s, (const float *)qkv, hidden, hw,
                           kstats, ctx);
        hipLaunchKernelGGL(linattn_weff_kernel<float>, dim3((C * hidden + 255) / 256, B), dim3(256), 0, s, ctx, wout,
                           (float *)wtot, C, hidden);
        hipLaunchKernelGGL(softmax_heads_kernel<float>, dim3((unsigned)((B * hw * heads + 255) / 256)), dim3(256), 0, s,
                           (float *)qkv, 3 * hidden, 0, heads, 0.17677669529663687f, (int64_t)B * hw);
    }
    FD_LAUNCH_OK("fd_linear_attention");
    return FD_OK;
}

extern "C" int fd_attention(int dtype, const void *qkv, void *out, int B, int64_t n, int hidden, void *stream) {
    FD_REQUIRE(qkv && out && hidden % 32 == 0 && n > 0, "fd_attention: bad args");
    FD_REQUIRE(hidden % 8 == 0, "fd_attention: hidden must be a multiple of 8");
    dim3 grid((unsigned)((n + 4 * AT_QW - 1) / (4 * AT_QW)), hidden / 32, B), block(256);
    if (dtype == FD_BF16)
        hipLaunchKernelGGL(attention_mfma_kernel<bf16>, grid, block, 0, (hipStream_t)stream, (const bf16 *)qkv, (bf16 *)out,
                           hidden, n, 0.17677669529663687f);
    else
        hipLaunchKernelGGL(attention_mfma_kernel<float>, grid, block, 0, (hipStream_t)stream, (const float *)qkv,
                           (float *)out, hidden, n, 0.17677669529663687f);
    FD_LAUNCH_OK("fd_attention");
    return FD_OK;
}

extern "C" int fd_lincomb3(const float *a, const float *b, const float *c, float ca, float cb, float cc, int clamp,
                           float *out, int64_t n, void *stream) {
    FD_REQUIRE(a && out, "fd_lincomb3: null pointer");
    hipLaunchKernelGGL(lincomb3_kernel, dim3(g1v(n)), dim3(256), 0, (hipStream_t)stream, a, b, c, ca, cb, cc, clamp, out, n);
    FD_LAUNCH_OK("fd_lincomb3");
    return FD_OK;
}
