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

// Softmax attention, dim_head 32: one lane per query, keys/values streamed wave-uniformly with an
// online softmax (fp32).  qkv [B, n, 3*hidden]; out [B, n, hidden].
template <typename T>
__global__ __launch_bounds__(64) void attention_rows_kernel(const T *__restrict__ qkv, T *__restrict__ out, int hidden,
                                                           int64_t n, float scale) {
    const int head = blockIdx.y, b = blockIdx.z;
    const int64_t qi = (int64_t)blockIdx.x * 64 + threadIdx.x;
    const int ld = 3 * hidden;
    const T *base = qkv + (int64_t)b * n * ld;
    float q[32], o[32];
    const bool ok = qi < n;
#pragma unroll
    for (int c = 0; c < 32; ++c) { q[c] = ok ? ld1(base + qi * ld + head * 32 + c) * scale : 0.f; o[c] = 0.f; }
    float m = -3.4e38f, l = 0.f;
    for (int64_t j = 0; j < n; ++j) {
        const T *kp = base + j * ld + hidden + head * 32;
        const T *vp = base + j * ld + 2 * hidden + head * 32;
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < 32; ++c) s += q[c] * ld1(kp + c);
        const float mn = fmaxf(m, s);
        const float corr = __expf(m - mn), pj = __expf(s - mn);
        l = l * corr + pj;
#pragma unroll
        for (int c = 0; c < 32; ++c) o[c] = o[c] * corr + pj * ld1(vp + c);
        m = mn;
    }
    if (ok) {
        const float inv = 1.f / l;
#pragma unroll
        for (int c = 0; c < 32; ++c) st1(out + ((int64_t)b * n + qi) * hidden + head * 32 + c, o[c] * inv);
    }
}

__global__ void lincomb3_kernel(const float *a, const float *b, const float *c, float ca, float cb, float cc, int clamp,
                                float *out, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float v = ca * a[i];
        if (b) v += cb * b[i];
        if (c) v += cc * c[i];
        if (clamp) v = fminf(fmaxf(v, -1.f), 1.f);
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
    FD_REQUIRE(qkv && out && hidden % 32 == 0, "fd_attention: bad args");
    dim3 grid((unsigned)((n + 63) / 64), hidden / 32, B), block(64);
    if (dtype == FD_BF16)
        hipLaunchKernelGGL(attention_rows_kernel<bf16>, grid, block, 0, (hipStream_t)stream, (const bf16 *)qkv, (bf16 *)out,
                           hidden, n, 0.17677669529663687f);
    else
        hipLaunchKernelGGL(attention_rows_kernel<float>, grid, block, 0, (hipStream_t)stream, (const float *)qkv,
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
