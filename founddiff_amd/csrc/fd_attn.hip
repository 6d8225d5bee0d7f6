// fd_attn.hip -- Restormer-style channel ("transposed") attention for gfx950.
//
// The attention matrix is (C/heads x C/heads) = 32x32 per head with the reduction over ALL
// pixels, so the work is a tall-skinny Gram product q^T k (split over pixel blocks) followed
// by a tiny softmax.  attn @ v and the 1x1 project_out are both linear per pixel, so they are
// folded into one per-batch CxC matrix Weff and applied as a single GEMM over v (fd_conv2d):
// the (B,heads,32,HW) attention output never exists in HBM.
// The Gram uses the exact-f32 MFMA (v_mfma_f32_16x16x4_f32): its A/B operands are one f32 per
// lane indexed [channel][pixel], which is exactly a row-major read of a pixel-major LDS tile,
// so no transpose is needed; in bf16 mode q/k are widened while staging.
#include "fd_common.h"

namespace {

// pixels per Gram block: 1024 for big images, 256 for small ones so the low-resolution levels
// still spread over the chip (a function of the image size only -> batch-invariant results)
__host__ __device__ inline int gram_pb(int64_t hw) { return hw >= 65536 ? 1024 : 256; }
constexpr int LDP = 48;     // LDS row stride in floats (32 channels + pad: conflict-free b32 reads)

template <typename T>
__global__ __launch_bounds__(256) void gram_kernel(const T *__restrict__ qkv, int64_t hw, int C,
                                                  float *__restrict__ partial, int nblk) {
    __shared__ __attribute__((aligned(16))) float sQ[64 * LDP], sK[64 * LDP];
    __shared__ float sN[4][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int blk = blockIdx.x, head = blockIdx.y, b = blockIdx.z, heads = gridDim.y;
    const int pr = tid >> 2, cv = tid & 3;
    const int PB = gram_pb(hw);
    const int64_t p0 = (int64_t)blk * PB;
    const int64_t p1 = min(p0 + PB, hw);
    const T *base = qkv + (int64_t)b * hw * 3 * C;
    const int i0 = 16 * (wave >> 1), j0 = 16 * (wave & 1);
    const int fr = lane & 15, fg = lane >> 4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    float sq[8], sk[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) sq[e] = sk[e] = 0.f;
    if constexpr (sizeof(T) == 2) {
        // bf16: q, k are bf16 in HBM, so the bf16 MFMA is exact too (bf16 x bf16 products are exact in fp32) and
        // contracts 32 pixels per instruction instead of 4 -- 2 MFMAs per 64-pixel step instead of 16 (PMC: the f32
        // form kept the matrix pipe 37 % busy on a kernel that should be a pure q/k stream).  Its operands want 8
        // consecutive PIXELS of one channel per lane: the tiles are staged transposed ([channel][pixel], 2-byte
        // writes from the pixel-major loads).  The next step's loads are in flight during this step's LDS + MFMA.
        constexpr int TLD = 64 + 8;                  // pixels per row + pad (bf16 elements)
        bf16 *tQ = (bf16 *)sQ, *tK = (bf16 *)sK;     // [32][TLD] each (4.6 KB of the 12 KB arrays)
        u32x4 qn = {0, 0, 0, 0}, kn = {0, 0, 0, 0};
        auto fetch = [&](int64_t pt) {
            const int64_t p = pt + pr;
            qn = kn = (u32x4){0, 0, 0, 0};
            if (p < p1) {
                qn = *(const u32x4 *)(base + p * 3 * C + head * 32 + cv * 8);
                kn = *(const u32x4 *)(base + p * 3 * C + C + head * 32 + cv * 8);
            }
        };
        fetch(p0);
        for (int64_t pt = p0; pt < p1; pt += 64) {
            const bf16x8 q8 = __builtin_bit_cast(bf16x8, qn), k8 = __builtin_bit_cast(bf16x8, kn);
            if (pt + 64 < p1) fetch(pt + 64);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float qf = (float)q8[e], kf = (float)k8[e];
                sq[e] += qf * qf;
                sk[e] += kf * kf;
                tQ[(cv * 8 + e) * TLD + pr] = q8[e];
                tK[(cv * 8 + e) * TLD + pr] = k8[e];
            }
            __syncthreads();
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const bf16x8 a = *(const bf16x8 *)&tQ[(i0 + fr) * TLD + 32 * s + 8 * fg];
                const bf16x8 bb = *(const bf16x8 *)&tK[(j0 + fr) * TLD + 32 * s + 8 * fg];
                acc = FD_MFMA16(a, bb, acc, 0, 0, 0);
            }
            __syncthreads();
        }
    } else {
    for (int64_t pt = p0; pt < p1; pt += 64) {
        const int64_t p = pt + pr;
        float q8[8], k8[8];
        if (p < p1) {
            load8(base + p * 3 * C + head * 32 + cv * 8, q8);
            load8(base + p * 3 * C + C + head * 32 + cv * 8, k8);
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) q8[e] = k8[e] = 0.f;
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            sq[e] += q8[e] * q8[e];
            sk[e] += k8[e] * k8[e];
            sQ[pr * LDP + cv * 8 + e] = q8[e];
            sK[pr * LDP + cv * 8 + e] = k8[e];
        }
        __syncthreads();
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const float a = sQ[(4 * s + fg) * LDP + i0 + fr];
            const float bb = sK[(4 * s + fg) * LDP + j0 + fr];
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bb, acc, 0, 0, 0);
        }
        __syncthreads();
    }
    }
    float *out = partial + (((int64_t)b * heads + head) * nblk + blk) * (1024 + 64);
#pragma unroll
    for (int e = 0; e < 4; ++e) out[(i0 + fg * 4 + e) * 32 + j0 + fr] = acc[e];
    // sums of squares: lanes with equal (tid & 3) own the same 8 channels
#pragma unroll
    for (int e = 0; e < 8; ++e) {
#pragma unroll
        for (int o = 4; o < 64; o <<= 1) {
            sq[e] += __shfl_xor(sq[e], o, 64);
            sk[e] += __shfl_xor(sk[e], o, 64);
        }
    }
    if (lane < 4) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            sN[wave][lane * 8 + e] = sq[e];
            sN[wave][32 + lane * 8 + e] = sk[e];
        }
    }
    __syncthreads();
    if (tid < 64) out[1024 + tid] = sN[0][tid] + sN[1][tid] + sN[2][tid] + sN[3][tid];
}

// Sum the pixel-block partials of one (batch, head) in place into block 0's slot: a workgroup
// owns 64 of the 1088 entries, 4 thread groups split the blocks, fixed order -> deterministic.
__global__ __launch_bounds__(256) void gram_reduce_kernel(float *__restrict__ partial, int nblk) {
    __shared__ float sh[4][64];
    const int col = blockIdx.x * 64 + (threadIdx.x & 63), grp = threadIdx.x >> 6;
    float *pp = partial + (int64_t)blockIdx.y * nblk * (1024 + 64);
    float s = 0.f;
    if (col < 1024 + 64) {
        int k = grp;
        for (; k + 12 < nblk; k += 16) {       // 4 independent loads in flight per thread
            const float a0 = pp[(int64_t)k * 1088 + col], a1 = pp[(int64_t)(k + 4) * 1088 + col];
            const float a2 = pp[(int64_t)(k + 8) * 1088 + col], a3 = pp[(int64_t)(k + 12) * 1088 + col];
            s += (a0 + a1) + (a2 + a3);
        }
        for (; k < nblk; k += 4) s += pp[(int64_t)k * 1088 + col];
    }
    sh[grp][threadIdx.x & 63] = s;
    __syncthreads();
    if (grp == 0 && col < 1024 + 64) pp[col] = (sh[0][threadIdx.x] + sh[1][threadIdx.x]) + (sh[2][threadIdx.x] + sh[3][threadIdx.x]);
}

template <typename T>
__global__ __launch_bounds__(256) void weff_kernel(const float *__restrict__ partial, int nblk,
                                                  const float *__restrict__ temperature,
                                                  const float *__restrict__ wproj, T *__restrict__ weff, int C) {
    __shared__ float sG[32 * 33];
    __shared__ float sNrm[64];
    const int tid = threadIdx.x;
    const int head = blockIdx.x, b = blockIdx.y, heads = gridDim.x;
    const float *pp = partial + ((int64_t)b * heads + head) * nblk * (1024 + 64);   // slot 0 = reduced
    for (int i = tid; i < 1024 + 64; i += 256) {
        const float s = pp[i];
        if (i < 1024) sG[(i >> 5) * 33 + (i & 31)] = s;
        else sNrm[i - 1024] = fmaxf(sqrtf(s), 1e-12f);     // F.normalize eps
    }
    __syncthreads();
    const float temp = temperature[head];
    if (tid < 32) {   // one softmax row per thread (32x32: tiny)
        float row[32], mx = -3.4e38f;
#pragma unroll
        for (int j = 0; j < 32; ++j) {
            row[j] = sG[tid * 33 + j] / (sNrm[tid] * sNrm[32 + j]) * temp;
            mx = fmaxf(mx, row[j]);
        }
        float den = 0.f;
#pragma unroll
        for (int j = 0; j < 32; ++j) { row[j] = __expf(row[j] - mx); den += row[j]; }
        const float inv = 1.f / den;
#pragma unroll
        for (int j = 0; j < 32; ++j) sG[tid * 33 + j] = row[j] * inv;
    }
    __syncthreads();
    // 64 output rows per workgroup (blockIdx.z): C/64 workgroups per (head, batch)
    const int o0 = blockIdx.z * 64, o1 = min(o0 + 64, C);
    for (int idx = o0 * 32 + tid; idx < o1 * 32; idx += 256) {
        const int o = idx >> 5, j = idx & 31;
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 32; ++i) s += wproj[(int64_t)o * C + head * 32 + i] * sG[i * 33 + j];
        st1(weff + ((int64_t)b * C + o) * C + head * 32 + j, s);
    }
}

// one query per (batch, head) against T keys: DA-CLIP attention pool (token 0 only)
__global__ __launch_bounds__(256) void attnpool_core_kernel(const float *__restrict__ q, int q_ld, const float *__restrict__ kv,
                                                           int ld, int koff, int voff, float *__restrict__ out,
                                                           int T, int C, int heads) {
    extern __shared__ float sS[];   // T scores
    __shared__ float red[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int head = blockIdx.x, b = blockIdx.y;
    const int dh = C / heads;
    const float scale = rsqrtf((float)dh);
    const float *qh = q + (int64_t)b * q_ld + head * dh;
    float mx = -3.4e38f;
    for (int t = tid; t < T; t += 256) {
        const float *kt = kv + ((int64_t)b * T + t) * ld + koff + head * dh;
        float s = 0.f;
        for (int c = 0; c < dh; ++c) s += qh[c] * scale * kt[c];
        sS[t] = s;
        mx = fmaxf(mx, s);
    }
    mx = wave_max(mx);
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float den = 0.f;
    for (int t = tid; t < T; t += 256) {
        float e = __expf(sS[t] - mx);
        sS[t] = e;
        den += e;
    }
    den = wave_sum(den);
    if (lane == 0) red[wave] = den;
    __syncthreads();
    den = red[0] + red[1] + red[2] + red[3];
    for (int c = tid; c < dh; c += 256) {
        float s = 0.f;
        for (int t = 0; t < T; ++t) s += sS[t] * kv[((int64_t)b * T + t) * ld + voff + head * dh + c];
        out[(int64_t)b * C + head * dh + c] = s / den;
    }
}

}  // namespace

extern "C" int fd_chan_attn_nblk(int64_t hw) { return (int)((hw + gram_pb(hw) - 1) / gram_pb(hw)); }

extern "C" int fd_chan_attn_gram(int dtype, const void *qkv, int B, int64_t hw, int C, float *partial, void *stream) {
    FD_REQUIRE(qkv && partial, "fd_chan_attn_gram: null pointer");
    FD_REQUIRE(C % 32 == 0, "fd_chan_attn_gram: C=%d must be a multiple of 32 (heads = C/32)", C);
    const int nblk = fd_chan_attn_nblk(hw);
    dim3 grid(nblk, C / 32, B), block(256);
    if (dtype == FD_BF16)
        hipLaunchKernelGGL(gram_kernel<bf16>, grid, block, 0, (hipStream_t)stream, (const bf16 *)qkv, hw, C, partial, nblk);
    else
        hipLaunchKernelGGL(gram_kernel<float>, grid, block, 0, (hipStream_t)stream, (const float *)qkv, hw, C, partial, nblk);
    FD_LAUNCH_OK("fd_chan_attn_gram");
    return FD_OK;
}

extern "C" int fd_chan_attn_weff(int dtype, float *partial, int nblk, const float *temperature,
                                 const float *wproj, void *weff, int B, int C, void *stream) {
    FD_REQUIRE(partial && temperature && wproj && weff && C % 32 == 0, "fd_chan_attn_weff: bad args");
    dim3 grid(C / 32, B, (C + 63) / 64), block(256);
    if (nblk > 1)
        hipLaunchKernelGGL(gram_reduce_kernel, dim3(17, B * (C / 32)), dim3(256), 0, (hipStream_t)stream, partial, nblk);
    if (dtype == FD_BF16)
        hipLaunchKernelGGL(weff_kernel<bf16>, grid, block, 0, (hipStream_t)stream, partial, nblk, temperature, wproj, (bf16 *)weff, C);
    else
        hipLaunchKernelGGL(weff_kernel<float>, grid, block, 0, (hipStream_t)stream, partial, nblk, temperature, wproj, (float *)weff, C);
    FD_LAUNCH_OK("fd_chan_attn_weff");
    return FD_OK;
}

extern "C" int fd_attnpool_core(const float *q, int q_ld, const float *kv, int ld, int koff, int voff, float *out, int B,
                                int T, int C, int heads, void *stream) {
    FD_REQUIRE(q && kv && out && heads > 0 && C % heads == 0, "fd_attnpool_core: bad args");
    hipLaunchKernelGGL(attnpool_core_kernel, dim3(heads, B), dim3(256), T * sizeof(float), (hipStream_t)stream, q, q_ld, kv,
                       ld, koff, voff, out, T, C, heads);
    FD_LAUNCH_OK("fd_attnpool_core");
    return FD_OK;
}
