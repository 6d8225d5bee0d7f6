// fd_rows.hip -- HBM-bound row kernels on NHWC activations: LayerNorm+adaLN modulate,
// out_norm * z + local, depthwise 3x3, avg-pool, plane packing, final 1-channel conv.
// All are one pass over their operands with 16-byte (bf16) / 32-byte (f32) per-lane vectors.
#include "fd_common.h"

namespace {

// Row of C channels handled by LPR lanes x VPL 8-element vectors per lane (C = 8*VPL*LPR),
// LPR a power of two <= 64 so the reduction is an xor-shuffle butterfly inside one wave.
template <typename T, int VPL, bool GATE, bool VEC>
__global__ __launch_bounds__(256) void ln_rows_kernel(
    const T *__restrict__ x, const float *__restrict__ gamma, const float *__restrict__ beta, float eps,
    const float *__restrict__ shift, const float *__restrict__ scale, int mod_ld,
    const T *__restrict__ z, int ldz, int offz, T *__restrict__ out, int64_t hw, int C, int lpr,
    int64_t nrows) {
    const int tid = threadIdx.x;
    const int rpb = 256 / lpr;  // rows per block
    const int sub = tid % lpr;
    const int64_t row = (int64_t)blockIdx.x * rpb + tid / lpr;
    const bool active = row < nrows;
    const int64_t rr = active ? row : 0;
    const int b = (int)(rr / hw);
    // packed fp32 throughout (v_pk_add / v_pk_fma: two channels per issue slot) and DPP row sums for the first four
    // butterfly steps: this kernel lives on row parallelism and ran at 97 % VALU busy (PMC, round 3)
    auto allsum = [&](float t) -> float {
        if (lpr >= 2) t = fd_dpp_add<0xB1>(t);
        if (lpr >= 4) t = fd_dpp_add<0x4E>(t);
        if (lpr >= 8) t = fd_dpp_add<0x141>(t);
        if (lpr >= 16) t = fd_dpp_add<0x140>(t);
        if (lpr >= 32) t += __shfl_xor(t, 16, 64);
        if (lpr >= 64) t += __shfl_xor(t, 32, 64);
        return t;
    };
    f32x2 v[VPL][4];
    f32x2 s2 = {0.f, 0.f};
#pragma unroll
    for (int j = 0; j < VPL; ++j) {
        float t8[8];
        load8(x + rr * C + (j * lpr + sub) * 8, t8);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            v[j][e] = f32x2{t8[2 * e], t8[2 * e + 1]};
            s2 += v[j][e];
        }
    }
    const float mean = allsum(s2.x + s2.y) / C;
    f32x2 q2 = {0.f, 0.f};
#pragma unroll
    for (int j = 0; j < VPL; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const f32x2 d = v[j][e] - mean;
            q2 = d * d + q2;
            v[j][e] = d;                    // the apply below works on the centred value: (x - mean) * rstd, not x * rstd - mean * rstd
                                            // (a row with |mean| >> std would lose the difference's low bits in the second form)
        }
    const float rstd = rsqrtf(allsum(q2.x + q2.y) / C + eps);
    if (!active) return;
#pragma unroll
    for (int j = 0; j < VPL; ++j) {
        const int c0 = (j * lpr + sub) * 8;
        float o8[8], zz[8], gg[8], bb[8], sh[8], sc[8];
        if (GATE) load8(z + rr * ldz + offz + c0, zz);
        if (VEC) {   // all per-channel vectors 16-byte aligned: 2 x dwordx4 each instead of 8 dword loads
            if (gamma) { load8(gamma + c0, gg); load8(beta + c0, bb); }
            load8(shift + (int64_t)b * mod_ld + c0, sh);
            if (!GATE) load8(scale + (int64_t)b * mod_ld + c0, sc);
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                if (gamma) { gg[e] = gamma[c0 + e]; bb[e] = beta[c0 + e]; }
                sh[e] = shift[(int64_t)b * mod_ld + c0 + e];
                if (!GATE) sc[e] = scale[(int64_t)b * mod_ld + c0 + e];
            }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            f32x2 y = v[j][e] * rstd;
            if (gamma) y = y * f32x2{gg[2 * e], gg[2 * e + 1]} + f32x2{bb[2 * e], bb[2 * e + 1]};
            const f32x2 sh2 = {sh[2 * e], sh[2 * e + 1]};
            if (GATE) y = y * f32x2{zz[2 * e], zz[2 * e + 1]} + sh2;       // shift == local
            else y = y * (f32x2{sc[2 * e], sc[2 * e + 1]} + 1.f) + sh2;
            o8[2 * e] = y.x;
            o8[2 * e + 1] = y.y;
        }
        store8(out + rr * C + c0, o8);
    }
}

template <typename T, bool GATE>
int launch_ln(const T *x, const float *gamma, const float *beta, float eps, const float *shift,
              const float *scale, int mod_ld, const T *z, int ldz, int offz, T *out, int B, int64_t hw,
              int C, hipStream_t s) {
    int vpl = (C + 511) / 512;
    int lpr = C / (8 * vpl);
    if (lpr * 8 * vpl != C || (lpr & (lpr - 1)) || lpr > 64 || vpl > 2) {
        fd_set_error("fd_ln_*: C=%d unsupported (need C = 8*v*2^k, v<=2, 2^k<=64)", C);
        return FD_ERR_ARG;
    }
    int64_t nrows = (int64_t)B * hw;
    int rpb = 256 / lpr;
    dim3 grid((unsigned)((nrows + rpb - 1) / rpb)), block(256);
    auto al16 = [](const void *p) { return ((uintptr_t)p & 15) == 0; };
    const bool vec = al16(gamma) && al16(beta) && al16(shift) && al16(scale) && (mod_ld % 4 == 0);
#define FD_LN_LAUNCH(V, VEC_)                                                                              \
    hipLaunchKernelGGL((ln_rows_kernel<T, V, GATE, VEC_>), grid, block, 0, s, x, gamma, beta, eps, shift, \
                       scale, mod_ld, z, ldz, offz, out, hw, C, lpr, nrows)
    if (vpl == 1) { if (vec) FD_LN_LAUNCH(1, true); else FD_LN_LAUNCH(1, false); }
    else { if (vec) FD_LN_LAUNCH(2, true); else FD_LN_LAUNCH(2, false); }
#undef FD_LN_LAUNCH
    return FD_OK;
}

// depthwise 3x3, pad 1, LDS-tiled: a workgroup owns an 8 x 16 pixel tile x 64 channels.  The
// (8+2) x (16+2) halo tile is read from HBM once (coalesced 128-byte pixel rows) into LDS; a
// thread owns one 8-channel vector of one tile column, keeps its 9 tap vectors in registers and
// walks down the 8 rows reading 9 LDS vectors per output (a wave reads 1 KiB contiguous: no
// bank conflicts).  HBM traffic = 1.33x input (halo) + output instead of relying on L1/L2 for
// the 9x tap reuse.
constexpr int DW_TY = 8, DW_TX = 16, DW_CB = 64;
template <typename T>
__global__ __launch_bounds__(256) void dwconv3x3_kernel(const T *__restrict__ in, int ld_in, int off_in,
                                                       const float *__restrict__ w, const float *__restrict__ bias,
                                                       int silu, T *__restrict__ out, int ld_out, int off_out,
                                                       int H, int W, int C, int cblocks) {
    extern __shared__ __attribute__((aligned(16))) unsigned char dw_smem[];
    T *tile = (T *)dw_smem;                          // [(TY+2)][(TX+2)][CB]
    const int tid = threadIdx.x;
    // channel block fastest: the workgroups that read the other 128-byte pieces of the same pixel rows
    // are dispatched back to back (same DRAM pages, same L2 lines for the halo)
    const int cb = blockIdx.x % cblocks;
    const int64_t img = blockIdx.z;
    const int x0 = (blockIdx.x / cblocks) * DW_TX, y0 = blockIdx.y * DW_TY;
    const int cbase = cb * DW_CB;
    constexpr int HX = DW_TX + 2, HY = DW_TY + 2;
    for (int idx = tid; idx < HY * HX * 8; idx += 256) {
        const int v = idx & 7, pxl = idx >> 3;
        const int hy = pxl / HX, hx = pxl - hy * HX;
        const int yy = y0 + hy - 1, xx = x0 + hx - 1, c0 = cbase + v * 8;
        u32x4 val = {0, 0, 0, 0};
        if (yy >= 0 && yy < H && xx >= 0 && xx < W && c0 < C) {
            const T *src = in + ((img * H + yy) * W + xx) * ld_in + off_in + c0;
            if constexpr (sizeof(T) == 2) val = *(const u32x4 *)src;
            else {
                *(u32x4 *)(tile + (int64_t)pxl * DW_CB + v * 8) = *(const u32x4 *)src;
                val = *(const u32x4 *)(src + 4);
                *(u32x4 *)(tile + (int64_t)pxl * DW_CB + v * 8 + 4) = val;
                continue;
            }
        } else if constexpr (sizeof(T) == 4) {
            *(u32x4 *)(tile + (int64_t)pxl * DW_CB + v * 8) = val;
            *(u32x4 *)(tile + (int64_t)pxl * DW_CB + v * 8 + 4) = val;
            continue;
        }
        *(u32x4 *)(tile + (int64_t)pxl * DW_CB + v * 8) = val;
    }
    const int cv = tid & 7, px = (tid >> 3) & 15, rhalf = tid >> 7;   // 2 row halves x 16 columns x 8 vectors
    const int c0 = cbase + cv * 8;
    float wt[9][8], bs[8];
    const bool cok = c0 < C;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        if (cok) load8(w + t * C + c0, wt[t]);
        else {
#pragma unroll
            for (int e = 0; e < 8; ++e) wt[t][e] = 0.f;
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) bs[e] = (bias && cok) ? bias[c0 + e] : 0.f;
    __syncthreads();
    const int x = x0 + px;
    if (x >= W || !cok) return;
#pragma unroll
    for (int rr = 0; rr < DW_TY / 2; ++rr) {
        const int r = rhalf * (DW_TY / 2) + rr;
        const int y = y0 + r;
        if (y >= H) break;
        float acc[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = bs[e];
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                float v[8];
                load8(tile + ((r + dy) * HX + px + dx) * DW_CB + cv * 8, v);
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] += v[e] * wt[dy * 3 + dx][e];
            }
        if (silu) fd_silu8(acc);
        store8(out + ((img * H + y) * W + x) * ld_out + off_out + c0, acc);
    }
}

// bf16 specialisation (the production mode).  Two changes against the generic kernel, both aimed at
// its measured co-bottlenecks (VALU 53 % busy, LDS reads ~55 % of the run time):
//  * 16 x 16 pixel tile, a thread walks 8 rows of one column keeping a 3-row x 3-tap window of packed
//    bf16 pairs in registers: 30 ds_read_b128 per 8 outputs instead of 72;
//  * each multiply-accumulate is ONE v_dot2c_f32_bf16 on the packed pair register with a weight
//    register that holds the tap weight in the wanted half and 0 in the other -- no bf16->f32 unpack.
//    The tap weights are therefore rounded to bf16 (as every dense conv weight in this mode is);
//    accumulation stays f32.
constexpr int DWB_T = 16;
__global__ __launch_bounds__(256) void dwconv3x3_bf16_kernel(const bf16 *__restrict__ in, int ld_in, int off_in,
                                                            const float *__restrict__ w,
                                                            const float *__restrict__ bias, int silu,
                                                            bf16 *__restrict__ out, int ld_out, int off_out, int H,
                                                            int W, int C, int cblocks) {
    extern __shared__ __attribute__((aligned(16))) unsigned char dw_smem[];
    bf16 *tile = (bf16 *)dw_smem;                    // [18][18][64]
    const int tid = threadIdx.x;
    const int cb = blockIdx.x % cblocks;
    const int64_t img = blockIdx.z;
    const int x0 = (blockIdx.x / cblocks) * DWB_T, y0 = blockIdx.y * DWB_T;
    const int cbase = cb * DW_CB;
    const bf16 *in_img = in + img * H * W * ld_in + off_in;
    bf16 *out_img = out + img * H * W * ld_out + off_out;
    constexpr int HX = DWB_T + 2, HY = DWB_T + 2;
    // all halo loads of the thread are issued before the first LDS write (one HBM round trip per
    // workgroup instead of one per loop iteration)
    constexpr int NLD = (HY * HX * 8 + 255) / 256;
    u32x4 vals[NLD];
    uint32_t okm = 0;
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
        const int idx = tid + k * 256;
        const int v = idx & 7, pxl = idx >> 3;
        const int hy = pxl / HX, hx = pxl - hy * HX;
        const int yy = y0 + hy - 1, xx = x0 + hx - 1, c0 = cbase + v * 8;
        // branch-free: always load from a clamped in-bounds address, zero the padding on the LDS write
        const bool ok = yy >= 0 && yy < H && xx >= 0 && xx < W && c0 < C;
        okm |= (ok ? 1u : 0u) << k;
        const int yc = min(max(yy, 0), H - 1), xc = min(max(xx, 0), W - 1), cc = min(c0, C - 8);
        // 32-bit element offset from the image's (wave-uniform) base with 24-bit multiplies: the 64-bit form cost
        // seven quarter-rate integer instructions per load in a kernel that is 80 % VALU-busy (launcher: H*W < 2^24)
        vals[k] = *(const u32x4 *)(in_img + (__umul24(__umul24(yc, W) + xc, ld_in) + cc));
    }
    const int cv = tid & 7, px = (tid >> 3) & 15, rhalf = tid >> 7;
    const int c0 = cbase + cv * 8;
    const bool cok = c0 < C;
    const int cw = min(c0, C - 8);                    // threads beyond C exit below; keep their loads in bounds
    uint32_t wlo[9][4], whi[9][4];
    float bs[8];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        float wv[8];
        load8(w + t * C + cw, wv);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bf16 a = (bf16)wv[2 * j], b = (bf16)wv[2 * j + 1];
            wlo[t][j] = (uint32_t)__builtin_bit_cast(uint16_t, a);
            whi[t][j] = (uint32_t)__builtin_bit_cast(uint16_t, b) << 16;
        }
    }
    if (bias) load8(bias + cw, bs);
    else {
#pragma unroll
        for (int e = 0; e < 8; ++e) bs[e] = 0.f;
    }
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
        const int idx = tid + k * 256;
        const u32x4 z4 = {0, 0, 0, 0};
        if (idx < HY * HX * 8) *(u32x4 *)(tile + (idx >> 3) * DW_CB + (idx & 7) * 8) = ((okm >> k) & 1) ? vals[k] : z4;
    }
    __syncthreads();
    const int x = x0 + px;
    if (x >= W || !cok) return;
    const int r0 = rhalf * (DWB_T / 2);
    uint32_t win[3][3][4];                            // [row slot][dx][channel pair]
    const bf16 *col = tile + px * DW_CB + cv * 8;
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            const u32x4 t4 = *(const u32x4 *)(col + ((r0 + s) * HX + dx) * DW_CB);
            win[s][dx][0] = t4.x; win[s][dx][1] = t4.y; win[s][dx][2] = t4.z; win[s][dx][3] = t4.w;
        }
#pragma unroll
    for (int rr = 0; rr < DWB_T / 2; ++rr) {
        const int r = r0 + rr, y = y0 + r;
        if (y >= H) break;
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            const u32x4 t4 = *(const u32x4 *)(col + ((r + 2) * HX + dx) * DW_CB);
            uint32_t *wr = win[(rr + 2) % 3][dx];
            wr[0] = t4.x; wr[1] = t4.y; wr[2] = t4.z; wr[3] = t4.w;
        }
        float acc[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = bs[e];
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const bf16x2 xv = __builtin_bit_cast(bf16x2, win[(rr + dy) % 3][dx][j]);
                    acc[2 * j] = fd_dot2(xv, __builtin_bit_cast(bf16x2, wlo[dy * 3 + dx][j]), acc[2 * j]);
                    acc[2 * j + 1] = fd_dot2(xv, __builtin_bit_cast(bf16x2, whi[dy * 3 + dx][j]), acc[2 * j + 1]);
                }
        if (silu) fd_silu8(acc);
        store8(out_img + (__umul24(__umul24(y, W) + x, ld_out) + c0), acc);
    }
}

template <typename T>
__global__ void avgpool_kernel(const T *__restrict__ in, T *__restrict__ out, int H, int W, int C, int k,
                               int64_t total) {
    const int OH = H / k, OW = W / k, vpp = C / 8;
    const float inv = 1.f / (k * k);
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int cv = (int)(i % vpp);
        const int64_t pix = i / vpp;
        const int ox = (int)(pix % OW), oy = (int)((pix / OW) % OH);
        const int64_t img = pix / ((int64_t)OW * OH);
        float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int dy = 0; dy < k; ++dy)
            for (int dx = 0; dx < k; ++dx) {
                float v[8];
                load8(in + ((img * H + oy * k + dy) * W + ox * k + dx) * C + cv * 8, v);
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] += v[e];
            }
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] *= inv;
        store8(out + pix * C + cv * 8, acc);
    }
}

template <typename T>
__global__ void pack_planes_kernel(const float *__restrict__ p0, const float *__restrict__ p1,
                                   const float *__restrict__ p2, T *__restrict__ out, int64_t npix, int cpad) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < npix;
         i += (int64_t)gridDim.x * blockDim.x) {
        float v[8] = {p0[i], p1 ? p1[i] : 0.f, p2 ? p2[i] : 0.f, 0, 0, 0, 0, 0};
        store8(out + i * cpad, v);
        for (int c = 8; c < cpad; c += 8) {
            float zz[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            store8(out + i * cpad + c, zz);
        }
    }
}

// out[p] = b + sum_c x[p,c] w[c]; C/8 lanes cooperate on one pixel.
template <typename T>
__global__ __launch_bounds__(256) void final_conv1_kernel(const T *__restrict__ x, const float *__restrict__ w,
                                                         const float *__restrict__ bias, float *__restrict__ out,
                                                         int64_t npix, int C, int lpp) {
    const int ppb = 256 / lpp;
    const int sub = threadIdx.x % lpp;
    const int64_t pix = (int64_t)blockIdx.x * ppb + threadIdx.x / lpp;
    const bool active = pix < npix;
    float s = 0.f;
    if (active)
        for (int c0 = sub * 8; c0 < C; c0 += lpp * 8) {
            float v[8], ww[8];
            load8(x + pix * C + c0, v);
            load8(w + c0, ww);
#pragma unroll
            for (int e = 0; e < 8; ++e) s += v[e] * ww[e];
        }
    for (int o = 1; o < lpp; o <<= 1) s += __shfl_xor(s, o, 64);
    if (active && sub == 0) out[pix] = s + bias[0];
}

// tokens[b,0,:] = mean over hw; tokens[b,1+p,:] = x[b,p,:]
template <typename T>
__global__ void attnpool_tokens_kernel(const T *__restrict__ x, T *__restrict__ tok, int64_t hw, int C) {
    const int b = blockIdx.y;
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float s = 0.f;
    for (int64_t p = 0; p < hw; ++p) {
        float v = ld1(x + ((int64_t)b * hw + p) * C + c);
        s += v;
        st1(tok + ((int64_t)b * (hw + 1) + 1 + p) * C + c, v);
    }
    st1(tok + (int64_t)b * (hw + 1) * C + c, s / (float)hw);
}

unsigned grid1d(int64_t n, int block = 256, int cap = 4096) {
    int64_t g = (n + block - 1) / block;
    return (unsigned)(g > cap ? cap : (g < 1 ? 1 : g));
}

}  // namespace

extern "C" int fd_ln_modulate(int dtype, const void *x, const float *gamma, const float *beta, float eps,
                              const float *shift, const float *scale, int mod_ld, void *out, int B,
                              int64_t hw, int C, void *stream) {
    FD_REQUIRE(x && out && shift && scale, "fd_ln_modulate: null pointer");
    FD_REQUIRE((gamma == nullptr) == (beta == nullptr), "fd_ln_modulate: gamma/beta must both be set or NULL");
    int rc = dtype == FD_BF16
                 ? launch_ln<bf16, false>((const bf16 *)x, gamma, beta, eps, shift, scale, mod_ld, nullptr, 0, 0,
                                          (bf16 *)out, B, hw, C, (hipStream_t)stream)
                 : launch_ln<float, false>((const float *)x, gamma, beta, eps, shift, scale, mod_ld, nullptr, 0, 0,
                                           (float *)out, B, hw, C, (hipStream_t)stream);
    if (rc) return rc;
    FD_LAUNCH_OK("fd_ln_modulate");
    return FD_OK;
}

extern "C" int fd_ln_gate(int dtype, const void *y, const float *gamma, const float *beta, float eps,
                          const void *z, int ldz, int offz, const float *local, int local_ld, void *out,
                          int B, int64_t hw, int C, void *stream) {
    FD_REQUIRE(y && z && local && out && gamma && beta, "fd_ln_gate: null pointer");
    FD_REQUIRE(ldz % 8 == 0 && offz % 8 == 0, "fd_ln_gate: z stride/offset must be multiples of 8");
    int rc = dtype == FD_BF16
                 ? launch_ln<bf16, true>((const bf16 *)y, gamma, beta, eps, local, nullptr, local_ld, (const bf16 *)z,
                                         ldz, offz, (bf16 *)out, B, hw, C, (hipStream_t)stream)
                 : launch_ln<float, true>((const float *)y, gamma, beta, eps, local, nullptr, local_ld,
                                          (const float *)z, ldz, offz, (float *)out, B, hw, C, (hipStream_t)stream);
    if (rc) return rc;
    FD_LAUNCH_OK("fd_ln_gate");
    return FD_OK;
}

extern "C" int fd_dwconv3x3(int dtype, const void *in, int ld_in, int off_in, const float *weight,
                            const float *bias, int silu, void *out, int ld_out, int off_out, int B, int H,
                            int W, int C, void *stream) {
    FD_REQUIRE(in && out && weight, "fd_dwconv3x3: null pointer");
    FD_REQUIRE(C % 8 == 0 && ld_in % 8 == 0 && off_in % 8 == 0 && ld_out % 8 == 0 && off_out % 8 == 0,
               "fd_dwconv3x3: channels/strides/offsets must be multiples of 8");
    const int cblocks = (C + DW_CB - 1) / DW_CB;
    dim3 grid(((W + DW_TX - 1) / DW_TX) * cblocks, (H + DW_TY - 1) / DW_TY, (unsigned)B), block(256);
    const size_t lds = (size_t)(DW_TY + 2) * (DW_TX + 2) * DW_CB * (dtype == FD_BF16 ? 2 : 4);
    if (dtype == FD_BF16) {
        FD_REQUIRE((int64_t)H * W < (1 << 24) && (int64_t)H * W * (ld_in > ld_out ? ld_in : ld_out) < (1ll << 31),
                   "fd_dwconv3x3: image too large for the 24-bit pixel / 32-bit element indices of the bf16 kernel");
        dim3 gridb(((W + DWB_T - 1) / DWB_T) * cblocks, (H + DWB_T - 1) / DWB_T, (unsigned)B);
        const size_t ldsb = (size_t)(DWB_T + 2) * (DWB_T + 2) * DW_CB * 2;
        hipLaunchKernelGGL(dwconv3x3_bf16_kernel, gridb, block, ldsb, (hipStream_t)stream, (const bf16 *)in, ld_in,
                           off_in, weight, bias, silu, (bf16 *)out, ld_out, off_out, H, W, C, cblocks);
    } else {
        static bool attr_set = false;
        if (!attr_set) {
            (void)hipFuncSetAttribute((const void *)dwconv3x3_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            attr_set = true;
        }
        hipLaunchKernelGGL(dwconv3x3_kernel<float>, grid, block, lds, (hipStream_t)stream, (const float *)in, ld_in,
                           off_in, weight, bias, silu, (float *)out, ld_out, off_out, H, W, C, cblocks);
    }
    FD_LAUNCH_OK("fd_dwconv3x3");
    return FD_OK;
}

extern "C" int fd_avgpool(int dtype, const void *in, void *out, int B, int H, int W, int C, int k, void *stream) {
    // floor semantics of nn.AvgPool2d (src/DACLIP.py:169,187,337): a trailing partial window is dropped
    FD_REQUIRE(C % 8 == 0 && k > 0 && H >= k && W >= k, "fd_avgpool: C%%8 must be 0 and H, W >= k");
    int64_t total = (int64_t)B * (H / k) * (W / k) * (C / 8);
    dim3 grid(grid1d(total)), block(256);
    if (dtype == FD_BF16)
        hipLaunchKernelGGL(avgpool_kernel<bf16>, grid, block, 0, (hipStream_t)stream, (const bf16 *)in, (bf16 *)out, H, W,
                           C, k, total);
    else
        hipLaunchKernelGGL(avgpool_kernel<float>, grid, block, 0, (hipStream_t)stream, (const float *)in, (float *)out, H,
                           W, C, k, total);
    FD_LAUNCH_OK("fd_avgpool");
    return FD_OK;
}

// activation tensor in the other storage type (hybrid-precision forward: fp32 <-> bf16 at a level boundary)
template <typename S, typename D>
__global__ void cast_kernel(const S *__restrict__ in, D *__restrict__ out, int64_t n8) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
        float v[8];
        load8(in + 8 * i, v);
        store8(out + 8 * i, v);
    }
}

extern "C" int fd_cast(int src_dtype, const void *in, int dst_dtype, void *out, int64_t n, void *stream) {
    FD_REQUIRE(in && out && n % 8 == 0 && (((uintptr_t)in | (uintptr_t)out) & 15) == 0, "fd_cast: n %% 8 and 16-byte alignment");
    FD_REQUIRE(src_dtype != dst_dtype, "fd_cast: same dtype");
    dim3 grid(grid1d(n / 8)), block(256);
    if (src_dtype == FD_F32)
        hipLaunchKernelGGL((cast_kernel<float, bf16>), grid, block, 0, (hipStream_t)stream, (const float *)in, (bf16 *)out, n / 8);
    else
        hipLaunchKernelGGL((cast_kernel<bf16, float>), grid, block, 0, (hipStream_t)stream, (const bf16 *)in, (float *)out, n / 8);
    FD_LAUNCH_OK("fd_cast");
    return FD_OK;
}

extern "C" int fd_pack_planes3(int dtype, const float *p0, const float *p1, const float *p2, void *out, int B,
                               int64_t hw, int cpad, void *stream) {
    FD_REQUIRE(p0 && out && cpad >= 8 && cpad % 8 == 0, "fd_pack_planes: bad args");
    int64_t npix = (int64_t)B * hw;
    dim3 grid(grid1d(npix)), block(256);
    if (dtype == FD_BF16)
        hipLaunchKernelGGL(pack_planes_kernel<bf16>, grid, block, 0, (hipStream_t)stream, p0, p1, p2, (bf16 *)out, npix, cpad);
    else
        hipLaunchKernelGGL(pack_planes_kernel<float>, grid, block, 0, (hipStream_t)stream, p0, p1, p2, (float *)out, npix, cpad);
    FD_LAUNCH_OK("fd_pack_planes");
    return FD_OK;
}
extern "C" int fd_pack_planes(int dtype, const float *p0, const float *p1, void *out, int B, int64_t hw, int cpad,
                              void *stream) {
    return fd_pack_planes3(dtype, p0, p1, nullptr, out, B, hw, cpad, stream);
}

extern "C" int fd_final_conv1(int dtype, const void *x, const float *w, const float *b, float *out, int64_t npix,
                              int C, void *stream) {
    FD_REQUIRE(x && w && b && out && C % 8 == 0, "fd_final_conv1: bad args");
    int lpp = 1;
    while (lpp * 2 * 8 <= C && lpp < 64) lpp *= 2;
    int ppb = 256 / lpp;
    dim3 grid((unsigned)((npix + ppb - 1) / ppb)), block(256);
    if (dtype == FD_BF16)
        hipLaunchKernelGGL(final_conv1_kernel<bf16>, grid, block, 0, (hipStream_t)stream, (const bf16 *)x, w, b, out, npix, C, lpp);
    else
        hipLaunchKernelGGL(final_conv1_kernel<float>, grid, block, 0, (hipStream_t)stream, (const float *)x, w, b, out, npix, C, lpp);
    FD_LAUNCH_OK("fd_final_conv1");
    return FD_OK;
}

extern "C" int fd_attnpool_tokens(int dtype, const void *x, void *tok, int B, int64_t hw, int C, void *stream) {
    dim3 grid((C + 255) / 256, B), block(256);
    if (dtype == FD_BF16)
        hipLaunchKernelGGL(attnpool_tokens_kernel<bf16>, grid, block, 0, (hipStream_t)stream, (const bf16 *)x, (bf16 *)tok, hw, C);
    else
        hipLaunchKernelGGL(attnpool_tokens_kernel<float>, grid, block, 0, (hipStream_t)stream, (const float *)x, (float *)tok, hw, C);
    FD_LAUNCH_OK("fd_attnpool_tokens");
    return FD_OK;
}
