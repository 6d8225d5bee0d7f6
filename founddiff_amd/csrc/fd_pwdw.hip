// fd_pwdw.hip -- fused [LayerNorm + adaLN modulate] -> 1x1 conv -> depthwise 3x3 (+bias, SiLU) for the
// high-resolution Mamba blocks (bf16, Cin = 64):
//     SS2D:               in_proj (C -> 2D)  then conv2d(3x3, groups = D) + SiLU on the x half,
//                         SiLU on the z half                                  src/emamba2.py:716-722
//     TransposedAttention qkv (C -> 3C)      then qkv_dwconv(3x3, groups = 3C)      src/DADiff.py:266,275
// Unfused, the 1x1 output (128 / 192 channels per pixel) is written to HBM by the row-GEMM and read
// back by the depthwise kernel: 2 x 0.5..0.8 GB per launch pair at 512x512, batch 8 -- a quarter of a
// Mamba block's traffic.  Here a workgroup owns an 8 x 16 pixel tile:
//   phase 0  load the (8+2) x (16+2) halo of the 64-channel input once (all loads in flight before the
//            first use), LayerNorm + modulate it in registers (a pixel's 64 channels sit in 8
//            neighbouring lanes: three xor-shuffles), park it in LDS as bf16, swizzled for MFMA reads;
//   phase Z  (in_proj) z = SiLU(W_z . xn) for the 128 interior pixels straight from the MFMA
//            accumulators to HBM -- the row-GEMM's transposed issue, 8 consecutive channels per lane;
//   per 64-channel chunk of the depthwise part:
//   phase 1  t = W_chunk . xn over ALL 180 halo pixels (the 1.41x recompute is MFMA time nobody
//            misses), rounded to bf16 into a second LDS tile -- the same rounding point as the unfused
//            path's HBM round trip -- with out-of-image pixels forced to the conv's zero padding;
//   phase 2  the depthwise 3x3 of dwconv3x3_bf16_kernel on that tile (register row window,
//            v_dot2c_f32_bf16 against pre-masked bf16 tap weights), 16-byte stores.
#include "fd_common.h"

namespace {

constexpr int PT_H = 8, PT_W = 16, PH_Y = PT_H + 2, PH_X = PT_W + 2, PHP = PH_Y * PH_X;   // 180 halo pixels
constexpr int PMT = (PHP + 15) / 16;                                                      // 12 m-tiles
constexpr int XS_B = PMT * 16 * 128, TS_B = PHP * 128;
typedef __attribute__((ext_vector_type(2))) __bf16 pd_bf16x2;

// row & 7 (not the generic kernel's (row >> 1) & 7): conflict-free for 16 consecutive rows starting at ANY row -- the
// z phase reads its fragments from tile rows that start at odd halo pixels (fd_conv3x3.hip, swz)
__device__ __forceinline__ int xs_off(int row, int chunk) { return row * 128 + ((chunk ^ (row & 7)) << 4); }
__device__ __forceinline__ int ts_off(int row, int chunk) { return row * 128 + ((chunk ^ (row & 7)) << 4); }

// hp / 18 for hp < 512 with one full-rate 24-bit multiply (the compiler's exact division is a quarter-rate mul_hi)
__device__ __forceinline__ int div18(int hp) { return (int)(__umul24((unsigned)hp, 57u) >> 10); }

// Workgroup barrier that waits for this wave's LDS traffic only.  __syncthreads() is a fence + barrier: hipcc emits
// s_waitcnt vmcnt(0) in front of it, i.e. every wave stops until all its outstanding global STORES are acknowledged
// -- here that is up to 32 KiB of z / depthwise output per workgroup per phase, and it is what made the phases of this
// kernel add up instead of overlapping (z stores 109 us + depthwise 107 us + ... = the whole 472 us).  The tiles only
// communicate through LDS, so the stores may stay in flight across the barrier.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

struct PwDwParams {
    const bf16 *x; int ld_x, off_x;
    const float *ln_gamma, *ln_beta, *ln_shift, *ln_scale; int ln_ld; float ln_eps;
    const bf16 *w_pw;                 // [Cdw + Cz][64]
    int Cdw; const uint32_t *w_dw; const float *b_dw; int dw_silu;
    bf16 *out_dw; int ld_dw, off_dw;
    int Cz; bf16 *out_z; int ld_z, off_z;
    int H, W;
};

__global__ __launch_bounds__(256, 3) void pwdw_kernel(const PwDwParams p) {
    __shared__ __attribute__((aligned(16))) unsigned char xs[XS_B];     // LN'd input halo, [192][64] bf16
    __shared__ __attribute__((aligned(16))) unsigned char ts[TS_B];     // 1x1 output of one chunk, [180][64] bf16
    __shared__ float sV[2][64];
    // tap-pair words [5][Cdw] + bias [Cdw] of the depthwise conv, staged once: a global load issued after a phase's
    // stores cannot be waited for without waiting for those stores too (vmcnt retires in order)
    __shared__ __attribute__((aligned(16))) uint32_t sW[6 * 192];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fg = lane >> 4;
    const int tiles_x = p.W / PT_W;
    const int ty0 = (blockIdx.x / tiles_x) * PT_H, tx0 = (blockIdx.x % tiles_x) * PT_W;
    const int64_t img = blockIdx.y;
    const bf16 *xin = p.x + img * p.H * p.W * p.ld_x + p.off_x;                 // wave-uniform image bases
    bf16 *zout = p.Cz > 0 ? p.out_z + img * p.H * p.W * p.ld_z + p.off_z : nullptr;
    bf16 *dwout = p.out_dw + img * p.H * p.W * p.ld_dw + p.off_dw;

    // ---- phase 0: halo load -> LayerNorm + modulate -> xs
    if (tid < 64) {
        const float sc = 1.f + p.ln_scale[img * p.ln_ld + tid], sh = p.ln_shift[img * p.ln_ld + tid];
        const float g = p.ln_gamma ? p.ln_gamma[tid] : 1.f, be = p.ln_beta ? p.ln_beta[tid] : 0.f;
        sV[0][tid] = g * sc;
        sV[1][tid] = be * sc + sh;
    }
    for (int i = tid; i < 5 * p.Cdw; i += 256) sW[i] = p.w_dw[i];
    for (int i = tid; i < p.Cdw; i += 256) sW[5 * p.Cdw + i] = p.b_dw ? __builtin_bit_cast(uint32_t, p.b_dw[i]) : 0u;
    constexpr int NLD = (PMT * 16 * 8) / 256;         // 6 chunks per thread cover all 192 rows
    const int v = tid & 7;
    u32x4 raw[NLD];
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
        const int hp = (tid + k * 256) >> 3;
        const int hy = div18(hp), hx = hp - hy * PH_X;
        const int yc = min(max(ty0 + hy - 1, 0), p.H - 1), xc = min(max(tx0 + hx - 1, 0), p.W - 1);
        // 32-bit element offsets from a per-image scalar base, 24-bit multiplies: the 64-bit form of this address
        // cost 3 v_mad_u64_u32 + 4 v_mul_lo_u32 (quarter rate) per load
        raw[k] = *(const u32x4 *)(xin + (__umul24(__umul24(yc, p.W) + xc, p.ld_x) + v * 8));
    }
    lds_barrier();                                   // sV
    float g8[8], b8[8];
    load8(&sV[0][v * 8], g8);
    load8(&sV[1][v * 8], b8);
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
        const int hp = (tid + k * 256) >> 3;
        float f[8];
        const bf16x8 xv = __builtin_bit_cast(bf16x8, raw[k]);
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] = (float)xv[e];
        float s = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) s += f[e];
        s += __shfl_xor(s, 1, 64);
        s += __shfl_xor(s, 2, 64);
        s += __shfl_xor(s, 4, 64);
        const float mean = s * (1.f / 64);
        float q = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float d = f[e] - mean; q += d * d; }
        q += __shfl_xor(q, 1, 64);
        q += __shfl_xor(q, 2, 64);
        q += __shfl_xor(q, 4, 64);
        const float rstd = rsqrtf(q * (1.f / 64) + p.ln_eps);
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (bf16)((f[e] - mean) * rstd * g8[e] + b8[e]);
        *(bf16x8 *)(xs + xs_off(hp, v)) = o;
    }
    // which of this lane's halo pixels (m-tile mt, row fr) lie inside the image: the depthwise conv
    // zero-pads ITS input, i.e. the 1x1 output -- not LN(0)
    uint32_t inside = 0;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int hp = (3 * wave + i) * 16 + fr;
        const int hy = div18(hp), hx = hp - hy * PH_X;
        const int yy = ty0 + hy - 1, xx = tx0 + hx - 1;
        if (hp < PHP && yy >= 0 && yy < p.H && xx >= 0 && xx < p.W) inside |= 1u << i;
    }
    lds_barrier();

    // weight rows of a 32-channel group, permuted so that a lane ends up with 8 consecutive channels
    const int rperm = 8 * (fr >> 2) + (fr & 3);
    // The 1x1 weights are consumed in 32-row groups: first the Cz/32 pass-through groups (rows Cdw + 32g),
    // then two groups per depthwise chunk (rows 32g).  Group g+1 is loaded (L2) while group g is in the
    // MFMAs -- loaded right before use, each group exposed a full L2 round trip (~15k cycles per workgroup).
    const int nz = p.Cz / 32, ngroups = nz + p.Cdw / 32;
    bf16x8 wnext[4];
    auto wload = [&](int g) {
        const int rb = g < nz ? p.Cdw + 32 * g : 32 * (g - nz);
        const bf16 *wr = p.w_pw + (int64_t)(rb + rperm) * 64 + fg * 8;
        wnext[0] = *(const bf16x8 *)(wr);
        wnext[1] = *(const bf16x8 *)(wr + 32);
        wnext[2] = *(const bf16x8 *)(wr + 4 * 64);
        wnext[3] = *(const bf16x8 *)(wr + 4 * 64 + 32);
    };
    int gi = 0;
    wload(0);

    // ---- phase Z: pass-through channels (z), interior pixels only, accumulators -> HBM
    if (p.Cz > 0) {
        bf16x8 xb[2][2];                               // [tile row][ks]
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = (2 * wave + i + 1) * PH_X + 1 + fr;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) xb[i][ks] = *(const bf16x8 *)(xs + xs_off(row, ks * 4 + fg));
        }
        for (int ng = 0; ng < nz; ++ng) {
            const bf16x8 wa[2] = {wnext[0], wnext[1]}, wb[2] = {wnext[2], wnext[3]};
            if (++gi < ngroups) wload(gi);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    a0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[ks], xb[i][ks], a0, 0, 0, 0);
                    a1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[ks], xb[i][ks], a1, 0, 0, 0);
                }
                float val[8];
#pragma unroll
                for (int e = 0; e < 4; ++e) { val[e] = fd_silu(a0[e]); val[4 + e] = fd_silu(a1[e]); }
                const int y = ty0 + 2 * wave + i, x = tx0 + fr;
                store8(zout + (__umul24(__umul24(y, p.W) + x, p.ld_z) + 32 * ng + 8 * fg), val);
            }
        }
    }

    // ---- depthwise part, 64 channels at a time
    // phase-2 roles: 8-channel vector cv of tile column px, rows 4*rh .. 4*rh+3
    const int cv = tid & 7, px = (tid >> 3) & 15, rh = tid >> 7;
    int toff[6][3];                                    // swizzled LDS offsets of the 6 x 3 window positions
#pragma unroll
    for (int r = 0; r < 6; ++r)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            const int hp = (4 * rh + r) * PH_X + px + dx;
            toff[r][dx] = ts_off(hp, cv);
        }
    for (int ch = 0; ch < p.Cdw / 64; ++ch) {
        // tap weights of this chunk ([5][Cdw] words, see fd_pw_dw3x3): issued
        // before the MFMAs so they arrive during phase 1
        const int c0 = 64 * ch + cv * 8;
        uint32_t wt[5][8];
        float bs[8];
#pragma unroll
        for (int t = 0; t < 5; ++t) {
            const u32x4 w0 = *(const u32x4 *)(sW + t * p.Cdw + c0), w1 = *(const u32x4 *)(sW + t * p.Cdw + c0 + 4);
            wt[t][0] = w0.x; wt[t][1] = w0.y; wt[t][2] = w0.z; wt[t][3] = w0.w;
            wt[t][4] = w1.x; wt[t][5] = w1.y; wt[t][6] = w1.z; wt[t][7] = w1.w;
        }
        load8((const float *)(sW + 5 * p.Cdw + c0), bs);
        // phase 1: t[hp][64 ch] = W_chunk . xn, 3 m-tiles per wave (fragments re-read per chunk: registers
        // are the scarce resource of this kernel, LDS bandwidth is not)
        bf16x8 xh[3][2];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
                xh[i][ks] = *(const bf16x8 *)(xs + xs_off((3 * wave + i) * 16 + fr, ks * 4 + fg));
#pragma unroll
        for (int ng = 0; ng < 2; ++ng) {
            const bf16x8 wa[2] = {wnext[0], wnext[1]}, wb[2] = {wnext[2], wnext[3]};
            if (++gi < ngroups) wload(gi);
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    a0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[ks], xh[i][ks], a0, 0, 0, 0);
                    a1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[ks], xh[i][ks], a1, 0, 0, 0);
                }
                const int hp = (3 * wave + i) * 16 + fr;
                const bool in = (inside >> i) & 1;
                bf16x8 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    o[e] = (bf16)(in ? a0[e] : 0.f);
                    o[4 + e] = (bf16)(in ? a1[e] : 0.f);
                }
                if (hp < PHP) *(bf16x8 *)(ts + ts_off(hp, 4 * ng + fg)) = o;
            }
        }
        lds_barrier();
        // phase 2: depthwise 3x3 on ts
        uint32_t win[3][3][4];
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const u32x4 t4 = *(const u32x4 *)(ts + toff[s][dx]);
                win[s][dx][0] = t4.x; win[s][dx][1] = t4.y; win[s][dx][2] = t4.z; win[s][dx][3] = t4.w;
            }
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const u32x4 t4 = *(const u32x4 *)(ts + toff[rr + 2][dx]);
                uint32_t *wr_ = win[(rr + 2) % 3][dx];
                wr_[0] = t4.x; wr_[1] = t4.y; wr_[2] = t4.z; wr_[3] = t4.w;
            }
            float acc[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] = bs[e];
            // tap pairs: the dot2 contracts TWO taps of one channel -- v_perm gathers the channel's
            // two tap inputs into one word; the weight word holds the two tap weights (40 weight
            // registers per 8 channels instead of 72 pre-masked ones, same instruction count)
#pragma unroll
            for (int pr = 0; pr < 4; ++pr)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const uint32_t xa = pr < 3 ? win[rr % 3][pr][j] : win[(rr + 2) % 3][0][j];
                    const uint32_t xb = pr < 3 ? win[(rr + 1) % 3][pr][j] : win[(rr + 2) % 3][1][j];
                    const uint32_t lo = __builtin_amdgcn_perm(xb, xa, 0x05040100);    // (xa.lo, xb.lo): channel 2j
                    const uint32_t hi = __builtin_amdgcn_perm(xb, xa, 0x07060302);    // (xa.hi, xb.hi): channel 2j+1
                    acc[2 * j] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(pd_bf16x2, lo),
                        __builtin_bit_cast(pd_bf16x2, wt[pr][2 * j]), acc[2 * j], false);
                    acc[2 * j + 1] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(pd_bf16x2, hi),
                        __builtin_bit_cast(pd_bf16x2, wt[pr][2 * j + 1]), acc[2 * j + 1], false);
                }
#pragma unroll
            for (int j = 0; j < 4; ++j) {              // the ninth tap alone: weight in the channel's half, 0 in the other
                const pd_bf16x2 xv = __builtin_bit_cast(pd_bf16x2, win[(rr + 2) % 3][2][j]);
                acc[2 * j] = __builtin_amdgcn_fdot2_f32_bf16(xv, __builtin_bit_cast(pd_bf16x2, wt[4][2 * j]), acc[2 * j], false);
                acc[2 * j + 1] = __builtin_amdgcn_fdot2_f32_bf16(xv, __builtin_bit_cast(pd_bf16x2, wt[4][2 * j + 1]), acc[2 * j + 1], false);
            }
            if (p.dw_silu) {
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] = fd_silu(acc[e]);
            }
            const int y = ty0 + 4 * rh + rr, x = tx0 + px;
            store8(dwout + (__umul24(__umul24(y, p.W) + x, p.ld_dw) + c0), acc);
        }
        lds_barrier();                               // ts is rewritten by the next chunk
    }
}

}  // namespace

extern "C" int fd_pw_dw3x3_ok(int dtype, int Cin, int Cdw, int Cz, int H, int W) {
    return dtype == FD_BF16 && Cin == 64 && Cdw > 0 && Cdw <= 192 && Cdw % 64 == 0 && Cz >= 0 && Cz % 32 == 0 && H % PT_H == 0 &&
           W % PT_W == 0 && (int64_t)H * W >= 32768 && (int64_t)H * W * 256 < (1ll << 31);   // 32-bit element offsets
}

extern "C" int fd_pw_dw3x3(int dtype, const void *x, int ld_x, int off_x, int Cin, const float *ln_gamma,
                           const float *ln_beta, float ln_eps, const float *ln_shift, const float *ln_scale,
                           int ln_ld, const void *w_pw, int Cdw, const uint32_t *w_dw, const float *b_dw,
                           int dw_silu, void *out_dw, int ld_dw, int off_dw, int Cz, void *out_z, int ld_z,
                           int off_z, int B, int H, int W, void *stream) {
    FD_REQUIRE(fd_pw_dw3x3_ok(dtype, Cin, Cdw, Cz, H, W),
               "fd_pw_dw3x3: unsupported shape (bf16, Cin=64, Cdw%%64, Cz%%32, H%%8, W%%16, >= 32768 px): "
               "Cin=%d Cdw=%d Cz=%d H=%d W=%d", Cin, Cdw, Cz, H, W);
    FD_REQUIRE(x && ln_shift && ln_scale && w_pw && w_dw && out_dw && (Cz == 0 || out_z), "fd_pw_dw3x3: null pointer");
    FD_REQUIRE(ld_x % 8 == 0 && off_x % 8 == 0 && ld_dw % 8 == 0 && off_dw % 8 == 0 && ld_z % 8 == 0 && off_z % 8 == 0,
               "fd_pw_dw3x3: strides / offsets must be multiples of 8 channels");
    FD_REQUIRE(((uintptr_t)w_dw & 15) == 0 && (!b_dw || ((uintptr_t)b_dw & 15) == 0), "fd_pw_dw3x3: weights must be 16-byte aligned");
    PwDwParams p;
    p.x = (const bf16 *)x; p.ld_x = ld_x; p.off_x = off_x;
    p.ln_gamma = ln_gamma; p.ln_beta = ln_beta; p.ln_shift = ln_shift; p.ln_scale = ln_scale; p.ln_ld = ln_ld; p.ln_eps = ln_eps;
    p.w_pw = (const bf16 *)w_pw;
    p.Cdw = Cdw; p.w_dw = w_dw; p.b_dw = b_dw; p.dw_silu = dw_silu;
    p.out_dw = (bf16 *)out_dw; p.ld_dw = ld_dw; p.off_dw = off_dw;
    p.Cz = Cz; p.out_z = (bf16 *)out_z; p.ld_z = ld_z; p.off_z = off_z;
    p.H = H; p.W = W;
    dim3 grid((H / PT_H) * (W / PT_W), B), block(256);
    hipLaunchKernelGGL(pwdw_kernel, grid, block, 0, (hipStream_t)stream, p);
    FD_LAUNCH_OK("fd_pw_dw3x3");
    return FD_OK;
}
