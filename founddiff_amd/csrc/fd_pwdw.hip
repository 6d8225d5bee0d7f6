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
#include <stdlib.h>

namespace {

constexpr int PT_H = 8, PT_W = 16, PH_Y = PT_H + 2, PH_X = PT_W + 2, PHP = PH_Y * PH_X;   // 180 halo pixels
constexpr int PMT = (PHP + 15) / 16;                                                      // 12 m-tiles
constexpr int XS_B = PMT * 16 * 128, TS_B = PHP * 128;
typedef __attribute__((ext_vector_type(2))) bf16 pd_bf16x2;
typedef _Float16 pd_h2 __attribute__((ext_vector_type(2)));
typedef _Float16 pd_h8 __attribute__((ext_vector_type(8)));

// ---- the depthwise 3x3 on PACKED fp16 (round 3).  The 1x1 output t lives in LDS as fp16 channel pairs (11 mantissa
// bits instead of bf16's 8: the tile never leaves the CU, so its format is free), the tap weights are fp16 channel
// pairs, and one v_pk_fma_f16 does a tap for TWO channels: 36 instructions per row of 8 channels instead of 32 v_perm +
// 40 v_dot2 (the bf16 form had to gather tap pairs per channel because the dot product contracts WITHIN a register).
// The 9-term sum is accumulated in fp16 (~7e-4 rms relative, below the bf16 rounding of the output it feeds; the
// input side gains 3 bits).  v_cvt_pkrtz_f16_f32 saturates instead of producing infinities (|t| > 65504).
// (binary16 build: v_cvt_pk_f16_f32, round to nearest even -- there the tile's rounding is no longer below the output's, and a
//  round-toward-zero conversion is a bias of half an ulp in every stored xc / q / k / v; out-of-range values become infinities like
//  everywhere else in that build, and sample() reports them)
__device__ __forceinline__ uint32_t pk_h2(float a, float b) {
#ifdef FD_HALF_F16
    return __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2){a, b}, pd_h2));
#else
    return __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pkrtz(a, b));
#endif
}
// a dword of two stored 16-bit elements as an fp16 pair (exact from bfloat16 inside fp16's range; the identity in the FD_HALF_F16 build)
__device__ __forceinline__ uint32_t h16_to_h2(uint32_t w) {
#ifdef FD_HALF_F16
    return w;
#else
    return pk_h2(fd_h_lo(w), fd_h_hi(w));
#endif
}

__device__ __forceinline__ void dw_row_f16(const uint32_t (&win)[3][3][4], int rr, const uint32_t (&wt)[9][4],
                                           const uint32_t (&b2)[4], uint32_t (&out)[4]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        pd_h2 acc = __builtin_bit_cast(pd_h2, b2[j]);
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx)
                acc = __builtin_elementwise_fma(__builtin_bit_cast(pd_h2, win[(rr + dy) % 3][dx][j]),
                                                __builtin_bit_cast(pd_h2, wt[dy * 3 + dx][j]), acc);
        out[j] = __builtin_bit_cast(uint32_t, acc);
    }
}

__device__ __forceinline__ void unpack_h8(const uint32_t (&a)[4], float (&f)[8]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const pd_h2 v = __builtin_bit_cast(pd_h2, a[j]);
        f[2 * j] = (float)v.x;
        f[2 * j + 1] = (float)v.y;
    }
}

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
    int tpw, ntiles;                  // consecutive tiles per workgroup (round 3: the tap weights / modulation vectors are
                                      // staged once per workgroup, the 1x1 weight prefetch runs across tiles)
    // PROJ form (fd_pw_dw3x3_proj): a second 1x1 (64 -> 64, per-image weights) on the depthwise output + gated residual
    const bf16 *w2; int64_t w2_bstride;
    const float *gate; int gate_ld;
    bf16 *out2; int ld_o2, off_o2;
};

// CIN = 64 (round 1: levels 0-1, three workgroups per CU) or 128 (round 3: the C = 128 blocks at 256x256; the halo tile
// is 48 KB, rows of 256 bytes swizzled over their 16 chunks, K = 128 in four K32 steps; two workgroups per CU).
// PROJ (round 4, CIN = 64, Cdw = 64, no pass-through): the v branch of TransposedAttention up to the block output,
//   x2 = x + gate . (Weff[b] . dwconv3x3(W_v . LNmod(x)))                       src/DADiff.py:266-285, 483-488
// with Weff[b] = project_out . blockdiag(softmax(q k^T)) from fd_chan_attn_weff.  The Gram kernel then handles q and k only
// and v never reaches HBM: the depthwise output goes as bf16 (the rounding point of the stored v) over the dead
// LayerNorm'd halo tile, Weff[b] (8 KB) over that tile's unused rows 128..191, and the second 1x1 runs on it like the
// row-GEMM (transposed issue, 8 consecutive channels per lane) with the residual re-read from L2.
template <int CIN, bool PROJ = false>
__global__ __launch_bounds__(256, CIN == 64 ? 3 : 2) void pwdw_kernel(const PwDwParams p) {
    constexpr int CH = CIN / 8, KS = CIN / 32, LCH = CIN == 64 ? 3 : 4;      // 16-byte chunks per pixel row, K32 steps
    __shared__ __attribute__((aligned(16))) unsigned char xs[PMT * 16 * CIN * 2];   // LN'd input halo, [192][CIN] bf16
    __shared__ __attribute__((aligned(16))) unsigned char ts[TS_B];     // 1x1 output of one chunk, [180][64] fp16
    __shared__ float sV[2][CIN];
    // row & (CH - 1): conflict-free for 16 consecutive rows starting at ANY row (the z phase reads tile rows that start
    // at odd halo pixels)
    auto xs_off = [](int row, int chunk) -> int { return row * (CIN * 2) + ((chunk ^ (row & (CH - 1))) << 4); };
    // fp16 tap weights [9][Cdw/2] (+ bias [Cdw] at word 5 Cdw) of the depthwise conv, staged once: a global load issued after a phase's
    // stores cannot be waited for without waiting for those stores too (vmcnt retires in order)
    __shared__ __attribute__((aligned(16))) uint32_t sW[6 * (CIN == 64 ? 192 : 256)];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fg = lane >> 4;
    const int tiles_x = p.W / PT_W;
    const int64_t img = blockIdx.y;
    const bf16 *xin = p.x + img * p.H * p.W * p.ld_x + p.off_x;                 // wave-uniform image bases
    bf16 *zout = p.Cz > 0 ? p.out_z + img * p.H * p.W * p.ld_z + p.off_z : nullptr;
    bf16 *dwout = p.out_dw + img * p.H * p.W * p.ld_dw + p.off_dw;

    // ---- once per workgroup: modulation vectors, tap weights
    if (tid < CIN) {
        const float sc = 1.f + p.ln_scale[img * p.ln_ld + tid], sh = p.ln_shift[img * p.ln_ld + tid];
        const float g = p.ln_gamma ? p.ln_gamma[tid] : 1.f, be = p.ln_beta ? p.ln_beta[tid] : 0.f;
        sV[0][tid] = g * sc;
        sV[1][tid] = be * sc + sh;
    }
    for (int i = tid; i < 9 * p.Cdw / 2; i += 256) sW[i] = p.w_dw[i];         // [9][Cdw/2] fp16 channel pairs
    for (int i = tid; i < p.Cdw; i += 256) sW[5 * p.Cdw + i] = p.b_dw ? __builtin_bit_cast(uint32_t, p.b_dw[i]) : 0u;
    constexpr int NLD = (PMT * 16 * CH) / 256;        // 6 (12) chunks per thread cover all 192 rows
    const int v = tid & (CH - 1);
    // weight rows of a 32-channel group, permuted so that a lane ends up with 8 consecutive channels
    const int rperm = 8 * (fr >> 2) + (fr & 3);
    // The 1x1 weights are consumed in 32-row groups: first the Cz/32 pass-through groups (rows Cdw + 32g),
    // then two groups per depthwise chunk (rows 32g).  Group g+1 is loaded (L2) while group g is in the
    // MFMAs -- loaded right before use, each group exposed a full L2 round trip (~15k cycles per workgroup).
    const int nz = p.Cz / 32, ngroups = nz + p.Cdw / 32;
    bf16x8 wnext[2 * KS];                            // [row set a | b][ks]
    auto wload = [&](int g) {
        const int rb = g < nz ? p.Cdw + 32 * g : 32 * (g - nz);
        const bf16 *wr = p.w_pw + (int64_t)(rb + rperm) * CIN + fg * 8;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            wnext[ks] = *(const bf16x8 *)(wr + 32 * ks);
            wnext[KS + ks] = *(const bf16x8 *)(wr + 4 * CIN + 32 * ks);
        }
    };
    wload(0);
    u32x4 w2r[2] = {};                               // PROJ: this thread's two 16-byte pieces of Weff[img] ([64][64] bf16)
    if constexpr (PROJ) {
#pragma unroll
        for (int k = 0; k < 2; ++k) w2r[k] = *(const u32x4 *)(p.w2 + img * p.w2_bstride + (tid + k * 256) * 8);
    }
    lds_barrier();                                   // sV, sW
    const int tile0 = blockIdx.x * p.tpw;
    for (int tt = 0; tt < p.tpw; ++tt) {
    const int tile = tile0 + tt;
    if (tile >= p.ntiles) break;                     // workgroup-uniform
    const int ty0 = (tile / tiles_x) * PT_H, tx0 = (tile % tiles_x) * PT_W;
    // the halo geometry of a thread does not depend on the tile: an opaque copy of tid keeps it (and ~40 registers)
    // from being hoisted out of the tile loop (see pwdw_gram_kernel)
    int tl = tid;
    asm volatile("" : "+v"(tl));
    // ---- phase 0: halo load -> LayerNorm + modulate -> xs
    u32x4 raw[NLD];
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
        const int hp = (tl + k * 256) >> LCH;
        const int hy = div18(hp), hx = hp - hy * PH_X;
        const int yc = min(max(ty0 + hy - 1, 0), p.H - 1), xc = min(max(tx0 + hx - 1, 0), p.W - 1);
        // 32-bit element offsets from a per-image scalar base, 24-bit multiplies: the 64-bit form of this address
        // cost 3 v_mad_u64_u32 + 4 v_mul_lo_u32 (quarter rate) per load
        raw[k] = *(const u32x4 *)(xin + (__umul24(__umul24(yc, p.W) + xc, p.ld_x) + v * 8));
    }
    f32x2 g2[4], b2m[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        g2[j] = *(const f32x2 *)&sV[0][v * 8 + 2 * j];
        b2m[j] = *(const f32x2 *)&sV[1][v * 8 + 2 * j];
    }
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
        const int hp = (tl + k * 256) >> LCH;
        *(u32x4 *)(xs + xs_off(hp, v)) = fd_ln_mod_chunk<CH>(raw[k], g2, b2m, p.ln_eps);
    }
    // which of this lane's halo pixels (m-tile mt, row fr) lie inside the image: the depthwise conv
    // zero-pads ITS input, i.e. the 1x1 output -- not LN(0)
    uint32_t inside = 0;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int hp = (3 * (tl >> 6) + i) * 16 + (tl & 15);
        const int hy = div18(hp), hx = hp - hy * PH_X;
        const int yy = ty0 + hy - 1, xx = tx0 + hx - 1;
        if (hp < PHP && yy >= 0 && yy < p.H && xx >= 0 && xx < p.W) inside |= 1u << i;
    }
    lds_barrier();
    int gi = 0;

    // ---- phase Z: pass-through channels (z), interior pixels only, accumulators -> HBM
    if (p.Cz > 0) {
        bf16x8 xb[2][KS];                              // [tile row][ks]
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = (2 * wave + i + 1) * PH_X + 1 + fr;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) xb[i][ks] = *(const bf16x8 *)(xs + xs_off(row, ks * 4 + fg));
        }
        for (int ng = 0; ng < nz; ++ng) {
            bf16x8 wa[KS], wb[KS];
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) { wa[ks] = wnext[ks]; wb[ks] = wnext[KS + ks]; }
            ++gi;
            wload(gi < ngroups ? gi : 0);               // after the last group: the next tile's first
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    a0 = FD_MFMA16(wa[ks], xb[i][ks], a0, 0, 0, 0);
                    a1 = FD_MFMA16(wb[ks], xb[i][ks], a1, 0, 0, 0);
                }
                float val[8] = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
                fd_silu8(val);
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
        uint32_t wt[9][4], b2[4];
        float bs[8];
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const u32x4 w0 = *(const u32x4 *)(sW + t * (p.Cdw / 2) + c0 / 2);
            wt[t][0] = w0.x; wt[t][1] = w0.y; wt[t][2] = w0.z; wt[t][3] = w0.w;
        }
        load8((const float *)(sW + 5 * p.Cdw + c0), bs);
        // the bias is the fp16 accumulator's initial value (v_cvt_pkrtz: round toward zero, |b| > 65504 saturates)
#pragma unroll
        for (int j = 0; j < 4; ++j) b2[j] = pk_h2(bs[2 * j], bs[2 * j + 1]);
        // phase 1: t[hp][64 ch] = W_chunk . xn, 3 m-tiles per wave (fragments re-read per chunk: registers
        // are the scarce resource of this kernel, LDS bandwidth is not)
        bf16x8 xh[3][KS];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
                xh[i][ks] = *(const bf16x8 *)(xs + xs_off((3 * wave + i) * 16 + fr, ks * 4 + fg));
#pragma unroll
        for (int ng = 0; ng < 2; ++ng) {
            bf16x8 wa[KS], wb[KS];
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) { wa[ks] = wnext[ks]; wb[ks] = wnext[KS + ks]; }
            ++gi;
            wload(gi < ngroups ? gi : 0);               // after the last group: the next tile's first
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    a0 = FD_MFMA16(wa[ks], xh[i][ks], a0, 0, 0, 0);
                    a1 = FD_MFMA16(wb[ks], xh[i][ks], a1, 0, 0, 0);
                }
                const int hp = (3 * wave + i) * 16 + fr;
                const bool in = (inside >> i) & 1;
                // fp16 pairs (dw_row_f16); out-of-image halo pixels are the depthwise conv's zero padding
                u32x4 o = {pk_h2(a0[0], a0[1]), pk_h2(a0[2], a0[3]), pk_h2(a1[0], a1[1]), pk_h2(a1[2], a1[3])};
                if (!in) o = (u32x4){0, 0, 0, 0};
                if (hp < PHP) *(u32x4 *)(ts + ts_off(hp, 4 * ng + fg)) = o;
            }
        }
        lds_barrier();
        if constexpr (PROJ) {
            // xs is dead (this chunk's 1x1 was its last reader and every wave is past the barrier): Weff over its rows 128..191
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int idx = tid + k * 256;
                *(u32x4 *)(xs + xs_off(128 + (idx >> 3), idx & 7)) = w2r[k];
            }
        }
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
            uint32_t o2[4];
            dw_row_f16(win, rr, wt, b2, o2);
            float acc[8];
            unpack_h8(o2, acc);
            if (p.dw_silu) fd_silu8(acc);
            if constexpr (PROJ) {
                const u32x4 o = {fd_pack_bf16(f32x2{acc[0], acc[1]}), fd_pack_bf16(f32x2{acc[2], acc[3]}),
                                 fd_pack_bf16(f32x2{acc[4], acc[5]}), fd_pack_bf16(f32x2{acc[6], acc[7]})};
                *(u32x4 *)(xs + xs_off((4 * rh + rr) * PT_W + px, cv)) = o;       // [128 tile pixels][64 ch]
            } else {
                const int y = ty0 + 4 * rh + rr, x = tx0 + px;
                store8(dwout + (__umul24(__umul24(y, p.W) + x, p.ld_dw) + c0), acc);
            }
        }
        lds_barrier();                               // ts is rewritten by the next chunk (xs by the next tile)
    }
    if constexpr (PROJ) {
        // ---- second 1x1 on the tile + gated residual: wave w owns tile rows 2w, 2w + 1 (16 pixels each)
        bf16x8 vb[2][KS];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) vb[i][ks] = *(const bf16x8 *)(xs + xs_off((2 * wave + i) * PT_W + fr, ks * 4 + fg));
        bf16 *o2 = p.out2 + img * p.H * p.W * p.ld_o2 + p.off_o2;
#pragma unroll
        for (int ng = 0; ng < 2; ++ng) {
            bf16x8 wa[KS], wb[KS];
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                wa[ks] = *(const bf16x8 *)(xs + xs_off(128 + 32 * ng + rperm, ks * 4 + fg));
                wb[ks] = *(const bf16x8 *)(xs + xs_off(128 + 32 * ng + rperm + 4, ks * 4 + fg));
            }
            const int n0 = 32 * ng + 8 * fg;
            float gt[8];
            load8(p.gate + img * p.gate_ld + n0, gt);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    a0 = FD_MFMA16(wa[ks], vb[i][ks], a0, 0, 0, 0);
                    a1 = FD_MFMA16(wb[ks], vb[i][ks], a1, 0, 0, 0);
                }
                const int y = ty0 + 2 * wave + i, x = tx0 + fr;
                const int pix = __umul24(y, p.W) + x;
                float rs[8];
                load8(xin + (__umul24(pix, p.ld_x) + n0), rs);                     // the block input again (L2)
                float val[8] = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
#pragma unroll
                for (int e = 0; e < 8; ++e) val[e] = __builtin_fmaf(gt[e], val[e], rs[e]);
                store8(o2 + (__umul24(pix, p.ld_o2) + n0), val);
            }
        }
        lds_barrier();                               // xs is rewritten by the next tile
    }
    }
}


// ---------------------------------------------------------------------------------------------------------------
// qkv -> qkv_dwconv -> L2 norms + q k^T (src/DADiff.py:266-276) in ONE pass: q and k never reach HBM.
// The attention of TransposedAttention is a 32x32 matrix per head whose reduction runs over ALL pixels, so q and k
// are consumed by nothing but their Gram product and their norms.  The unfused sequence wrote them (2/3 of the 192
// channels the qkv variant of pwdw_kernel stores) and gram_kernel read them back: 4 of the ~26 tensor passes of a
// 64-channel Mamba block.  Here a workgroup walks `tpw` consecutive 8x16 tiles; per tile the chunks run in the order
// v, q, k:  v goes to HBM as before; q's depthwise output is parked in registers (16 packed words) while k's 1x1
// runs, then written as a [128 px][64 ch] bf16 tile over xs (the LayerNorm'd halo is dead once k's 1x1 has read
// it); k's output goes over ts the same way; the 32x32 Gram of both heads and the sums of squares are 16 bf16
// MFMAs per wave whose operands (8 consecutive PIXELS of one channel per lane) come out of those pixel-major tiles
// through the transposing LDS read ds_read_b64_tr_b16 -- no 2-byte LDS scatter as in gram_kernel (59 % bank
// conflicts).  LDS stays at the 52 KB of pwdw_kernel (3 workgroups per CU).  The norms are the diagonals of q q^T and
// k k^T from the same fragments (the matrix pipe is idle in this kernel), i.e. norms and Gram see the same
// bf16-rounded values, as in the unfused path.  Accumulators persist over the workgroup's tiles; one partial of
// 2 x (1024 + 64) floats per workgroup, reduced in fixed order by fd_chan_attn_weff.
typedef __attribute__((ext_vector_type(4))) short pd_s16x4;
typedef __attribute__((ext_vector_type(8))) short pd_s16x8;

struct PwGramParams {
    const bf16 *x; int ld_x, off_x;
    const float *ln_gamma, *ln_beta, *ln_shift, *ln_scale; int ln_ld; float ln_eps;
    const bf16 *w_pw;                 // [192][64]: q | k | v rows
    const uint32_t *w_dw;             // [9][96] fp16 channel pairs
    bf16 *out_v; int ld_v, off_v;
    float *part; int nblk;            // [B][2 heads][nblk][1024 + 64]
    int H, W, tpw, ntiles;
};

// q / k tile: row = tile pixel (128 B = 64 channels), 16-byte chunk index XORed with an EVEN value so that the two
// chunks of a 16-channel fragment column stay adjacent; the 8 rows one 32-lane half of a transposing read touches
// (pixels 8g .. 8g+3 of two k-groups) land on 8 distinct 8-bank slots
__device__ __forceinline__ int gt_swz(int pix) { return (((pix >> 1) & 1) << 1) | (((pix >> 3) & 1) << 2); }
__device__ __forceinline__ int gt_off(int pix, int chunk) { return pix * 128 + ((chunk ^ gt_swz(pix)) << 4); }

__device__ __forceinline__ pd_s16x4 lds_tr16(const unsigned char *p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) pd_s16x4 *)p);
}

__global__ __launch_bounds__(256, 3) void pwdw_gram_kernel(const PwGramParams p) {
    __shared__ __attribute__((aligned(16))) unsigned char xs[XS_B];
    __shared__ __attribute__((aligned(16))) unsigned char ts[TS_B];
    __shared__ float sV[2][64];
    __shared__ __attribute__((aligned(16))) uint32_t sW[5 * 192];
    constexpr int CDW = 192;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fg = lane >> 4;
    const int tiles_x = p.W / PT_W;
    const int64_t img = blockIdx.y;
    const bf16 *xin = p.x + img * p.H * p.W * p.ld_x + p.off_x;
    bf16 *vout = p.out_v + img * p.H * p.W * p.ld_v + p.off_v;
    if (tid < 64) {
        const float sc = 1.f + p.ln_scale[img * p.ln_ld + tid], sh = p.ln_shift[img * p.ln_ld + tid];
        const float g = p.ln_gamma ? p.ln_gamma[tid] : 1.f, be = p.ln_beta ? p.ln_beta[tid] : 0.f;
        sV[0][tid] = g * sc;
        sV[1][tid] = be * sc + sh;
    }
    for (int i = tid; i < 9 * CDW / 2; i += 256) sW[i] = p.w_dw[i];           // [9][96] fp16 channel pairs
    constexpr int NLD = (PMT * 16 * 8) / 256;
    const int v = tid & 7;
    const int rperm = 8 * (fr >> 2) + (fr & 3);
    // 1x1 weight rows in 32-row groups, in the order the chunks run: v (rows 128..191), q (0..63), k (64..127)
    bf16x8 wnext[4];
    auto wload = [&](int si) {
        const int rb = si < 2 ? 128 + 32 * si : 32 * (si - 2);
        const bf16 *wr = p.w_pw + (int64_t)(rb + rperm) * 64 + fg * 8;
        wnext[0] = *(const bf16x8 *)(wr);
        wnext[1] = *(const bf16x8 *)(wr + 32);
        wnext[2] = *(const bf16x8 *)(wr + 4 * 64);
        wnext[3] = *(const bf16x8 *)(wr + 4 * 64 + 32);
    };
    // phase-2 roles: 8-channel vector cv of tile column px, rows 4*rh .. 4*rh+3
    const int cv = tid & 7, px = (tid >> 3) & 15, rh = tid >> 7;
    int toff[6][3];
#pragma unroll
    for (int r = 0; r < 6; ++r)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) toff[r][dx] = ts_off((4 * rh + r) * PH_X + px + dx, cv);
    // Gram roles: wave = (head, 16-row block of q channels); transposing-read lane roles
    const int hd = wave >> 1, mb = wave & 1;
    const int tq = (lane >> 2) & 3, tp = lane & 3;
    f32x4 gacc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}, gnq = {0.f, 0.f, 0.f, 0.f}, gnk = {0.f, 0.f, 0.f, 0.f};
    // out_v == NULL (round 4): q and k only -- v is recomputed by the kernel that consumes it (pwdw_kernel<64, PROJ>)
    const int ci0 = p.out_v ? 0 : 1;
    wload(2 * ci0);
    lds_barrier();                                   // sV, sW

    const int tile0 = blockIdx.x * p.tpw;
    for (int tt = 0; tt < p.tpw; ++tt) {
        const int tile = tile0 + tt;
        if (tile >= p.ntiles) break;                 // workgroup-uniform
        const int ty0 = (tile / tiles_x) * PT_H, tx0 = (tile % tiles_x) * PT_W;
        // the halo geometry of a thread does not depend on the tile; hoisted out of this loop it would be ~40 more
        // live registers (spilled: scratch reloads in front of every tile).  An opaque copy of tid keeps it inside.
        int tl = tid;
        asm volatile("" : "+v"(tl));
        // ---- phase 0: halo load -> LayerNorm + modulate -> xs
        u32x4 raw[NLD];
#pragma unroll
        for (int k = 0; k < NLD; ++k) {
            const int hp = (tl + k * 256) >> 3;
            const int hy = div18(hp), hx = hp - hy * PH_X;
            const int yc = min(max(ty0 + hy - 1, 0), p.H - 1), xc = min(max(tx0 + hx - 1, 0), p.W - 1);
            raw[k] = *(const u32x4 *)(xin + (__umul24(__umul24(yc, p.W) + xc, p.ld_x) + v * 8));
        }
        f32x2 g2[4], b2m[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            g2[j] = *(const f32x2 *)&sV[0][v * 8 + 2 * j];
            b2m[j] = *(const f32x2 *)&sV[1][v * 8 + 2 * j];
        }
#pragma unroll
        for (int k = 0; k < NLD; ++k) {
            const int hp = (tl + k * 256) >> 3;
            *(u32x4 *)(xs + xs_off(hp, v)) = fd_ln_mod_chunk<8>(raw[k], g2, b2m, p.ln_eps);
        }
        uint32_t inside = 0;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int hp = (3 * (tl >> 6) + i) * 16 + (tl & 15);
            const int hy = div18(hp), hx = hp - hy * PH_X;
            const int yy = ty0 + hy - 1, xx = tx0 + hx - 1;
            if (hp < PHP && yy >= 0 && yy < p.H && xx >= 0 && xx < p.W) inside |= 1u << i;
        }
        lds_barrier();

        u32x4 pk[4];                                   // q, then k: this thread's 4 rows x 8 channels, bf16
#pragma unroll 1
        for (int ci = ci0; ci < 3; ++ci) {
            const int ch = ci == 0 ? 2 : ci - 1;       // v, q, k
            // phase 1: t[hp][64 ch] = W_chunk . xn, 3 m-tiles per wave
            bf16x8 xh[3][2];
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
                    xh[i][ks] = *(const bf16x8 *)(xs + xs_off((3 * wave + i) * 16 + fr, ks * 4 + fg));
#pragma unroll
            for (int ng = 0; ng < 2; ++ng) {
                const bf16x8 wa[2] = {wnext[0], wnext[1]}, wb[2] = {wnext[2], wnext[3]};
                const int si = 2 * ci + ng + 1;
                wload(si < 6 ? si : 2 * ci0);          // after k: the next tile's first group
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) {
                        a0 = FD_MFMA16(wa[ks], xh[i][ks], a0, 0, 0, 0);
                        a1 = FD_MFMA16(wb[ks], xh[i][ks], a1, 0, 0, 0);
                    }
                    const int hp = (3 * wave + i) * 16 + fr;
                    const bool in = (inside >> i) & 1;
                    u32x4 o = {pk_h2(a0[0], a0[1]), pk_h2(a0[2], a0[3]), pk_h2(a1[0], a1[1]), pk_h2(a1[2], a1[3])};
                    if (!in) o = (u32x4){0, 0, 0, 0};
                    if (hp < PHP) *(u32x4 *)(ts + ts_off(hp, 4 * ng + fg)) = o;
                }
            }
            lds_barrier();
            if (ci == 2) {
                // xs is dead (k's 1x1 was its last reader, and every wave is past the barrier): park q there
#pragma unroll
                for (int rr = 0; rr < 4; ++rr)
                    *(u32x4 *)(xs + gt_off((4 * rh + rr) * PT_W + px, cv)) = pk[rr];
            }
            // phase 2: depthwise 3x3 on ts (packed fp16, dw_row_f16)
            const int c0 = 64 * ch + cv * 8;
            uint32_t wt[9][4];
            const uint32_t b2[4] = {0u, 0u, 0u, 0u};
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const u32x4 w0 = *(const u32x4 *)(sW + t * (CDW / 2) + c0 / 2);
                wt[t][0] = w0.x; wt[t][1] = w0.y; wt[t][2] = w0.z; wt[t][3] = w0.w;
            }
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
                uint32_t o2[4];
                dw_row_f16(win, rr, wt, b2, o2);
                if (ci == 0) {
                    float acc[8];
                    unpack_h8(o2, acc);
                    const int y = ty0 + 4 * rh + rr, x = tx0 + px;
                    store8(vout + (__umul24(__umul24(y, p.W) + x, p.ld_v) + cv * 8), acc);
                } else {
                    pk[rr] = (u32x4){o2[0], o2[1], o2[2], o2[3]};      // q / k stay fp16: the Gram runs on the f16 MFMA
                }
            }
            lds_barrier();                               // every read of ts is done
        }
        // k tile over ts, then the Gram of this tile
#pragma unroll
        for (int rr = 0; rr < 4; ++rr)
            *(u32x4 *)(ts + gt_off((4 * rh + rr) * PT_W + px, cv)) = pk[rr];
        lds_barrier();
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            pd_s16x4 a[2], b0[2], b1[2];
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                const int pix = 32 * ks + 8 * fg + 4 * hf + tq;
                const int rowb = pix * 128, sw = gt_swz(pix), sub = 8 * (tp & 1);
                const int cq = hd * 4 + mb * 2 + (tp >> 1), ck0 = hd * 4 + (tp >> 1), ck1 = ck0 + 2;
                a[hf] = lds_tr16(xs + rowb + ((cq ^ sw) << 4) + sub);
                b0[hf] = lds_tr16(ts + rowb + ((ck0 ^ sw) << 4) + sub);
                b1[hf] = lds_tr16(ts + rowb + ((ck1 ^ sw) << 4) + sub);
            }
            // q, k tiles hold fp16 (the depthwise accumulators as they are): the f16 MFMA, same rate as bf16
            const pd_h8 A = __builtin_bit_cast(pd_h8, __builtin_shufflevector(a[0], a[1], 0, 1, 2, 3, 4, 5, 6, 7));
            const pd_h8 B0 = __builtin_bit_cast(pd_h8, __builtin_shufflevector(b0[0], b0[1], 0, 1, 2, 3, 4, 5, 6, 7));
            const pd_h8 B1 = __builtin_bit_cast(pd_h8, __builtin_shufflevector(b1[0], b1[1], 0, 1, 2, 3, 4, 5, 6, 7));
            gacc[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A, B0, gacc[0], 0, 0, 0);
            gacc[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A, B1, gacc[1], 0, 0, 0);
            gnq = __builtin_amdgcn_mfma_f32_16x16x32_f16(A, A, gnq, 0, 0, 0);
            const pd_h8 Bm = mb ? B1 : B0;
            gnk = __builtin_amdgcn_mfma_f32_16x16x32_f16(Bm, Bm, gnk, 0, 0, 0);
        }
        lds_barrier();                                   // the next tile's phase 0 rewrites xs
    }
    // partial of this workgroup: per head [32 q channels][32 k channels], then sum q^2 (32), sum k^2 (32) -- staged
    // in LDS (ts is free: the loop ends with a barrier) so that the 8.7 KB leave as 16-byte stores
    float *sp = (float *)ts + hd * (1024 + 64);
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int e = 0; e < 4; ++e) sp[(mb * 16 + 4 * fg + e) * 32 + nb * 16 + fr] = gacc[nb][e];
    if ((fr >> 2) == fg) {                               // lanes that hold a diagonal element: row 4*fg + e == column fr
        const int e = fr & 3;
        const float nq = e == 0 ? gnq[0] : (e == 1 ? gnq[1] : (e == 2 ? gnq[2] : gnq[3]));
        const float nk = e == 0 ? gnk[0] : (e == 1 ? gnk[1] : (e == 2 ? gnk[2] : gnk[3]));
        sp[1024 + mb * 16 + fr] = nq;
        sp[1024 + 32 + mb * 16 + fr] = nk;
    }
    lds_barrier();
    for (int i = tid; i < 2 * (1024 + 64) / 4; i += 256) {
        const int h2 = i / ((1024 + 64) / 4), r = i - h2 * ((1024 + 64) / 4);
        float *out = p.part + ((img * 2 + h2) * p.nblk + blockIdx.x) * (1024 + 64);
        *(f32x4 *)(out + 4 * r) = *(const f32x4 *)((const float *)ts + h2 * (1024 + 64) + 4 * r);
    }
}


// ---------------------------------------------------------------------------------------------------------------
// qkv_dwconv -> L2 norms + q k^T for the WIDER Mamba blocks (C >= 128: the 1x1 qkv stays a separate GEMM there), in
// one pass over q and k: the depthwise outputs of q and k go straight into the Gram (same tile images, transposing
// reads and MFMAs as pwdw_gram_kernel) and never reach HBM; v keeps the plain depthwise kernel.  Unfused, q and k
// were written by dwconv3x3_bf16_kernel and read back by gram_kernel: 4C of the 9C channel passes of this branch.
// A workgroup owns (tile group, pair of heads = 64 q channels + the matching 64 k channels):
//   q halo (10 x 18 x 64 ch) -> LDS; the k halo's global loads are issued right away (registers);
//   depthwise(q) -> registers -> Q tile;   k halo -> LDS (the q halo is dead);   depthwise(k) -> registers ->
//   K tile written OVER the halo;   16 MFMAs per wave;  next tile.   39 KB of LDS: 4 workgroups per CU.
struct DwGramParams {
    const bf16 *qkv; int ld;          // [B,H,W,ld]: q at channel 0, k at channel C
    int C;
    const uint32_t *w_dw;             // [9][3C/2] fp16 channel pairs (fd_pw_dw3x3's layout)
    float *part; int nblk;            // [B][C/32 heads][nblk][1024 + 64]
    int H, W, tpw, ntiles;
};

__global__ __launch_bounds__(256, 3) void dwconv_gram_kernel(const DwGramParams p) {
    __shared__ __attribute__((aligned(16))) unsigned char hs[TS_B];       // input halo [180][64] bf16, later the K tile
    __shared__ __attribute__((aligned(16))) unsigned char qs[128 * 128];  // Q tile [128 px][64 ch]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fg = lane >> 4;
    const int tiles_x = p.W / PT_W;
    const int hp2 = blockIdx.y;                         // pair of heads: q channels [64 hp2, +64), k channels C + the same
    const int64_t img = blockIdx.z;
    const bf16 *base = p.qkv + img * p.H * p.W * p.ld;
    const int cv = tid & 7, px = (tid >> 3) & 15, rh = tid >> 7;
    const int C3 = 3 * p.C;
    // fp16 tap weights (channel pairs) of the 64 q and the 64 k channels, staged once: [2][9][32] words
    __shared__ __attribute__((aligned(16))) uint32_t sWd[2 * 9 * 32];
    for (int i = tid; i < 2 * 9 * 32; i += 256) {
        const int qk = i / 288, t = (i % 288) / 32, c2 = i % 32;
        sWd[i] = p.w_dw[t * (C3 / 2) + (qk * p.C + 64 * hp2) / 2 + c2];
    }
    int toff[6][3];
#pragma unroll
    for (int r = 0; r < 6; ++r)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) toff[r][dx] = ts_off((4 * rh + r) * PH_X + px + dx, cv);
    const int hd = wave >> 1, mb = wave & 1;
    const int tq = (lane >> 2) & 3, tp = lane & 3;
    f32x4 gacc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}, gnq = {0.f, 0.f, 0.f, 0.f}, gnk = {0.f, 0.f, 0.f, 0.f};
    constexpr int NLD = (PHP * 8 + 255) / 256;          // 6 halo chunks per thread (180 px x 8)

    // one depthwise pass over the halo image in hs: this thread's 4 rows x 8 channels, packed bf16
    auto dw = [&](int qk, u32x4 (&pk)[4]) {
        uint32_t wt[9][4];
        const uint32_t b2[4] = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const u32x4 a0 = *(const u32x4 *)(sWd + (qk * 9 + t) * 32 + cv * 4);
            wt[t][0] = a0.x; wt[t][1] = a0.y; wt[t][2] = a0.z; wt[t][3] = a0.w;
        }
        uint32_t win[3][3][4];
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const u32x4 t4 = *(const u32x4 *)(hs + toff[s][dx]);
                win[s][dx][0] = t4.x; win[s][dx][1] = t4.y; win[s][dx][2] = t4.z; win[s][dx][3] = t4.w;
            }
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const u32x4 t4 = *(const u32x4 *)(hs + toff[rr + 2][dx]);
                uint32_t *wr_ = win[(rr + 2) % 3][dx];
                wr_[0] = t4.x; wr_[1] = t4.y; wr_[2] = t4.z; wr_[3] = t4.w;
            }
            uint32_t o2[4];
            dw_row_f16(win, rr, wt, b2, o2);
            pk[rr] = (u32x4){o2[0], o2[1], o2[2], o2[3]};
        }
    };

    const int tile0 = blockIdx.x * p.tpw;
    for (int tt = 0; tt < p.tpw; ++tt) {
        const int tile = tile0 + tt;
        if (tile >= p.ntiles) break;                    // workgroup-uniform
        const int ty0 = (tile / tiles_x) * PT_H, tx0 = (tile % tiles_x) * PT_W;
        int tl = tid;
        asm volatile("" : "+v"(tl));                    // keep the halo geometry inside the loop (see pwdw_gram_kernel)
        u32x4 rq[NLD], rk[NLD];
        uint32_t okm = 0;
#pragma unroll
        for (int k = 0; k < NLD; ++k) {
            const int idx = tl + k * 256;
            const int hp = min(idx >> 3, PHP - 1), v = idx & 7;
            const int hy = div18(hp), hx = hp - hy * PH_X;
            const int yy = ty0 + hy - 1, xx = tx0 + hx - 1;
            if (yy >= 0 && yy < p.H && xx >= 0 && xx < p.W) okm |= 1u << k;
            const int yc = min(max(yy, 0), p.H - 1), xc = min(max(xx, 0), p.W - 1);
            const bf16 *src = base + (__umul24(__umul24(yc, p.W) + xc, p.ld) + 64 * hp2 + v * 8);
            rq[k] = *(const u32x4 *)src;
            rk[k] = *(const u32x4 *)(src + p.C);
        }
        auto halo_store = [&](const u32x4 (&r)[NLD]) {
#pragma unroll
            for (int k = 0; k < NLD; ++k) {
                const int idx = tl + k * 256, hp = idx >> 3;
                const u32x4 z4 = {0, 0, 0, 0};
                // bf16 (HBM) -> fp16 pairs (dw_row_f16): the 8-bit MANTISSA of a bf16 always fits fp16's 11 bits, the
                // exponent does not -- magnitudes below 6.1e-5 become fp16 subnormals, below ~3e-8 zero, above 65504
                // saturate.  q and k are L2-normalised per channel afterwards (src/DADiff.py:273-274), so the host packs
                // their 1x1 rows and depthwise taps with a per-channel power-of-two scale that brings them to O(1)
                // (engine.py: _qk_prescale; the norm cancels it exactly): in range for any checkpoint.
                const u32x4 rv = r[k];
                const uint32_t rw[4] = {rv.x, rv.y, rv.z, rv.w};
                uint32_t hw4[4];
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    hw4[e] = h16_to_h2(rw[e]);
                const u32x4 h4 = {hw4[0], hw4[1], hw4[2], hw4[3]};
                if (hp < PHP) *(u32x4 *)(hs + ts_off(hp, idx & 7)) = ((okm >> k) & 1) ? h4 : z4;
            }
        };
        u32x4 pk[4];
        halo_store(rq);
        lds_barrier();
        dw(0, pk);
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) *(u32x4 *)(qs + gt_off((4 * rh + rr) * PT_W + px, cv)) = pk[rr];
        lds_barrier();                                  // every read of the q halo is done
        halo_store(rk);
        lds_barrier();
        dw(1, pk);
        lds_barrier();                                  // every read of the k halo is done: the K tile goes over it
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) *(u32x4 *)(hs + gt_off((4 * rh + rr) * PT_W + px, cv)) = pk[rr];
        lds_barrier();
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            pd_s16x4 a[2], b0[2], b1[2];
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                const int pix = 32 * ks + 8 * fg + 4 * hf + tq;
                const int rowb = pix * 128, sw = gt_swz(pix), sub = 8 * (tp & 1);
                const int cq = hd * 4 + mb * 2 + (tp >> 1), ck0 = hd * 4 + (tp >> 1), ck1 = ck0 + 2;
                a[hf] = lds_tr16(qs + rowb + ((cq ^ sw) << 4) + sub);
                b0[hf] = lds_tr16(hs + rowb + ((ck0 ^ sw) << 4) + sub);
                b1[hf] = lds_tr16(hs + rowb + ((ck1 ^ sw) << 4) + sub);
            }
            // q, k tiles hold fp16 (the depthwise accumulators as they are): the f16 MFMA, same rate as bf16
            const pd_h8 A = __builtin_bit_cast(pd_h8, __builtin_shufflevector(a[0], a[1], 0, 1, 2, 3, 4, 5, 6, 7));
            const pd_h8 B0 = __builtin_bit_cast(pd_h8, __builtin_shufflevector(b0[0], b0[1], 0, 1, 2, 3, 4, 5, 6, 7));
            const pd_h8 B1 = __builtin_bit_cast(pd_h8, __builtin_shufflevector(b1[0], b1[1], 0, 1, 2, 3, 4, 5, 6, 7));
            gacc[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A, B0, gacc[0], 0, 0, 0);
            gacc[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A, B1, gacc[1], 0, 0, 0);
            gnq = __builtin_amdgcn_mfma_f32_16x16x32_f16(A, A, gnq, 0, 0, 0);
            const pd_h8 Bm = mb ? B1 : B0;
            gnk = __builtin_amdgcn_mfma_f32_16x16x32_f16(Bm, Bm, gnk, 0, 0, 0);
        }
        lds_barrier();                                  // the next tile's halo rewrites hs
    }
    float *sp = (float *)hs + hd * (1024 + 64);
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int e = 0; e < 4; ++e) sp[(mb * 16 + 4 * fg + e) * 32 + nb * 16 + fr] = gacc[nb][e];
    if ((fr >> 2) == fg) {
        const int e = fr & 3;
        const float nq = e == 0 ? gnq[0] : (e == 1 ? gnq[1] : (e == 2 ? gnq[2] : gnq[3]));
        const float nk = e == 0 ? gnk[0] : (e == 1 ? gnk[1] : (e == 2 ? gnk[2] : gnk[3]));
        sp[1024 + mb * 16 + fr] = nq;
        sp[1024 + 32 + mb * 16 + fr] = nk;
    }
    lds_barrier();
    const int heads = p.C / 32;
    for (int i = tid; i < 2 * (1024 + 64) / 4; i += 256) {
        const int h2 = i / ((1024 + 64) / 4), r = i - h2 * ((1024 + 64) / 4);
        float *out = p.part + ((img * heads + 2 * hp2 + h2) * p.nblk + blockIdx.x) * (1024 + 64);
        *(f32x4 *)(out + 4 * r) = *(const f32x4 *)((const float *)hs + h2 * (1024 + 64) + 4 * r);
    }
}

}  // namespace

extern "C" int fd_pw_dw3x3_ok(int dtype_opts, int Cin, int Cdw, int Cz, int H, int W) {
    // `dtype | FD_OPT_LOW_LATENCY`: the kernel set for ONE slice keeps the C = 128 block on row-GEMM + depthwise (the fused
    // 128-channel form has two workgroups per CU and too few tiles for a lone slice: 155 -> 153 ms per 50-step slice)
    const int dtype = dtype_opts & 0xff;
    const bool no128 = fd_dev(FD_DEV_NO_PWDW128);        // development switch
    const int minpix = fd_dev(FD_DEV_PWDW_MINPIX);    // development
    const bool c64 = Cin == 64 && Cdw <= 192, c128 = Cin == 128 && Cdw <= 256 && Cz <= 256 && !no128 && !(dtype_opts & FD_OPT_LOW_LATENCY);
    return dtype == FD_BF16 && (c64 || c128) && Cdw > 0 && Cdw % 64 == 0 && Cz >= 0 && Cz % 32 == 0 && H % PT_H == 0 &&
           W % PT_W == 0 && (int64_t)H * W >= minpix && (int64_t)H * W * 256 < (1ll << 31);   // 32-bit element offsets
}

extern "C" int fd_pw_dw3x3(int dtype, const void *x, int ld_x, int off_x, int Cin, const float *ln_gamma,
                           const float *ln_beta, float ln_eps, const float *ln_shift, const float *ln_scale,
                           int ln_ld, const void *w_pw, int Cdw, const uint32_t *w_dw, const float *b_dw,
                           int dw_silu, void *out_dw, int ld_dw, int off_dw, int Cz, void *out_z, int ld_z,
                           int off_z, int B, int H, int W, void *stream) {
    dtype &= 0xff;
    FD_REQUIRE(fd_pw_dw3x3_ok(dtype, Cin, Cdw, Cz, H, W),
               "fd_pw_dw3x3: unsupported shape (bf16, Cin=64|128, Cdw%%64, Cz%%32, H%%8, W%%16, >= 32768 px): "
               "Cin=%d Cdw=%d Cz=%d H=%d W=%d", Cin, Cdw, Cz, H, W);
    FD_REQUIRE(x && ln_shift && ln_scale && w_pw && w_dw && out_dw && (Cz == 0 || out_z), "fd_pw_dw3x3: null pointer");
    FD_REQUIRE(ld_x % 8 == 0 && off_x % 8 == 0 && ld_dw % 8 == 0 && off_dw % 8 == 0 && ld_z % 8 == 0 && off_z % 8 == 0,
               "fd_pw_dw3x3: strides / offsets must be multiples of 8 channels");
    FD_REQUIRE(((uintptr_t)w_dw & 15) == 0 && (!b_dw || ((uintptr_t)b_dw & 15) == 0), "fd_pw_dw3x3: weights must be 16-byte aligned");
    PwDwParams p = {};
    p.x = (const bf16 *)x; p.ld_x = ld_x; p.off_x = off_x;
    p.ln_gamma = ln_gamma; p.ln_beta = ln_beta; p.ln_shift = ln_shift; p.ln_scale = ln_scale; p.ln_ld = ln_ld; p.ln_eps = ln_eps;
    p.w_pw = (const bf16 *)w_pw;
    p.Cdw = Cdw; p.w_dw = w_dw; p.b_dw = b_dw; p.dw_silu = dw_silu;
    p.out_dw = (bf16 *)out_dw; p.ld_dw = ld_dw; p.off_dw = off_dw;
    p.Cz = Cz; p.out_z = (bf16 *)out_z; p.ld_z = ld_z; p.off_z = off_z;
    p.H = H; p.W = W;
    p.ntiles = (H / PT_H) * (W / PT_W);
    const int tpw_env = fd_dev(FD_DEV_PWDW_TPW);
    // (4 consecutive tiles per workgroup once the batch fills the chip; ONE for a lone slice: 154.6 -> 152.8 ms per 50-step
    //  slice at batch 1.  Tiles are independent: the split does not touch the results)
    p.tpw = tpw_env > 0 ? tpw_env : (B >= 4 ? 4 : 1);
    dim3 grid((p.ntiles + p.tpw - 1) / p.tpw, B), block(256);
    const size_t pad = fd_occ_pad(FD_DEV_PAD_PWDW);
    if (Cin == 64) hipLaunchKernelGGL(pwdw_kernel<64>, grid, block, pad, (hipStream_t)stream, p);
    else hipLaunchKernelGGL(pwdw_kernel<128>, grid, block, pad, (hipStream_t)stream, p);
    FD_LAUNCH_OK("fd_pw_dw3x3");
    return FD_OK;
}

// The v branch of the 64-channel TransposedAttention through project_out and the gated residual (pwdw_kernel<64, PROJ>):
//   out = x + gate[b] . (w2[b] . dwconv3x3(w_pw . (LN(x) (1 + scale[b]) + shift[b])))
// w_pw [64][64] (the v rows of qkv.weight), w_dw [9][32] fp16 channel pairs (the v taps of qkv_dwconv), w2 [B][64][64]
// (fd_chan_attn_weff's output).  With fd_pw_dw3x3_gram(out_v = NULL) in front, v never reaches HBM.
extern "C" int fd_pw_dw3x3_proj_ok(int dtype_opts, int Cin, int H, int W) {
    const bool off = fd_dev(FD_DEV_NO_PWDW_PROJ);      // development switch
    // (`dtype | FD_OPT_LOW_LATENCY`: for ONE slice the row-GEMM on the stored v is the shorter chain -- 151.7 against
    //  152.7 ms per 50-step slice at batch 1; the throughput set gains 0.8 % per batch-8 forward with this kernel)
    return !off && !(dtype_opts & FD_OPT_LOW_LATENCY) && Cin == 64 && fd_pw_dw3x3_ok(dtype_opts, Cin, 64, 0, H, W);
}

extern "C" int fd_pw_dw3x3_proj(int dtype, const void *x, int ld_x, int off_x, int Cin, const float *ln_gamma,
                                const float *ln_beta, float ln_eps, const float *ln_shift, const float *ln_scale,
                                int ln_ld, const void *w_pw, const uint32_t *w_dw, const void *w2, const float *gate,
                                int gate_ld, void *out, int ld_o, int off_o, int B, int H, int W, void *stream) {
    FD_REQUIRE(fd_pw_dw3x3_proj_ok(dtype & 0xff, Cin, H, W),
               "fd_pw_dw3x3_proj: unsupported shape (bf16, Cin=64, H%%8, W%%16, >= 32768 px): Cin=%d H=%d W=%d", Cin, H, W);
    FD_REQUIRE(x && ln_shift && ln_scale && w_pw && w_dw && w2 && gate && out, "fd_pw_dw3x3_proj: null pointer");
    FD_REQUIRE(ld_x % 8 == 0 && off_x % 8 == 0 && ld_o % 8 == 0 && off_o % 8 == 0,
               "fd_pw_dw3x3_proj: strides / offsets must be multiples of 8 channels");
    FD_REQUIRE(((uintptr_t)w_dw & 15) == 0 && ((uintptr_t)w2 & 15) == 0 && ((uintptr_t)gate & 15) == 0 && gate_ld % 4 == 0,
               "fd_pw_dw3x3_proj: weights / gate must be 16-byte aligned");
    PwDwParams p = {};
    p.x = (const bf16 *)x; p.ld_x = ld_x; p.off_x = off_x;
    p.ln_gamma = ln_gamma; p.ln_beta = ln_beta; p.ln_shift = ln_shift; p.ln_scale = ln_scale; p.ln_ld = ln_ld; p.ln_eps = ln_eps;
    p.w_pw = (const bf16 *)w_pw;
    p.Cdw = 64; p.w_dw = w_dw; p.b_dw = nullptr; p.dw_silu = 0;
    p.Cz = 0;
    p.w2 = (const bf16 *)w2; p.w2_bstride = 64 * 64; p.gate = gate; p.gate_ld = gate_ld;
    p.out2 = (bf16 *)out; p.ld_o2 = ld_o; p.off_o2 = off_o;
    p.H = H; p.W = W;
    p.ntiles = (H / PT_H) * (W / PT_W);
    const int tpw_env = fd_dev(FD_DEV_PWDW_TPW);
    p.tpw = tpw_env > 0 ? tpw_env : (B >= 4 ? 4 : 1);
    dim3 grid((p.ntiles + p.tpw - 1) / p.tpw, B), block(256);
    const size_t pad = fd_occ_pad(FD_DEV_PAD_PWDW);
    hipLaunchKernelGGL((pwdw_kernel<64, true>), grid, block, pad, (hipStream_t)stream, p);
    FD_LAUNCH_OK("fd_pw_dw3x3_proj");
    return FD_OK;
}

// tiles per workgroup of the Gram-fused qkv kernel: a constant of the kernel set (the order of the partial sums, hence
// the result, must not depend on the batch): 8 in the throughput set (round 4: -0.3 % per batch-8 forward against 4;
// 16: +0.6 %), 4 under FD_OPT_LOW_LATENCY (one slice: 8 leaves 256 workgroups for 256 CUs, +2.5 ms per 50-step slice);
// FD_GRAM_TPW overrides both for experiments
static int gram_tpw(int dtype_opts) {
    const int gv = fd_dev(FD_DEV_GRAM_TPW), forced = gv >= 1 && gv <= 64 ? gv : 0;
    return forced ? forced : ((dtype_opts & FD_OPT_LOW_LATENCY) ? 4 : 8);
}

extern "C" int fd_pw_dw3x3_gram_ok(int dtype, int Cin, int H, int W) {
    const bool off = fd_dev(FD_DEV_NO_GRAM_FUSE);      // development switch
    return !off && Cin == 64 && fd_pw_dw3x3_ok(dtype, Cin, 192, 0, H, W);      // pwdw_gram_kernel is the 64-channel form
}

extern "C" int fd_pw_dw3x3_gram_nblk_opts(int dtype_opts, int H, int W) {
    const int ntiles = (H / PT_H) * (W / PT_W), tpw = gram_tpw(dtype_opts);
    return (ntiles + tpw - 1) / tpw;
}

extern "C" int fd_pw_dw3x3_gram_nblk(int H, int W) { return fd_pw_dw3x3_gram_nblk_opts(0, H, W); }

extern "C" int fd_pw_dw3x3_gram(int dtype, const void *x, int ld_x, int off_x, int Cin, const float *ln_gamma,
                                const float *ln_beta, float ln_eps, const float *ln_shift, const float *ln_scale,
                                int ln_ld, const void *w_pw, const uint32_t *w_dw, void *out_v, int ld_v, int off_v,
                                float *partial, int B, int H, int W, void *stream) {
    const int dtype_opts = dtype;
    dtype &= 0xff;
    FD_REQUIRE(fd_pw_dw3x3_ok(dtype, Cin, 192, 0, H, W),
               "fd_pw_dw3x3_gram: unsupported shape (bf16, Cin=64, H%%8, W%%16, >= 32768 px): Cin=%d H=%d W=%d", Cin, H, W);
    FD_REQUIRE(x && ln_shift && ln_scale && w_pw && w_dw && partial, "fd_pw_dw3x3_gram: null pointer");      // out_v may be NULL: q, k only
    FD_REQUIRE(ld_x % 8 == 0 && off_x % 8 == 0 && ld_v % 8 == 0 && off_v % 8 == 0,
               "fd_pw_dw3x3_gram: strides / offsets must be multiples of 8 channels");
    FD_REQUIRE(((uintptr_t)w_dw & 15) == 0, "fd_pw_dw3x3_gram: weights must be 16-byte aligned");
    PwGramParams p;
    p.x = (const bf16 *)x; p.ld_x = ld_x; p.off_x = off_x;
    p.ln_gamma = ln_gamma; p.ln_beta = ln_beta; p.ln_shift = ln_shift; p.ln_scale = ln_scale; p.ln_ld = ln_ld; p.ln_eps = ln_eps;
    p.w_pw = (const bf16 *)w_pw; p.w_dw = w_dw;
    p.out_v = (bf16 *)out_v; p.ld_v = ld_v; p.off_v = off_v;
    p.part = partial; p.nblk = fd_pw_dw3x3_gram_nblk_opts(dtype_opts, H, W);
    p.H = H; p.W = W; p.tpw = gram_tpw(dtype_opts); p.ntiles = (H / PT_H) * (W / PT_W);
    dim3 grid(p.nblk, B), block(256);
    const size_t pad = fd_occ_pad(FD_DEV_PAD_PWDW);
    hipLaunchKernelGGL(pwdw_gram_kernel, grid, block, pad, (hipStream_t)stream, p);
    FD_LAUNCH_OK("fd_pw_dw3x3_gram");
    return FD_OK;
}

// ---- qkv_dwconv + Gram for C >= 128 (dwconv_gram_kernel): q, k of a [B,H,W,ld] qkv tensor -> Gram partials only
static int dwgram_tpw(int ntiles) { return ntiles >= 256 ? 4 : (ntiles >= 64 ? 2 : 1); }

extern "C" int fd_dwconv_gram_ok(int dtype, int C, int H, int W) {
    const bool off = fd_dev(FD_DEV_NO_DWGRAM);         // development switch
    return !off && dtype == FD_BF16 && C % 64 == 0 && C >= 64 && H % PT_H == 0 && W % PT_W == 0 &&
           (int64_t)H * W * 3 * C < (1ll << 31);
}

extern "C" int fd_dwconv_gram_nblk(int H, int W) {
    const int ntiles = (H / PT_H) * (W / PT_W), tpw = dwgram_tpw(ntiles);
    return (ntiles + tpw - 1) / tpw;
}

extern "C" int fd_dwconv_gram(int dtype, const void *qkv, int ld, int C, const uint32_t *w_dw, float *partial, int B, int H,
                              int W, void *stream) {
    FD_REQUIRE(fd_dwconv_gram_ok(dtype, C, H, W), "fd_dwconv_gram: unsupported shape (bf16, C %% 64, H %% 8, W %% 16): C=%d H=%d W=%d", C, H, W);
    FD_REQUIRE(qkv && w_dw && partial && ld % 8 == 0 && ld >= 2 * C && ((uintptr_t)w_dw & 15) == 0, "fd_dwconv_gram: bad args");
    DwGramParams p;
    p.qkv = (const bf16 *)qkv; p.ld = ld; p.C = C; p.w_dw = w_dw; p.part = partial;
    p.H = H; p.W = W; p.ntiles = (H / PT_H) * (W / PT_W); p.tpw = dwgram_tpw(p.ntiles); p.nblk = fd_dwconv_gram_nblk(H, W);
    dim3 grid(p.nblk, C / 64, B), block(256);
    hipLaunchKernelGGL(dwconv_gram_kernel, grid, block, 0, (hipStream_t)stream, p);
    FD_LAUNCH_OK("fd_dwconv_gram");
    return FD_OK;
}
