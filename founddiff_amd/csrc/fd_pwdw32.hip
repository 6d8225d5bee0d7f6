// fd_pwdw32.hip -- fused LayerNorm+modulate -> 1x1 conv -> depthwise 3x3 on FP32 STORAGE with split-bf16 contractions:
// the 64-channel Mamba blocks of the `fp32s` engine (precision='fp32s', the mode held to the north star's 1e-3 over the
// whole sampling loop).  Three forms of one kernel, the dataflow of the bf16 kernels of fd_pwdw.hip:
//   DW    x -> LNmod -> in_proj (x half) -> conv2d 3x3 + bias + SiLU -> xc          src/emamba2.py:716-722
//   GRAM  x1 -> LNmod -> q, k rows of qkv -> qkv_dwconv -> per-head q k^T + L2 norms  src/DADiff.py:266-276
//   PROJ  x1 -> LNmod -> v rows of qkv -> qkv_dwconv -> Weff[b] -> x1 + gate . ()    src/DADiff.py:266-285, 483-488
// Until round 6 the fp32s engine ran these as row kernels + generic GEMMs + fd_dwconv3x3 + fd_chan_attn_gram and moved
// the 2 d_inner / 3 C wide fp32 intermediates through HBM twice each.
//
// A workgroup owns an 8 x TW tile.  (0) the (8+2) x (TW+2) input halo is loaded once (8 lanes per pixel), LayerNorm'd in
// fp32 (two passes, DPP group sums) and parked in LDS as TWO bf16 images x = hi + lo.  Per 32-channel chunk of the 1x1:
// (1) t = W . x over all halo pixels, three bf16 MFMAs per product (hi.hi + hi.lo + lo.hi, ~2^-16: the arithmetic of
// every other fp32s contraction), transposed issue (rows = output channels: a lane ends up with 4 consecutive channels of
// one pixel), weights straight from L2 (pre-split by the caller into bf16 hi / lo), t as fp32 into a second LDS tile with out-of-image pixels
// forced to the depthwise conv's zero padding; (2) the depthwise 3x3 in fp32 from that tile (4 channels x RPT rows of one
// column per thread).  DW stores, GRAM parks q / k as fp32 [pixel][channel] tiles and contracts them over the pixels
// with the exact-f32 MFMA (operand = a plain 4-byte read of a pixel-major tile, fd_attn.hip), PROJ parks v as bf16
// hi / lo and applies Weff[b] with three MFMAs per product again.  No float atomics; partial sums in fixed order.
#include "fd_common.h"

namespace {

struct Pw32Params {
    const float *x; int ld_x, off_x;
    const float *ln_gamma, *ln_beta, *ln_shift, *ln_scale; int ln_ld; float ln_eps;
    const bf16 *w_hi, *w_lo;        // the 1x1 weights [rows][64] pre-split by the caller: w = hi + lo, hi = bf16(w), lo = bf16(w - hi)
    const float *w_dw; int ld_wdw;  // fp32 [9][ld_wdw], column = w_pw row
    const float *b_dw;              // [rows] or NULL
    int dw_silu, nchunk, ntap_ch;   // ntap_ch: depthwise channels (<= 128: staged in LDS)
    float *out; int ld_o, off_o;    // DW: out_dw;  PROJ: out
    const float *w2; const float *gate; int gate_ld;     // PROJ: [B][64][64], entries [b*gate_ld + n]
    float *part; int nblk;          // GRAM: [B][2][nblk][1024 + 64]
    int H, W, ntiles, tpw, tiles_x;
};

__device__ __forceinline__ void split8_32(const float (&f)[8], bf16x8 &hi, bf16x8 &lo) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const bf16 h = (bf16)f[e];
        hi[e] = h;
        lo[e] = (bf16)(f[e] - (float)h);
    }
}

#define FD_MFMA3(acc, wh, wl, xh, xl)                                                  \
    do {                                                                               \
        acc = FD_MFMA16(wl, xh, acc, 0, 0, 0);           \
        acc = FD_MFMA16(wh, xl, acc, 0, 0, 0);           \
        acc = FD_MFMA16(wh, xh, acc, 0, 0, 0);           \
    } while (0)

constexpr int PW32_DW = 0, PW32_GRAM = 1, PW32_PROJ = 2;

template <int TW>
struct Pw32Geo {
    static constexpr int TH = 8, HWD = TW + 2, HP = (TH + 2) * HWD, MT = (HP + 15) / 16, NPX = TH * TW;
    static constexpr int XB = MT * 16 * 128;                 // bytes of one bf16 half of the LayerNorm'd halo
    static constexpr int TB = HP * 128;                      // the fp32 1x1 tile [HP][32]
};

template <int MODE, int TW>
constexpr size_t pw32_lds() {
    using G = Pw32Geo<TW>;
    return 2 * G::XB + G::TB + (MODE == PW32_DW ? 0 : 2 * G::NPX * 128) + 2 * 64 * sizeof(float) + (MODE == PW32_PROJ ? 2 * 64 * 128 : 0) +
           10 * (MODE == PW32_PROJ ? 64 : 128) * sizeof(float);
}

template <int MODE, int TW>
__global__ __launch_bounds__(256, 2) void pwdw32_kernel(const Pw32Params p) {
    using G = Pw32Geo<TW>;
    constexpr int TH = G::TH, HWD = G::HWD, HP = G::HP, MT = G::MT, NPX = G::NPX;
    constexpr int RPT = TH / (256 / (8 * TW));               // output rows per thread in the depthwise phase
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char *sXh = smem, *sXl = smem + G::XB;
    unsigned char *sT = smem + 2 * G::XB;
    unsigned char *sA = sT + G::TB;                          // GRAM: q | k fp32 [NPX][32];  PROJ: v_hi | v_lo bf16 [NPX][64]
    float *sG = (float *)(sA + (MODE == PW32_DW ? 0 : 2 * NPX * 128));
    float *sB = sG + 64;
    unsigned char *sW2 = (unsigned char *)(sB + 64);         // PROJ: Weff[b] as bf16 hi | lo [64][64], rows of 128 bytes
    constexpr int NTC = MODE == PW32_PROJ ? 64 : 128;        // depthwise channels whose taps + bias are staged: [10][NTC] (DW: per 128-channel group)
    float *sTap = (float *)(sW2 + (MODE == PW32_PROJ ? 2 * 64 * 128 : 0));
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fg = lane >> 4;
    const int b = blockIdx.y;
    const float *xb = p.x + (int64_t)b * p.H * p.W * p.ld_x + p.off_x;
    if (MODE == PW32_PROJ) {
        // Weff[b] (fd_chan_attn_weff's fp32 output) split once per workgroup: 64 rows x 8 chunks of 8 channels
        const float *w2 = p.w2 + (int64_t)b * 64 * 64;
        for (int idx = tid; idx < 64 * 8; idx += 256) {
            const int row = idx >> 3, ch = idx & 7;
            float f[8];
            load8(w2 + row * 64 + ch * 8, f);
            bf16x8 hi, lo;
            split8_32(f, hi, lo);
            const int o = row * 128 + ((ch ^ (row & 7)) << 4);
            *(bf16x8 *)(sW2 + o) = hi;
            *(bf16x8 *)(sW2 + 64 * 128 + o) = lo;
        }
    }
    // depthwise taps + bias of this launch's channels in LDS (their reads then count on lgkmcnt, not behind the halo prefetch on
    // vmcnt): row k < 9 = tap k, row 9 = bias; the GRAM form's q / k rows 0..127 of the qkv taps, PROJ's 64 v rows, DW's rows
    for (int idx = tid; idx < 10 * NTC; idx += 256) {
        const int k = idx / NTC, c = idx - k * NTC;
        sTap[idx] = k < 9 ? (c < p.ntap_ch ? p.w_dw[(int64_t)k * p.ld_wdw + c] : 0.f) : (p.b_dw && c < p.ntap_ch ? p.b_dw[c] : 0.f);
    }
    if (tid < 64) {
        const float g = p.ln_gamma ? p.ln_gamma[tid] : 1.f, be = p.ln_beta ? p.ln_beta[tid] : 0.f;
        const float sc = 1.f + p.ln_scale[(int64_t)b * p.ln_ld + tid], sh = p.ln_shift[(int64_t)b * p.ln_ld + tid];
        sG[tid] = g * sc;
        sB[tid] = be * sc + sh;
    }
    // depthwise geometry of this thread: 4 channels (chunk-relative 4 c4 ..), column xx, rows rg * RPT ..
    const int c4 = tid & 7, xx = (tid >> 3) % TW, rg = tid / (8 * TW);
    int tcol[3];
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) tcol[dx] = (xx + dx) * 128 + ((c4 ^ ((xx + dx) & 7)) << 4);
    f32x4 gacc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};        // GRAM: this wave's 16 x 16 block of both heads
    f32x4 nrm[MODE == PW32_GRAM ? 4 : 1];                               // GRAM: sum of squares of this thread's channels
#pragma unroll
    for (int j = 0; j < (MODE == PW32_GRAM ? 4 : 1); ++j) nrm[j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // The halo loads of a tile: NIT x 32 bytes per thread (8 lanes per pixel), ALL issued before anything waits for them, from
    // clamped addresses (an out-of-image pixel loads its nearest image pixel and is zeroed at use): one HBM round trip per tile
    // instead of one per iteration -- and the NEXT tile's loads are issued as soon as this tile's are parked in LDS, so that
    // round trip runs under the tile's MFMA / depthwise phases.
    constexpr int NIT = (MT * 16 * 8 + 255) / 256;
    f32x4 raw[NIT][2];
    auto halo_load = [&](int t) {
        const int ty = t / p.tiles_x, tx = t - ty * p.tiles_x;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int task = it * 256 + tid;
            const int hp = min(task >> 3, HP - 1), sub = task & 7;
            const int hy = hp / HWD, hx = hp - hy * HWD;
            const int gy = min(max(ty * TH - 1 + hy, 0), p.H - 1), gx = min(max(tx * TW - 1 + hx, 0), p.W - 1);
            const float *src = xb + ((int64_t)gy * p.W + gx) * p.ld_x + sub * 8;
            raw[it][0] = *(const f32x4 *)src;
            raw[it][1] = *(const f32x4 *)(src + 4);
        }
    };
    const int t_begin = blockIdx.x * p.tpw, t_end = min(p.ntiles, ((int)blockIdx.x + 1) * p.tpw);
    if (t_begin < t_end) halo_load(t_begin);
    for (int t = t_begin; t < t_end; ++t) {
        const int ty = t / p.tiles_x, tx = t - ty * p.tiles_x;
        const int y0 = ty * TH, x0 = tx * TW;
        // ---- (0) halo -> LayerNorm + modulate -> bf16 hi / lo tiles
        __syncthreads();           // (first tile: sG / sB / sW2; later tiles: the previous tile's readers of sA are done)
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int task = it * 256 + tid;
            const int hp = task >> 3, sub = task & 7;
            float f[8] = {raw[it][0][0], raw[it][0][1], raw[it][0][2], raw[it][0][3], raw[it][1][0], raw[it][1][1], raw[it][1][2], raw[it][1][3]};
            float s = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) s += f[e];
            const float mean = fd_group_sum<8>(s) * (1.f / 64);
            float q = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float d = f[e] - mean; q += d * d; }
            const float rstd = rsqrtf(fd_group_sum<8>(q) * (1.f / 64) + p.ln_eps);
            float g8[8], b8[8];
            load8(sG + sub * 8, g8);
            load8(sB + sub * 8, b8);
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] = (f[e] - mean) * rstd * g8[e] + b8[e];      // (out-of-image pixels: a clamped neighbour's
                                                                                          //  values -- finite; t is zeroed there in phase 1)
            bf16x8 hi, lo;
            split8_32(f, hi, lo);
            const int o = hp * 128 + ((sub ^ (hp & 7)) << 4);
            if (hp < MT * 16) {
                *(bf16x8 *)(sXh + o) = hi;
                *(bf16x8 *)(sXl + o) = lo;
            }
        }
        // (GRAM / PROJ: a compile-time chunk count, fully unrolled -- nrm[ch] and gacc[ch >> 1] stay registers)
        constexpr int NCS = MODE == PW32_GRAM ? 4 : (MODE == PW32_PROJ ? 2 : 0);
        constexpr int NCU = NCS ? NCS : 1;
        const int nch = NCS ? NCS : p.nchunk;
#pragma unroll NCU
        for (int ch = 0; ch < nch; ++ch) {
            // weight rows of this chunk: DW / PROJ 32 ch ..; GRAM: q of head 0, k of head 0, q of head 1, k of head 1
            int row0 = MODE == PW32_GRAM ? (ch & 1) * 64 + (ch >> 1) * 32 : ch * 32;
            // (opaque: in the unrolled forms the compiler otherwise hoists the weight and tap loads of ALL chunks to the top of
            //  the tile -- 4 x 68 registers, 216 bytes of scratch in the Gram form)
            asm volatile("" : "+s"(row0));
            // ---- (1) t = W . x over the halo, zero outside the image
            bf16x8 wh[2][2], wl[2][2];
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const int64_t wo = (int64_t)(row0 + nt * 16 + fr) * 64 + ks * 32 + 8 * fg;
                    wh[nt][ks] = *(const bf16x8 *)(p.w_hi + wo);
                    wl[nt][ks] = *(const bf16x8 *)(p.w_lo + wo);
                }
            // The NEXT tile's halo loads go out behind the last chunk's weight loads: vmcnt retires in order, so a load issued
            // before them would have to land before the first MFMA of every later chunk (the prefetch would hide nothing); behind
            // them it runs under this chunk's MFMA, depthwise and Gram / project_out phases, which issue no vector loads.
            if (ch == nch - 1 && t + 1 < t_end) halo_load(t + 1);
            f32x4 wt[9], o[RPT];
            __syncthreads();       // the halo tiles are complete / the previous chunk's depthwise reads of sT are done
            for (int mt = wave; mt < MT; mt += 4) {
                const int hp = mt * 16 + fr;
                const int o0 = hp * 128 + ((fg ^ (hp & 7)) << 4), o1 = hp * 128 + (((4 + fg) ^ (hp & 7)) << 4);
                const bf16x8 xh0 = *(const bf16x8 *)(sXh + o0), xl0 = *(const bf16x8 *)(sXl + o0);
                const bf16x8 xh1 = *(const bf16x8 *)(sXh + o1), xl1 = *(const bf16x8 *)(sXl + o1);
                const int hy = hp / HWD, hx = hp - hy * HWD;
                const int gy = y0 - 1 + hy, gx = x0 - 1 + hx;
                const bool in = gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                    FD_MFMA3(acc, wh[nt][0], wl[nt][0], xh0, xl0);
                    FD_MFMA3(acc, wh[nt][1], wl[nt][1], xh1, xl1);
                    if (!in) acc = (f32x4){0.f, 0.f, 0.f, 0.f};
                    if (hp < HP) *(f32x4 *)(sT + hp * 128 + (((nt * 4 + fg) ^ (hx & 7)) << 4)) = acc;
                }
            }
            // ---- (2) depthwise 3x3 from the fp32 tile
            {
                const int tc = (MODE == PW32_DW ? (row0 & 127) : row0) + 4 * c4;     // (DW with more than 128 channels: not staged, see launcher)
#pragma unroll
                for (int k = 0; k < 9; ++k) wt[k] = *(const f32x4 *)(sTap + k * NTC + tc);
                const f32x4 bias = *(const f32x4 *)(sTap + 9 * NTC + tc);
#pragma unroll
                for (int oy = 0; oy < RPT; ++oy) o[oy] = bias;
            }
            __syncthreads();
#pragma unroll
            for (int r = 0; r < RPT + 2; ++r) {
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    // (swizzled by the halo COLUMN: a thread's 3 column bases are tile- and chunk-invariant, rows are immediates)
                    const f32x4 v = *(const f32x4 *)(sT + tcol[dx] + (rg * RPT + r) * (HWD * 128));
#pragma unroll
                    for (int oy = 0; oy < RPT; ++oy) {
                        const int dy = r - oy;
                        if (dy >= 0 && dy < 3) o[oy] += wt[dy * 3 + dx] * v;
                    }
                }
            }
#pragma unroll
            for (int oy = 0; oy < RPT; ++oy) {
                const int yl = rg * RPT + oy, pl = yl * TW + xx;
                if (MODE == PW32_DW) {
                    f32x4 v = o[oy];
                    if (p.dw_silu) {
                        const f32x2 a = fd_silu2(f32x2{v[0], v[1]}), c = fd_silu2(f32x2{v[2], v[3]});
                        v = (f32x4){a.x, a.y, c.x, c.y};
                    }
                    float *dst = p.out + (((int64_t)b * p.H + y0 + yl) * p.W + x0 + xx) * p.ld_o + p.off_o + row0 + 4 * c4;
                    *(f32x4 *)dst = v;
                } else if (MODE == PW32_GRAM) {
                    unsigned char *tile = sA + (ch & 1) * NPX * 128;
                    *(f32x4 *)(tile + pl * 128 + ((c4 ^ (pl & 7)) << 4)) = o[oy];
                    nrm[ch] += o[oy] * o[oy];
                } else {
                    const f32x4 v = o[oy];
                    typedef __attribute__((ext_vector_type(4))) bf16 bf16x4;
                    bf16x4 hi, lo;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const bf16 h = (bf16)v[e];
                        hi[e] = h;
                        lo[e] = (bf16)(v[e] - (float)h);
                    }
                    const int ofs = pl * 128 + (((ch * 4 + (c4 >> 1)) ^ (pl & 7)) << 4) + (c4 & 1) * 8;
                    *(bf16x4 *)(sA + ofs) = hi;
                    *(bf16x4 *)(sA + NPX * 128 + ofs) = lo;
                }
            }
            // ---- (3, GRAM) q k^T of the head whose k chunk has just been written: exact-f32 MFMA over the tile's pixels
            if (MODE == PW32_GRAM && (ch & 1)) {
                __syncthreads();
                const int it = wave >> 1, jt = wave & 1;
                const unsigned char *tq = sA, *tk = sA + NPX * 128;
                f32x4 acc = gacc[ch >> 1];
#pragma unroll 4
                for (int s = 0; s < NPX / 4; ++s) {
                    const int pl = 4 * s + fg;
                    const float a = *(const float *)(tq + pl * 128 + (((it * 4 + (fr >> 2)) ^ (pl & 7)) << 4) + (fr & 3) * 4);
                    const float k = *(const float *)(tk + pl * 128 + (((jt * 4 + (fr >> 2)) ^ (pl & 7)) << 4) + (fr & 3) * 4);
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, k, acc, 0, 0, 0);
                }
                gacc[ch >> 1] = acc;
            }
        }
        // ---- (3, PROJ) out = x + gate . (Weff[b] . v): rows = output channels, columns = this wave's 16 pixels
        if (MODE == PW32_PROJ) {
            __syncthreads();
            for (int mt = wave; mt < NPX / 16; mt += 4) {
                const int pl = mt * 16 + fr;
                const int o0 = pl * 128 + ((fg ^ (pl & 7)) << 4), o1 = pl * 128 + (((4 + fg) ^ (pl & 7)) << 4);
                const bf16x8 vh0 = *(const bf16x8 *)(sA + o0), vl0 = *(const bf16x8 *)(sA + NPX * 128 + o0);
                const bf16x8 vh1 = *(const bf16x8 *)(sA + o1), vl1 = *(const bf16x8 *)(sA + NPX * 128 + o1);
                const int yl = pl / TW, xl_ = pl - yl * TW;
                const int64_t pix = ((int64_t)b * p.H + y0 + yl) * p.W + x0 + xl_;
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) {
                        const int row = nt * 16 + fr;
                        const int wo = row * 128 + (((ks * 4 + fg) ^ (row & 7)) << 4);
                        const bf16x8 ah = *(const bf16x8 *)(sW2 + wo), al = *(const bf16x8 *)(sW2 + 64 * 128 + wo);
                        if (ks == 0) FD_MFMA3(acc, ah, al, vh0, vl0);
                        else FD_MFMA3(acc, ah, al, vh1, vl1);
                    }
                    const int n0 = nt * 16 + 4 * fg;
                    const f32x4 res = *(const f32x4 *)(p.x + pix * p.ld_x + p.off_x + n0);
                    const f32x4 gt = *(const f32x4 *)(p.gate + (int64_t)b * p.gate_ld + n0);
                    *(f32x4 *)(p.out + pix * p.ld_o + p.off_o + n0) = res + gt * acc;
                }
            }
        }
    }
    if (MODE == PW32_GRAM) {
        // one partial per workgroup: [head][32 x 32 Gram, rows = q channels | sum q^2 (32) | sum k^2 (32)]
        float *outp = p.part + (((int64_t)b * 2) * p.nblk + blockIdx.x) * (1024 + 64);
        const int64_t hstride = (int64_t)p.nblk * (1024 + 64);
        const int it = wave >> 1, jt = wave & 1;
#pragma unroll
        for (int hd = 0; hd < 2; ++hd)
#pragma unroll
            for (int e = 0; e < 4; ++e) outp[hd * hstride + (it * 16 + 4 * fg + e) * 32 + jt * 16 + fr] = gacc[hd][e];
        __syncthreads();
        float *scr = (float *)smem;                           // [256 threads][16]: 16 KB of the (dead) halo tiles
#pragma unroll
        for (int j = 0; j < 4; ++j) *(f32x4 *)(scr + tid * 16 + 4 * j) = nrm[MODE == PW32_GRAM ? j : 0];
        __syncthreads();
        if (tid < 128) {
            const int j = tid >> 5, chn = tid & 31;             // chunk j: q / k of head j >> 1
            const int cc = chn >> 2, e = chn & 3;
            float s = 0.f;
            for (int k = 0; k < 32; ++k) s += scr[(cc + 8 * k) * 16 + 4 * j + e];
            outp[(j >> 1) * hstride + 1024 + (j & 1) * 32 + chn] = s;
        }
    }
}

constexpr int PW32_GRAM_TPW = 16;       // 8 x 8 tiles per workgroup of the Gram form: fixes the order of the partial sums

bool pw32_shape_ok(int dtype, int Cin, int H, int W, int TW) {
    return dtype == FD_F32 && Cin == 64 && H % 8 == 0 && W % TW == 0 && (int64_t)H * W >= 16384 && (int64_t)H * W * 256 < (1ll << 31);
}

void pw32_common(Pw32Params &p, const void *x, int ld_x, int off_x, const float *ln_gamma, const float *ln_beta, float ln_eps,
                 const float *ln_shift, const float *ln_scale, int ln_ld, int H, int W, int TW) {
    p.x = (const float *)x; p.ld_x = ld_x; p.off_x = off_x;
    p.ln_gamma = ln_gamma; p.ln_beta = ln_beta; p.ln_shift = ln_shift; p.ln_scale = ln_scale; p.ln_ld = ln_ld; p.ln_eps = ln_eps;
    p.H = H; p.W = W; p.tiles_x = W / TW; p.ntiles = (H / 8) * (W / TW);
}

template <int MODE, int TW>
void pw32_launch(const Pw32Params &p, int B, hipStream_t s) {
    constexpr size_t lds = pw32_lds<MODE, TW>();
    static_assert(lds <= 80 * 1024, "pwdw32_kernel: two workgroups per CU");
    (void)hipFuncSetAttribute((const void *)pwdw32_kernel<MODE, TW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    dim3 grid((p.ntiles + p.tpw - 1) / p.tpw, B), block(256);
    hipLaunchKernelGGL((pwdw32_kernel<MODE, TW>), grid, block, lds, s, p);
}

}  // namespace

extern "C" int fd_pw_dw3x3_f32_ok(int dtype, int Cin, int Cdw, int H, int W) {
    return pw32_shape_ok(dtype & 0xff, Cin, H, W, 16) && Cdw > 0 && Cdw % 32 == 0 && Cdw <= 128;      // (taps of <= 128 channels staged in LDS)
}

extern "C" int fd_pw_dw3x3_f32(const void *x, int ld_x, int off_x, int Cin, const float *ln_gamma, const float *ln_beta,
                               float ln_eps, const float *ln_shift, const float *ln_scale, int ln_ld, const void *w_pw_hi, const void *w_pw_lo, int Cdw,
                               const float *w_dw, const float *b_dw, int dw_silu, void *out_dw, int ld_dw, int off_dw, int B, int H,
                               int W, void *stream) {
    FD_REQUIRE(fd_pw_dw3x3_f32_ok(FD_F32, Cin, Cdw, H, W), "fd_pw_dw3x3_f32: unsupported shape (Cin=64, Cdw%%32, H%%8, W%%16, >= 16384 px): "
               "Cin=%d Cdw=%d H=%d W=%d", Cin, Cdw, H, W);
    FD_REQUIRE(x && ln_shift && ln_scale && w_pw_hi && w_pw_lo && w_dw && out_dw, "fd_pw_dw3x3_f32: null pointer");
    FD_REQUIRE(ld_x % 4 == 0 && off_x % 4 == 0 && ld_dw % 4 == 0 && off_dw % 4 == 0, "fd_pw_dw3x3_f32: strides / offsets must be multiples of 4 channels");
    FD_REQUIRE((((uintptr_t)x | (uintptr_t)w_pw_hi | (uintptr_t)w_pw_lo | (uintptr_t)w_dw | (uintptr_t)out_dw | (uintptr_t)b_dw) & 15) == 0, "fd_pw_dw3x3_f32: 16-byte alignment");
    Pw32Params p = {};
    pw32_common(p, x, ld_x, off_x, ln_gamma, ln_beta, ln_eps, ln_shift, ln_scale, ln_ld, H, W, 16);
    p.w_hi = (const bf16 *)w_pw_hi; p.w_lo = (const bf16 *)w_pw_lo; p.w_dw = w_dw; p.ld_wdw = Cdw; p.b_dw = b_dw; p.dw_silu = dw_silu; p.nchunk = Cdw / 32; p.ntap_ch = Cdw;
    p.out = (float *)out_dw; p.ld_o = ld_dw; p.off_o = off_dw;
    p.tpw = B >= 4 ? 4 : 1;             // tiles are independent: the split does not touch the results
    pw32_launch<PW32_DW, 16>(p, B, (hipStream_t)stream);
    FD_LAUNCH_OK("fd_pw_dw3x3_f32");
    return FD_OK;
}

extern "C" int fd_pw_dw3x3_gram_f32_ok(int dtype, int Cin, int H, int W) { return pw32_shape_ok(dtype & 0xff, Cin, H, W, 8); }

extern "C" int fd_pw_dw3x3_gram_f32_nblk(int H, int W) { return ((H / 8) * (W / 8) + PW32_GRAM_TPW - 1) / PW32_GRAM_TPW; }

extern "C" int fd_pw_dw3x3_gram_f32(const void *x, int ld_x, int off_x, int Cin, const float *ln_gamma, const float *ln_beta,
                                    float ln_eps, const float *ln_shift, const float *ln_scale, int ln_ld, const void *w_pw_hi, const void *w_pw_lo,
                                    const float *w_dw, int ld_wdw, float *partial, int B, int H, int W, void *stream) {
    FD_REQUIRE(fd_pw_dw3x3_gram_f32_ok(FD_F32, Cin, H, W), "fd_pw_dw3x3_gram_f32: unsupported shape (Cin=64, H%%8, W%%8, >= 16384 px): "
               "Cin=%d H=%d W=%d", Cin, H, W);
    FD_REQUIRE(x && ln_shift && ln_scale && w_pw_hi && w_pw_lo && w_dw && partial, "fd_pw_dw3x3_gram_f32: null pointer");
    FD_REQUIRE(ld_x % 4 == 0 && off_x % 4 == 0 && ld_wdw % 4 == 0 && ld_wdw >= 128, "fd_pw_dw3x3_gram_f32: strides / offsets must be multiples of 4 channels");
    FD_REQUIRE((((uintptr_t)x | (uintptr_t)w_pw_hi | (uintptr_t)w_pw_lo | (uintptr_t)w_dw) & 15) == 0, "fd_pw_dw3x3_gram_f32: 16-byte alignment");
    Pw32Params p = {};
    pw32_common(p, x, ld_x, off_x, ln_gamma, ln_beta, ln_eps, ln_shift, ln_scale, ln_ld, H, W, 8);
    p.w_hi = (const bf16 *)w_pw_hi; p.w_lo = (const bf16 *)w_pw_lo; p.w_dw = w_dw; p.ld_wdw = ld_wdw; p.nchunk = 4; p.ntap_ch = 128;
    p.part = partial; p.nblk = fd_pw_dw3x3_gram_f32_nblk(H, W); p.tpw = PW32_GRAM_TPW;
    pw32_launch<PW32_GRAM, 8>(p, B, (hipStream_t)stream);
    FD_LAUNCH_OK("fd_pw_dw3x3_gram_f32");
    return FD_OK;
}

extern "C" int fd_pw_dw3x3_proj_f32_ok(int dtype, int Cin, int H, int W) { return pw32_shape_ok(dtype & 0xff, Cin, H, W, 8); }

extern "C" int fd_pw_dw3x3_proj_f32(const void *x, int ld_x, int off_x, int Cin, const float *ln_gamma, const float *ln_beta,
                                    float ln_eps, const float *ln_shift, const float *ln_scale, int ln_ld, const void *w_pw_hi, const void *w_pw_lo,
                                    const float *w_dw, int ld_wdw, const float *w2, const float *gate, int gate_ld, void *out,
                                    int ld_o, int off_o, int B, int H, int W, void *stream) {
    FD_REQUIRE(fd_pw_dw3x3_proj_f32_ok(FD_F32, Cin, H, W), "fd_pw_dw3x3_proj_f32: unsupported shape (Cin=64, H%%8, W%%8, >= 16384 px): "
               "Cin=%d H=%d W=%d", Cin, H, W);
    FD_REQUIRE(x && ln_shift && ln_scale && w_pw_hi && w_pw_lo && w_dw && w2 && gate && out, "fd_pw_dw3x3_proj_f32: null pointer");
    FD_REQUIRE(ld_x % 4 == 0 && off_x % 4 == 0 && ld_o % 4 == 0 && off_o % 4 == 0 && ld_wdw % 4 == 0 && gate_ld % 4 == 0,
               "fd_pw_dw3x3_proj_f32: strides / offsets must be multiples of 4 channels");
    FD_REQUIRE((((uintptr_t)x | (uintptr_t)w_pw_hi | (uintptr_t)w_pw_lo | (uintptr_t)w_dw | (uintptr_t)w2 | (uintptr_t)gate | (uintptr_t)out) & 15) == 0,
               "fd_pw_dw3x3_proj_f32: 16-byte alignment");
    Pw32Params p = {};
    pw32_common(p, x, ld_x, off_x, ln_gamma, ln_beta, ln_eps, ln_shift, ln_scale, ln_ld, H, W, 8);
    p.w_hi = (const bf16 *)w_pw_hi; p.w_lo = (const bf16 *)w_pw_lo; p.w_dw = w_dw; p.ld_wdw = ld_wdw; p.nchunk = 2; p.ntap_ch = 64;
    p.w2 = w2; p.gate = gate; p.gate_ld = gate_ld;
    p.out = (float *)out; p.ld_o = ld_o; p.off_o = off_o;
    p.tpw = B >= 4 ? 8 : 2;
    pw32_launch<PW32_PROJ, 8>(p, B, (hipStream_t)stream);
    FD_LAUNCH_OK("fd_pw_dw3x3_proj_f32");
    return FD_OK;
}
