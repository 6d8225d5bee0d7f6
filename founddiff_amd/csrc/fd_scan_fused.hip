// fd_scan_fused.hip -- the SS2D selective scan of the high-resolution levels as ONE launch (round 4).
//
// The 3-phase form (fd_scan.hip) visits every (channel, position) twice with the full per-step arithmetic -- dt_proj,
// softplus and the N decay exponentials in the local scan AND in the re-run from the carry-in -- reads u two to three
// times and round-trips x_dbl and the chunk states through HBM.  Here a workgroup owns a chunk of CL = S * W positions x
// all d_inner channels and keeps what the second pass needs ON CHIP:
//   * the chunk's u tile ([CL][D] bf16) is loaded ONCE into LDS; x_proj (src/emamba2.py:332) runs on MFMA from it, the
//     recurrence reads its u values from it, and y is written back over it and leaves as 16-byte row-contiguous stores;
//     x_dbl never exists in HBM;
//   * pass 1: wave (channel group, sub-chunk) walks its S steps once: dt = softplus(dt_proj), the N decays exp2(A dt)
//     and dt*u stay in REGISTERS ((N + 1) * S per lane); it folds its sub-chunk to (decay product, end state);
//   * the W sub-chunks are chained through LDS, the chunk's aggregate (sum dt, H[N]) per channel is published, and the
//     chunk's carry-in is composed from the aggregates of earlier chunks (below);
//   * pass 2 re-runs the S steps from the carry-in with the CACHED decays: h = da h + (dt u) B, y = h.C + D u --
//     no transcendental, no dt_proj: ~11 instead of ~28 issue slots per (channel, position).
// Per (channel, position): ~37 VALU issue slots instead of ~54, u read once instead of 2-3 times.
//
// Carry between chunks inside the launch -- a FIXED tree, so the result does not depend on timing, batch or world size
// (bitwise batch invariance is tested): chunks take tickets in start order (chunk-major over the B * 4 sequences);
// within a supergroup of 64 chunks, chunk j publishes its own aggregate (level 0) and, where (j + 1) is a multiple of
// 2^k, the aggregate of the 2^k chunks ending at j (level k) = left half (level k-1, published by chunk j - 2^(k-1))
// composed with its own level k-1 block; its carry-in is the supergroup's carry-in composed, left to right, with the
// <= 6 blocks of the binary decomposition of j (a Fenwick prefix); the last chunk of a supergroup publishes the next
// supergroup's carry-in.  Every dependency points to a LOWER ticket, i.e. to a workgroup that has already started:
// the lowest unfinished ticket never waits, so the launch cannot deadlock; every spin is bounded anyway (ctrl[1] != 0
// reports a timeout).  Hand-off per MI355X_MICROARCH.md "inter-workgroup visibility": payload by sc1 (agent-scope,
// write-through) stores, every storing wave drains vmcnt, barrier, ONE lane stores the flag; the consumer polls the
// flags with sc1 loads from one wave, barrier, then reads the payload with sc1 loads only.
#include "fd_common.h"

namespace {

constexpr int SGC = 64;                  // chunks per supergroup (2^6: levels 0..6)
constexpr int NLV = 7;
constexpr unsigned SPIN_LIMIT = 1u << 22;

struct FusedParams {
    const bf16 *xc; const bf16 *xw;      // u [B,H,W,D];  x_proj weights [4][CD][D]
    const float *dtw, *dtb, *A, *Ds;
    bf16 *y;
    unsigned *ctrl;                      // [0] ticket, [1] timeout word, then the flags
    float *agg;                          // level aggregates, [NA][D] floats each
    float *top;                          // supergroup carry-ins, [N][D] floats each
    int B, H, W, D, H2, W2, L, nch, nbk, nsg;
    int flag_off[NLV + 1], agg_off[NLV];  // per level: offset of its [nbk][cnt_k] block, in entries
    int dbg;                             // development (FD_SCANF_DBG): 1 = no look-back at all (timing only, wrong results), 2 = no polls
};

typedef unsigned int u32;
__device__ __forceinline__ void st_sc1(float *p, float v) {
    __hip_atomic_store((u32 *)p, __builtin_bit_cast(u32, v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float ld_sc1(const float *p) {
    return __builtin_bit_cast(float, __hip_atomic_load((const u32 *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ void drain_vm() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// one wave polls up to 8 flags (lane i < n polls word idx[i]); returns when all are set (or the spin bound is hit)
__device__ __forceinline__ void poll_flags(const unsigned *flags, const int (&idx)[8], int n, unsigned *tmo, int lane) {
    int my = idx[0];
#pragma unroll
    for (int i = 1; i < 8; ++i) my = lane == i ? idx[i] : my;
    const bool mine = lane < n;
    for (unsigned spins = 0;; ++spins) {
        unsigned v = 1u;
        if (mine) v = __hip_atomic_load(flags + my, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (__all(v != 0u)) break;
        if (spins > SPIN_LIMIT) {
            if (lane == 0) __hip_atomic_store(tmo, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            break;
        }
        __builtin_amdgcn_s_sleep(4);
    }
}

template <int N, int R, int DCH, int S, int W, int OCC>
__global__ __launch_bounds__(W * DCH, OCC) void scan_fused_kernel(const FusedParams p) {
    constexpr int NCG = DCH / 64, NTHR = W * DCH, CL = S * W;
    constexpr int CD = R + 2 * N, CDP = (CD + 3) & ~3, NA = N + 1;
    constexpr int NSL = N / W;                       // states per thread in the look-back mapping (thread = channel x state group)
    constexpr int NP = N / 2, RP = R / 2;
    constexpr int TROW = DCH * 2, TCH = TROW / 16;   // bytes / 16-byte chunks per tile row
    static_assert(N % W == 0 && N % 2 == 0 && R % 2 == 0 && CL % 16 == 0 && CL / 16 <= W * NCG, "scan_fused_kernel: shape");
    static_assert(TCH == 16 || TCH == 32, "tile rows of 256 / 512 bytes");
    __shared__ __attribute__((aligned(16))) unsigned char tile[CL * TROW];     // u, later y: row = position, swizzled chunks
    __shared__ __attribute__((aligned(16))) float sx[CL * CDP];                // x_dbl rows [dt_r | B | C ln2]
    __shared__ int spix[CL];
    __shared__ float chP[W][N][DCH], chH[W][N][DCH], chS[W][DCH];
    __shared__ float s_carry[N][DCH];
    __shared__ int s_ticket;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cg = wave % NCG, sc = wave / NCG;      // channel group, sub-chunk of this wave
    auto swz = [](int row, int chunk) -> int { return (chunk ^ (row & (TCH - 1))) << 4; };

    if (tid == 0) s_ticket = (int)atomicAdd(p.ctrl, 1u);
    __syncthreads();
    const int ticket = s_ticket;
    const int c = ticket / p.nbk, bk = ticket - c * p.nbk;          // chunk-major: the chunks of all sequences advance together
    const int b = bk >> 2, k = bk & 3;
    const int sg = c / SGC, j = c - sg * SGC;
    const int l0 = c * CL;

    // ---- positions -> pixels (EfficientScan: 4 stride-2 sub-grids, directions 1 / 3 column-major; src/emamba2.py:182-262)
    if (tid < CL) {
        const int l = l0 + tid;
        int h2, w2;
        if (k & 1) { w2 = l / p.H2; h2 = l - w2 * p.H2; }
        else { h2 = l / p.W2; w2 = l - h2 * p.W2; }
        spix[tid] = (2 * h2 + (k & 1)) * p.W + 2 * w2 + (k >> 1);
    }
    __syncthreads();
    // ---- the chunk's u tile, once
    const bf16 *ub = p.xc + (int64_t)b * p.H * p.W * DCH;
    {
        constexpr int RPP = NTHR / TCH;              // rows per pass
#pragma unroll
        for (int r0 = 0; r0 < CL; r0 += RPP) {
            const int row = r0 + tid / TCH, ch = tid % TCH;
            if (RPP <= CL - r0 || row < CL) {
                const u32x4 v = *(const u32x4 *)(ub + (int64_t)spix[row] * DCH + ch * 8);
                *(u32x4 *)(tile + row * TROW + swz(row, ch)) = v;
            }
        }
    }
    // per-lane channel constants (dt in base-2 units: log2e folded into w / bias, fd_scan.hip LOG2U)
    const int d = cg * 64 + lane, kd = k * DCH + d;
    constexpr float WS = 1.4426950408889634f;
    f32x2 w[RP], a2[NP];
#pragma unroll
    for (int r = 0; r < RP; ++r) w[r] = f32x2{p.dtw[(int64_t)kd * R + 2 * r], p.dtw[(int64_t)kd * R + 2 * r + 1]} * WS;
#pragma unroll
    for (int n = 0; n < NP; ++n) a2[n] = f32x2{p.A[(int64_t)kd * N + 2 * n], p.A[(int64_t)kd * N + 2 * n + 1]};
    const float bias = p.dtb[kd] * WS, Dd = p.Ds[kd];
    __syncthreads();

    // ---- x_proj on the matrix cores: x_dbl[l][e] = sum_c Wx[k][e][c] u[l][c]  (rows = outputs e, columns = 16 positions)
    if (wave < CL / 16) {
        const bf16 *Wk = p.xw + (int64_t)k * CD * DCH;
        const int fr = lane & 15, fg = lane >> 4;
        constexpr int MB = (CD + 15) / 16, KSN = DCH / 32;
        const int row = wave * 16 + fr;
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
            const int e = mb * 16 + fr;
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KSN; ++ks) {
                bf16x8 af = {0, 0, 0, 0, 0, 0, 0, 0};
                if (e < CD) af = *(const bf16x8 *)(Wk + (int64_t)e * DCH + 32 * ks + 8 * fg);
                const bf16x8 bfv = *(const bf16x8 *)(tile + row * TROW + swz(row, 4 * ks + fg));
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bfv, acc, 0, 0, 0);
            }
            const int e0 = mb * 16 + 4 * fg;
            if (e0 < CD) {
#pragma unroll
                for (int i = 0; i < 4; ++i)        // the ln2 of dt*u (dt is carried in base-2 units), once per C element
                    if (e0 + i >= R + N) acc[i] *= 0.6931471805599453f;
                *(f32x4 *)&sx[row * CDP + e0] = acc;
            }
        }
    }
    __syncthreads();

    // ---- pass 1: S steps, decays and dt*u cached in registers, sub-chunk folded to (P, H)
    f32x2 da[S][NP];
    float dtu[S];
    {
        f32x2 Hh[NP];
#pragma unroll
        for (int n = 0; n < NP; ++n) Hh[n] = f32x2{0.f, 0.f};
        float sdt = 0.f;
#pragma unroll
        for (int s = 0; s < S; ++s) {
            const int row = sc * S + s;
            const float *xr = &sx[row * CDP];
            const float u = __builtin_bit_cast(float, (u32)(*(const unsigned short *)(tile + row * TROW + swz(row, d >> 3) + (d & 7) * 2)) << 16);
            f32x2 dv2 = {bias, 0.f};
#pragma unroll
            for (int r = 0; r < RP; ++r) dv2 = w[r] * f32x2{xr[2 * r], xr[2 * r + 1]} + dv2;
            const float dt = __builtin_amdgcn_logf(1.0f + __builtin_amdgcn_exp2f(fminf(dv2.x + dv2.y, 126.f)));
            sdt += dt;
            dtu[s] = dt * u;
#pragma unroll
            for (int n = 0; n < NP; ++n) {
                const f32x2 t = a2[n] * dt;
                da[s][n] = f32x2{__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y)};
                Hh[n] = da[s][n] * Hh[n] + f32x2{xr[R + 2 * n], xr[R + 2 * n + 1]} * dtu[s];
            }
        }
        chS[sc][d] = sdt;
#pragma unroll
        for (int n = 0; n < NP; ++n) {
            const f32x2 t = a2[n] * sdt;
            chP[sc][2 * n][d] = __builtin_amdgcn_exp2f(t.x);
            chP[sc][2 * n + 1][d] = __builtin_amdgcn_exp2f(t.y);
            chH[sc][2 * n][d] = Hh[n].x;
            chH[sc][2 * n + 1][d] = Hh[n].y;
        }
    }
    __syncthreads();

    // ---- sub-chunk prefix of this wave (P_pre, H_pre over sub-chunks < sc, in order)
    f32x2 Ppre[NP], Hpre[NP];
#pragma unroll
    for (int n = 0; n < NP; ++n) { Ppre[n] = f32x2{1.f, 1.f}; Hpre[n] = f32x2{0.f, 0.f}; }
#pragma unroll
    for (int s2 = 0; s2 < W - 1; ++s2) {
        if (s2 < sc) {                               // wave-uniform
#pragma unroll
            for (int n = 0; n < NP; ++n) {
                const f32x2 P = {chP[s2][2 * n][d], chP[s2][2 * n + 1][d]}, Hv = {chH[s2][2 * n][d], chH[s2][2 * n + 1][d]};
                Hpre[n] = P * Hpre[n] + Hv;
                Ppre[n] = Ppre[n] * P;
            }
        }
    }
    // ---- look-back mapping: thread = (channel lc, state group ng of NSL states)
    const int lc = tid % DCH, ng = tid / DCH;
    float an[NSL], curH[NSL];
    float cur_s = 0.f;
#pragma unroll
    for (int i = 0; i < NSL; ++i) {
        const int n = ng * NSL + i;
        an[i] = p.A[((int64_t)k * DCH + lc) * N + n];
        float Ht = 0.f;
#pragma unroll
        for (int s2 = 0; s2 < W; ++s2) Ht = chP[s2][n][lc] * Ht + chH[s2][n][lc];
        curH[i] = Ht;
    }
#pragma unroll
    for (int s2 = 0; s2 < W; ++s2) cur_s += chS[s2][lc];
    unsigned *flags = p.ctrl + 4;
    unsigned *tmo = p.ctrl + 1;
    const int64_t aent = (int64_t)NA * DCH;
    auto cnt = [&](int lv) -> int { return (p.nch >> lv) + 1; };
    auto publish = [&](int lv, int slot, float s_, const float (&Hv)[NSL]) {
        float *dst = p.agg + ((int64_t)p.agg_off[lv] + (int64_t)bk * cnt(lv) + slot) * aent;
        if (ng == 0) st_sc1(dst + lc, s_);
#pragma unroll
        for (int i = 0; i < NSL; ++i) st_sc1(dst + (int64_t)(1 + ng * NSL + i) * DCH + lc, Hv[i]);
        drain_vm();                                  // every storing wave, before the barrier in front of the flag store
        __syncthreads();
        if (tid == 0) __hip_atomic_store(flags + p.flag_off[lv] + bk * cnt(lv) + slot, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    if (!(p.dbg & 1)) publish(0, c, cur_s, curH);
    // upward: the aggregate of the 2^lv chunks that end here
#pragma unroll 1
    for (int lv = 1; lv < NLV; ++lv) {
        if (((j + 1) & ((1 << lv) - 1)) != 0 || (p.dbg & 1)) break;            // workgroup-uniform
        const int lslot = (c - (1 << (lv - 1))) >> (lv - 1);
        if (wave == 0 && !(p.dbg & 2)) {
            const int idx[8] = {p.flag_off[lv - 1] + bk * cnt(lv - 1) + lslot, 0, 0, 0, 0, 0, 0, 0};
            poll_flags(flags, idx, 1, tmo, lane);
        }
        __syncthreads();
        const float *src = p.agg + ((int64_t)p.agg_off[lv - 1] + (int64_t)bk * cnt(lv - 1) + lslot) * aent;
        const float ls = ld_sc1(src + lc);
#pragma unroll
        for (int i = 0; i < NSL; ++i) {
            const float lh = ld_sc1(src + (int64_t)(1 + ng * NSL + i) * DCH + lc);
            curH[i] = __builtin_amdgcn_exp2f(an[i] * cur_s) * lh + curH[i];     // left block first, then this one
        }
        cur_s = ls + cur_s;
        publish(lv, c >> lv, cur_s, curH);
    }
    // ---- carry-in: supergroup carry-in composed with the Fenwick blocks of j, left to right
    // (binary decomposition of j from the most significant bit: the blocks come out left to right)
    if (wave == 0 && (j > 0 || sg > 0) && !(p.dbg & 3)) {
        int idx[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        int nf = 0, pos = 0;
#pragma unroll
        for (int lv = NLV - 2; lv >= 0; --lv) {
            if ((j >> lv) & 1) {
                const int slot = (sg * SGC + pos + (1 << lv) - 1) >> lv;
#pragma unroll
                for (int q = 0; q < 8; ++q) idx[q] = q == nf ? p.flag_off[lv] + bk * cnt(lv) + slot : idx[q];
                ++nf;
                pos += 1 << lv;
            }
        }
        if (sg > 0) {
#pragma unroll
            for (int q = 0; q < 8; ++q) idx[q] = q == nf ? p.flag_off[NLV] + bk * (p.nsg + 1) + sg : idx[q];
            ++nf;
        }
        poll_flags(flags, idx, nf, tmo, lane);
    }
    __syncthreads();
    float carry[NSL];
    {
        const float *tsrc = p.top + ((int64_t)bk * (p.nsg + 1) + sg) * (int64_t)N * DCH;
#pragma unroll
        for (int i = 0; i < NSL; ++i) carry[i] = (sg > 0 && !(p.dbg & 1)) ? ld_sc1(tsrc + (int64_t)(ng * NSL + i) * DCH + lc) : 0.f;
        int pos = 0;
#pragma unroll
        for (int lv = NLV - 2; lv >= 0; --lv) {
            if (((j >> lv) & 1) && !(p.dbg & 1)) {                     // workgroup-uniform
                const int slot = (sg * SGC + pos + (1 << lv) - 1) >> lv;
                pos += 1 << lv;
                const float *src = p.agg + ((int64_t)p.agg_off[lv] + (int64_t)bk * cnt(lv) + slot) * aent;
                const float bs = ld_sc1(src + lc);
#pragma unroll
                for (int i = 0; i < NSL; ++i) {
                    const float bh = ld_sc1(src + (int64_t)(1 + ng * NSL + i) * DCH + lc);
                    carry[i] = __builtin_amdgcn_exp2f(an[i] * bs) * carry[i] + bh;
                }
            }
        }
#pragma unroll
        for (int i = 0; i < NSL; ++i) s_carry[ng * NSL + i][lc] = carry[i];
    }
    if (j == SGC - 1 && c + 1 < p.nch && !(p.dbg & 1)) {             // the next supergroup's carry-in = through the end of this chunk
        float ownH[NSL], own_s = 0.f;
#pragma unroll
        for (int s2 = 0; s2 < W; ++s2) own_s += chS[s2][lc];
        float *dst = p.top + ((int64_t)bk * (p.nsg + 1) + sg + 1) * (int64_t)N * DCH;
#pragma unroll
        for (int i = 0; i < NSL; ++i) {
            const int n = ng * NSL + i;
            float Ht = 0.f;
#pragma unroll
            for (int s2 = 0; s2 < W; ++s2) Ht = chP[s2][n][lc] * Ht + chH[s2][n][lc];
            ownH[i] = Ht;
            st_sc1(dst + (int64_t)n * DCH + lc, __builtin_amdgcn_exp2f(an[i] * own_s) * carry[i] + ownH[i]);
        }
        drain_vm();
        __syncthreads();
        if (tid == 0) __hip_atomic_store(flags + p.flag_off[NLV] + bk * (p.nsg + 1) + sg + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();

    // ---- pass 2: the S steps again from the carry-in, with the cached decays; y over the u tile
    {
        f32x2 h[NP];
#pragma unroll
        for (int n = 0; n < NP; ++n) h[n] = Ppre[n] * f32x2{s_carry[2 * n][d], s_carry[2 * n + 1][d]} + Hpre[n];
#pragma unroll
        for (int s = 0; s < S; ++s) {
            const int row = sc * S + s;
            const float *xr = &sx[row * CDP];
            unsigned short *up = (unsigned short *)(tile + row * TROW + swz(row, d >> 3) + (d & 7) * 2);
            const float u = __builtin_bit_cast(float, (u32)(*up) << 16);
            f32x2 acc = {0.f, 0.f};
#pragma unroll
            for (int n = 0; n < NP; ++n) {
                h[n] = da[s][n] * h[n] + f32x2{xr[R + 2 * n], xr[R + 2 * n + 1]} * dtu[s];
                acc = h[n] * f32x2{xr[R + N + 2 * n], xr[R + N + 2 * n + 1]} + acc;
            }
            const bf16 yv = (bf16)(acc.x + acc.y + Dd * u);
            *up = __builtin_bit_cast(unsigned short, yv);
        }
    }
    __syncthreads();
    // ---- y tile -> HBM at the merged pixel positions (EfficientMerge = the same index map), 16-byte stores
    bf16 *yb = p.y + (int64_t)b * p.H * p.W * DCH;
    {
        constexpr int RPP = NTHR / TCH;
#pragma unroll
        for (int r0 = 0; r0 < CL; r0 += RPP) {
            const int row = r0 + tid / TCH, ch = tid % TCH;
            if (RPP <= CL - r0 || row < CL)
                *(u32x4 *)(yb + (int64_t)spix[row] * DCH + ch * 8) = *(const u32x4 *)(tile + row * TROW + swz(row, ch));
        }
    }
}

struct FusedCfg { int S, W; };
// (N, R, D) instances built; chunk length CL = S * W
inline bool fused_cfg(int D, int N, int R, FusedCfg &c) {
    if (D == 128 && N == 4 && R == 4) {
        static const int v = [] { const char *e = getenv("FD_SCANF_VAR"); return e ? atoi(e) : 0; }();     // development: tile variants
        c = v == 1 ? FusedCfg{8, 4} : FusedCfg{16, 4};
        return true;
    }
    return false;
}

struct FusedLayout { int64_t ctrl_words, agg_entries, top_entries; int flag_off[NLV + 1], agg_off[NLV]; int nch, nbk, nsg; };
inline FusedLayout fused_layout(int B, int L, int CL) {
    FusedLayout q;
    q.nch = L / CL; q.nbk = B * 4; q.nsg = (q.nch + SGC - 1) / SGC;
    int64_t f = 0, a = 0;
    for (int lv = 0; lv < NLV; ++lv) {
        const int cntl = (q.nch >> lv) + 1;
        q.flag_off[lv] = (int)f; q.agg_off[lv] = (int)a;
        f += (int64_t)q.nbk * cntl; a += (int64_t)q.nbk * cntl;
    }
    q.flag_off[NLV] = (int)f;
    f += (int64_t)q.nbk * (q.nsg + 1);
    q.ctrl_words = ((4 + f + 3) / 4) * 4;            // ticket, timeout, 2 pad words, flags; a multiple of 16 bytes
    q.agg_entries = a;
    q.top_entries = (int64_t)q.nbk * (q.nsg + 1);
    return q;
}

}  // namespace

// 1 if fd_selective_scan_fused serves this shape: bf16, even image sizes, (d_inner, d_state, dt_rank) of an instance,
// sequence length a multiple of the chunk.  A function of the shape only (never of the batch).
extern "C" int fd_selective_scan_fused_ok(int dtype_opts, int D, int N, int R, int H, int W) {
    static const bool off = getenv("FD_SCAN_NO_FUSED") != nullptr;       // development switch: the 3-phase form
    FusedCfg c;
    if (off || (dtype_opts & 0xff) != FD_BF16 || ((H | W) & 1) || !fused_cfg(D, N, R, c)) return 0;
    const int L = (H / 2) * (W / 2), CL = c.S * c.W;
    return L % CL == 0 && (int64_t)H * W * D * 2 < (1ll << 31);
}

extern "C" int64_t fd_scan_fused_ws_floats(int B, int H, int W, int D, int N, int R) {
    FusedCfg c;
    if (!fused_cfg(D, N, R, c)) return 0;
    const FusedLayout q = fused_layout(B, (H / 2) * (W / 2), c.S * c.W);
    return q.ctrl_words + q.agg_entries * (int64_t)(N + 1) * D + q.top_entries * (int64_t)N * D;
}

extern "C" int fd_selective_scan_fused(int dtype, const void *xc, const void *x_proj_w, const float *dtw, const float *dtb,
                                       const float *A, const float *Ds, void *y, float *ws, int B, int H, int W, int D,
                                       int N, int R, void *stream) {
    FD_REQUIRE(fd_selective_scan_fused_ok(dtype, D, N, R, H, W), "fd_selective_scan_fused: unsupported shape (bf16, even "
               "H / W, built (D, N, R), L %% chunk == 0): D=%d N=%d R=%d H=%d W=%d", D, N, R, H, W);
    FD_REQUIRE(xc && x_proj_w && dtw && dtb && A && Ds && y && ws, "fd_selective_scan_fused: null pointer");
    FD_REQUIRE(((uintptr_t)x_proj_w & 15) == 0 && ((uintptr_t)xc & 15) == 0 && ((uintptr_t)y & 15) == 0 && ((uintptr_t)ws & 15) == 0,
               "fd_selective_scan_fused: 16-byte alignment");
    FusedCfg c;
    fused_cfg(D, N, R, c);
    const int L = (H / 2) * (W / 2);
    const FusedLayout q = fused_layout(B, L, c.S * c.W);
    FusedParams p;
    p.xc = (const bf16 *)xc; p.xw = (const bf16 *)x_proj_w; p.dtw = dtw; p.dtb = dtb; p.A = A; p.Ds = Ds; p.y = (bf16 *)y;
    p.ctrl = (unsigned *)ws;
    p.agg = ws + q.ctrl_words;
    p.top = p.agg + q.agg_entries * (int64_t)(N + 1) * D;
    p.B = B; p.H = H; p.W = W; p.D = D; p.H2 = H / 2; p.W2 = W / 2; p.L = L; p.nch = q.nch; p.nbk = q.nbk; p.nsg = q.nsg;
    for (int i = 0; i <= NLV; ++i) p.flag_off[i] = q.flag_off[i];
    for (int i = 0; i < NLV; ++i) p.agg_off[i] = q.agg_off[i];
    static const int dbg = [] { const char *e = getenv("FD_SCANF_DBG"); return e ? atoi(e) : 0; }();
    p.dbg = dbg;
    hipStream_t s = (hipStream_t)stream;
    // ticket, timeout word and every flag are zeroed before EVERY launch (a memset node under graph capture)
    if (hipMemsetAsync(ws, 0, (size_t)q.ctrl_words * 4, s) != hipSuccess) {
        fd_set_error("fd_selective_scan_fused: hipMemsetAsync failed");
        return FD_ERR_LAUNCH;
    }
    const dim3 grid((unsigned)(q.nch * q.nbk));
    if (c.S == 16) hipLaunchKernelGGL((scan_fused_kernel<4, 4, 128, 16, 4, 4>), grid, dim3(512), 0, s, p);
    else hipLaunchKernelGGL((scan_fused_kernel<4, 4, 128, 8, 4, 6>), grid, dim3(512), 0, s, p);
    FD_LAUNCH_OK("fd_selective_scan_fused");
    return FD_OK;
}
