// fd_pwgemm.hip -- persistent pointwise GEMM for the wide 1x1 convolutions of the low-resolution levels (round 4).
//
// in_proj / qkv / out_proj of the 256- and 512-channel Mamba blocks (src/emamba2.py:717,748, src/DADiff.py:266,283) are
// [32768..131072 pixels] x [256..2048 channels] x K = 256..2048 GEMMs.  The generic 256x256 tile (fd_conv.hip,
// conv_igemm_kernel<bf16,256,256,4,2,PW>) runs their K loop at 0.70 of the MFMA peak but one workgroup per CU does
// prologue -> K loop -> epilogue with nothing to overlap: the first loads' latency and the 128 KB of output stores of
// every tile ADD to the K loop (0.27-0.31 of the peak for the whole kernel).  Here:
//   * one PERSISTENT workgroup per CU walks its tiles; the operands of stage s + 3 are requested while stage s is in
//     the MFMAs -- across tile boundaries too, so a tile never starts cold;
//   * operands go global -> LDS by DMA (global_load_lds_dwordx4, XOR swizzle applied on the source side) into a ring
//     of four 32 KB stages of ONE K32 step each (64-byte rows: 256 pixel rows + 256 weight rows): no staging
//     registers, counted s_waitcnt vmcnt;
//   * DEFERRED STORES: at the end of a tile the accumulators are converted (bias / SiLU / gate + residual) to packed
//     bf16; half of the 16 store instructions leave right there (nothing waits for them), the other half stays in 32
//     registers and leaves one per stage during the NEXT tile's first eight stages;
//   * XCD-aware schedule: the column tiles of one 256-pixel row block run at the same time on CUs of the SAME XCD
//     (workgroup w -> XCD w % 8): the row block comes from HBM once per XCD and from that XCD's L2 for its siblings.
// Same K order per output element as the generic tiles: bitwise identical results.
#include "fd_common.h"
#include <type_traits>

namespace {

constexpr int PG_T = 256;                      // tile: 256 pixels x TNC channels (TNC = 256: one 8-wave workgroup per CU;
                                               //   TNC = 128: two 4-wave workgroups per CU, one's epilogue under the other's MFMAs)
constexpr int PG_RB = 64;                      // bytes of K per LDS row: one K32 step of bf16

// 16-byte chunk c of 64-byte row r: conflict-free for the 16-row x 4-chunk fragment reads (round-3 ring experiment)
__device__ __forceinline__ int pg_swz(int row, int chunk) { return (chunk ^ ((-(row >> 2)) & 3)) << 4; }

// s_waitcnt vmcnt(n) alone for the counts that occur (a wave-uniform runtime value: the immediate must be a constant)
#define PG_WAIT_VM(n) __builtin_amdgcn_s_waitcnt(((n) & 0xF) | (((n) >> 4) << 14) | (0x7 << 4) | (0xF << 8))
__device__ __forceinline__ void pg_wait_vm(int n) {
    switch (n) {
    case 0: PG_WAIT_VM(0); break;
    case 1: PG_WAIT_VM(1); break;
    case 2: PG_WAIT_VM(2); break;
    case 4: PG_WAIT_VM(4); break;
    case 5: PG_WAIT_VM(5); break;
    case 6: PG_WAIT_VM(6); break;
    case 7: PG_WAIT_VM(7); break;
    case 8: PG_WAIT_VM(8); break;
    case 9: PG_WAIT_VM(9); break;
    case 10: PG_WAIT_VM(10); break;
    case 12: PG_WAIT_VM(12); break;
    case 13: PG_WAIT_VM(13); break;
    case 14: PG_WAIT_VM(14); break;
    case 15: PG_WAIT_VM(15); break;
    case 16: PG_WAIT_VM(16); break;
    case 17: PG_WAIT_VM(17); break;
    case 18: PG_WAIT_VM(18); break;
    default: PG_WAIT_VM(0); break;               // (not reached: DST x requests + stores + 8 [end-of-tile stores still younger])
    }
}

constexpr int PG_MAXC = 2048, PG_MAXG = 4096;     // bias: Cout floats; gate (GATE_RES): B x Cout floats (fd_pwgemm_ok)

// EPI: the epilogue as a template parameter (FD_EPI_NONE / _SILU_SPLIT / _GATE_RES): as a runtime value it was ~60 scalar
// branches per tile.  Only TNC = 256 is instantiated: two 4-wave workgroups per CU on 256 x 128 tiles (one's epilogue under
// the other's MFMAs) measured 0-8 % slower at every shape of the forward.
template <int TNC, int EPI>
__global__ __launch_bounds__(TNC * 2, 2) void pw_gemm_kernel(const fd_conv_params p, const int TM, const int TN, const int nst) {
    constexpr int NWV = TNC / 32, WNW = TNC / 128;                        // waves (8 | 4) as 4 x WNW of 64 pixels x 128 channels
    constexpr int PG_NS = TNC == 256 ? 4 : 3, LOOK = PG_NS - 1;           // ring slots; stages requested ahead of the one in the MFMAs
    constexpr int PG_SLOT = (PG_T + TNC) * PG_RB;                         // 32 | 24 KB: 256 pixel rows, then TNC weight rows
    constexpr int DPX = PG_T / 16 / NWV, DST = DPX + 2;                   // DMA instructions per wave and stage: pixel rows | all
    __shared__ __attribute__((aligned(1024))) unsigned char ring[PG_NS * PG_SLOT];
    // bias and gate of the WHOLE layer, staged once: staged per tile they were a global load in front of an LDS store, whose
    // s_waitcnt vmcnt(0) -- the compiler does not know the DMAs in flight -- drained the three requested stages at the start
    // of every tile
    __shared__ __attribute__((aligned(16))) float s_bias[PG_MAXC];
    __shared__ __attribute__((aligned(16))) float s_gate[EPI == FD_EPI_GATE_RES ? PG_MAXG : 4];
    static_assert(PG_NS * PG_SLOT + (PG_MAXC + PG_MAXG) * sizeof(float) <= 160 * 1024 - 1024,
                  "pw_gemm_kernel: ring + bias + gate tables must fit a CU's 160 KB of LDS (the gated form sits at 152 KB)");
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WNW, wn = wave % WNW;
    const int fr = lane & 15, fg = lane >> 4;
    const int K = p.c0, OHW = p.OH * p.OW, mpi = OHW / PG_T;              // row blocks per image
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3, per_xcd = gridDim.x >> 3;
    const int ntile_x = ((TM - xcd + 7) >> 3) * TN;                       // tiles of this XCD: row blocks tm = xcd (mod 8)
    const int rot_every = per_xcd / TN > 0 ? per_xcd / TN : 1;
    const bf16 *A = (const bf16 *)p.in0;
    const bf16 *Wt = (const bf16 *)p.weight;
    bf16 *O = (bf16 *)p.out;

    // ---- DMA roles.  A wave instruction fills 1 KiB = 16 rows x 64 B: lane -> (row lane / 4, physical chunk lane % 4),
    // and fetches the LOGICAL chunk that belongs there.  Wave w fills pixel rows [16 DPX w, 16 DPX (w + 1)) and weight rows
    // [32 w, 32 w + 32).
    const int drow = lane >> 2;
    const int dchunk = (lane & 3) ^ ((-(lane >> 4)) & 3);                 // (row >> 2) & 3 == lane >> 4 (row bases are 0 mod 16)
    unsigned voff_px[DPX], voff_w[2];
#pragma unroll
    for (int h = 0; h < DPX; ++h) {
        const int r = 16 * DPX * wave + 16 * h + drow;                    // tile row
        voff_px[h] = (unsigned)((r * p.ld0 + 8 * dchunk) * 2);
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int r = 32 * wave + 16 * h + drow;                          // weight row of the tile
        // weight row R of a 32-row group holds output channel ((R >> 2) & 3) * 8 + ((R >> 4) & 1) * 4 + (R & 3): the
        // transposed MFMA then leaves 8 CONSECUTIVE channels of one pixel in a lane (fd_conv.hip, PWE)
        const int R = r & 31, n = (r & ~31) | (((R >> 2) & 3) << 3) | (((R >> 4) & 1) << 2) | (R & 3);
        voff_w[h] = (unsigned)((n * K + 8 * dchunk) * 2);
    }
    const unsigned lds_px = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void *)ring + wave * (DPX * 1024);
    const unsigned lds_wt = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void *)ring + PG_T * PG_RB + wave * 2048;
    auto dma = [&](const char *base, unsigned voff, unsigned dst) {       // dst: wave-uniform LDS byte address (M0)
        unsigned m0_saved;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3\n\ts_mov_b32 m0, %0"
                     : "=&s"(m0_saved) : "s"(dst), "v"(voff), "s"(base) : "memory");
    };
    // producer cursor: the stage that is requested next (tile pq of this XCD's list, K32 step pks)
    int pq = idx, pks = 0, pslot = 0;
    const char *pa = nullptr, *pw = nullptr;                              // operand rows of the stage requested next (wave-uniform)
    auto produce = [&]() -> bool {                                        // request one stage; false when there is none left
        if (pq >= ntile_x) return false;
        if (pks == 0) {
            // the tile's coordinates once per tile, not per stage: four integer divisions by runtime values are ~150 instructions,
            // issued by every wave in front of the stage's MFMAs while -- one barrier per stage -- no other wave has matrix work either
            const int tm = (pq / TN) * 8 + xcd, tn = (pq - (pq / TN) * TN + (pq / TN) / rot_every) % TN;      // (column tiles rotate, see below)
            const int b = tm / mpi, m0 = (tm - b * mpi) * PG_T;
            pa = (const char *)(A + ((int64_t)b * OHW + m0) * p.ld0 + p.off0);
            pw = (const char *)(Wt + (int64_t)b * p.w_batch_stride + (int64_t)tn * TNC * K);
        }
#pragma unroll
        for (int h = 0; h < DPX; ++h) dma(pa, voff_px[h], lds_px + pslot * PG_SLOT + 1024 * h);
#pragma unroll
        for (int h = 0; h < 2; ++h) dma(pw, voff_w[h], lds_wt + pslot * PG_SLOT + 1024 * h);
        pslot = pslot + 1 == PG_NS ? 0 : pslot + 1;
        pa += 2 * 32;                                                     // the next K32 step
        pw += 2 * 32;
        if (++pks == nst) { pks = 0; pq += per_xcd; }
        return true;
    };

    f32x4 acc[4][8];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    u32x4 outq[8];                                                        // the second half of the previous tile's outputs, packed bf16
    bool pending = false;
    int sb = 0, sm0 = 0, stn = 0;                                         // coordinates of the tile waiting in outq
    // Half of a tile's 16 store instructions (channel pairs jp = 0, 1) leave at the end of the tile, the other half
    // (jp = 2, 3) waits in 32 registers and leaves one per stage during the NEXT tile's first eight stages.  (All 16
    // deferred would be 64 registers next to 128 accumulators: the kernel spilled, and a scratch access is a VMEM
    // operation that the counted waits below would have to account for.)
    auto store_q = [&](int u) {                                           // deferred store u of 8 (u: compile-time)
        const int jp = 2 + (u >> 2), i = u & 3;
        const int m = sm0 + 64 * wm + 16 * i + fr, n0 = stn * TNC + 128 * wn + 32 * jp + 8 * fg;
        *(u32x4 *)(O + ((int64_t)sb * OHW + m) * p.ldo + p.offo + n0) = outq[u];
    };

    // fragment offsets inside a slot (bytes): pixel rows 64 wm + 16 i + fr, weight rows 128 wn + 16 j + fr
    int poff[4], woff0;
#pragma unroll
    for (int i = 0; i < 4; ++i) { const int r = 64 * wm + 16 * i + fr; poff[i] = r * PG_RB + pg_swz(r, fg); }
    { const int r = 128 * wn + fr; woff0 = PG_T * PG_RB + r * PG_RB + pg_swz(r, fg); }     // + 16 j rows: (r + 16 j) >> 2 = same mod 4

    for (int n = tid; n < p.Cout; n += TNC * 2) s_bias[n] = p.bias ? p.bias[n] : 0.f;
    if constexpr (EPI == FD_EPI_GATE_RES)
        for (int i = tid; i < p.B * p.Cout; i += TNC * 2) s_gate[i] = p.gate[(int64_t)(i / p.Cout) * p.gate_ld + i % p.Cout];
    __syncthreads();
    // stages 0 .. LOOK - 1 in flight before the first step
    // history for the counted waits: dA / dB = was stage gs + 1 / gs + 2 requested, sA / sB = stores issued 2 / 1 steps ago
    // (LOOK = 2: one step of history -- dA and sA stay false, dB / sB are the step before)
    bool dA = false, dB = false, sA = false, sB = false;
    produce();
    if constexpr (LOOK == 3) { dA = produce(); dB = produce(); }
    else dB = produce();
    int e3 = 0;                                                           // steps for which the 8 stores of a tile's end are still younger than the awaited stage
    int slot = 0;
    for (int q = idx; q < ntile_x; q += per_xcd) {
        // the column tile of a row block's siblings rotates by one every `rot_every` row blocks (about once per round;
        // constant inside a sibling group, so the tile map stays a bijection): with TN a divisor of the workgroups per XCD
        // a workgroup would otherwise keep ONE column tile for all its rounds, and the column tiles are not equally
        // expensive (SILU_SPLIT applies the SiLU to the upper half of the channels only)
        const int tm = (q / TN) * 8 + xcd, tn = (q - (q / TN) * TN + (q / TN) / rot_every) % TN;
        const int b = tm / mpi, m0 = (tm - b * mpi) * PG_T;
        for (int g8 = 0; g8 < nst; g8 += 8) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                // stage `slot` has landed once at most the operations issued after its request are outstanding
                {
                    // (the two values of a tile's steady state first: as one switch the wait was a tree of ~10 scalar branches
                    //  per stage, 14 % of the K loop's time)
                    const int wv = DST * ((int)dA + (int)dB) + ((int)sA + (int)sB) + (e3 > 0 ? 8 : 0);
                    if (wv == (LOOK - 1) * DST) PG_WAIT_VM((LOOK - 1) * DST);
                    else if (wv == (LOOK - 1) * DST + (LOOK - 1)) PG_WAIT_VM((LOOK - 1) * DST + (LOOK - 1));
                    else pg_wait_vm(wv);
                }
                e3 = e3 > 0 ? e3 - 1 : 0;
                __builtin_amdgcn_s_barrier();
                const unsigned char *sl = ring + slot * PG_SLOT;
                {
                    // the stage's first fragments are requested right behind the barrier: their LDS round trip runs under the
                    // scalar work of the DMA requests below (every wave of the workgroup is at this point together -- one
                    // barrier per stage -- so nothing else would cover it)
                    bf16x8 pf[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) pf[i] = *(const bf16x8 *)(sl + poff[i]);
                    // weight fragments TWO ahead of their MFMAs (an LDS round trip under load is ~200 cycles: one group of four
                    // MFMAs per wave, 2 x 64 cycles with the SIMD's other wave, did not cover it -- the K loop alone ran at 0.48
                    // of the peak), never all eight at once
                    bf16x8 wn_ = *(const bf16x8 *)(sl + woff0), wn2_ = *(const bf16x8 *)(sl + woff0 + 16 * PG_RB);
                    const bool st = pending && g8 == 0;
                    if (st) store_q(u);
                    const bool dn = produce();                            // stage gs + LOOK into the slot stage gs - 1 was read from
                    if constexpr (LOOK == 3) { dA = dB; sA = sB; }
                    dB = dn; sB = st;
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const bf16x8 wf = wn_;
                        wn_ = wn2_;
                        if (j + 2 < 8) wn2_ = *(const bf16x8 *)(sl + woff0 + 16 * (j + 2) * PG_RB);
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                            // inline asm with the accumulator TIED (dst = src C): through the builtin hipcc 7.2 renames the
                            // accumulators across the unrolled stages (dst != src C) and spills ~70 registers.  Hazards by
                            // hand: the fragments come from ds_read (the compiler's s_waitcnt covers asm operands), the
                            // same accumulator recurs 32 MFMAs later, and the epilogue's first VALU read of the
                            // accumulators sits behind an s_nop block.
                            asm(FD_MFMA16_ASM " %0, %1, %2, %0" : "+v"(acc[i][j]) : "v"(wf), "v"(pf[i]));
                    }
                }
                slot = slot + 1 == PG_NS ? 0 : slot + 1;
            }
            if (g8 == 0) pending = false;
        }
        // the last inline-asm MFMAs must have written their accumulators before the epilogue's VALU reads them
        // (every accumulator an operand of the nop block or of the empty volatile asm behind it: fd_common.h)
        asm volatile(FD_MFMA_ASM_DRAIN : FD_TIE8(acc[0]), FD_TIE8(acc[1]) :: "memory");
        asm volatile("" : FD_TIE8(acc[2]), FD_TIE8(acc[3]));
        // ---- end of tile: accumulators -> packed outputs (lane = pixel fr of row block i; 8 consecutive channels per
        // pair of permuted 16-channel tiles), stored during the next tile's first eight stages
#pragma unroll
        for (int jp = 0; jp < 4; ++jp) {
            const int cl = 128 * wn + 32 * jp + 8 * fg, n0 = tn * TNC + cl;
            float bias8[8], gate8[8];
            load8(&s_bias[n0], bias8);
            if constexpr (EPI == FD_EPI_GATE_RES) load8(&s_gate[b * p.Cout + n0], gate8);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float val[8];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    val[e] = acc[i][2 * jp][e] + bias8[e];
                    val[4 + e] = acc[i][2 * jp + 1][e] + bias8[4 + e];
                    acc[i][2 * jp][e] = 0.f;
                    acc[i][2 * jp + 1][e] = 0.f;
                }
                if constexpr (EPI == FD_EPI_SILU_SPLIT) {
                    if (n0 >= p.epi_split) fd_silu8(val);
                } else if constexpr (EPI == FD_EPI_GATE_RES) {
                    const int m = m0 + 64 * wm + 16 * i + fr;
                    float rs[8];
                    load8((const bf16 *)p.res + ((int64_t)b * OHW + m) * p.ld_res + p.off_res + n0, rs);
#pragma unroll
                    // an explicit fma, here and in conv_igemm_kernel's epilogue: which tile a layer runs on depends on the batch, a
                    // slice's bits must not (and fd_cvt_h below: fd_common.h)
                    for (int e = 0; e < 8; ++e) val[e] = __builtin_fmaf(gate8[e], val[e], rs[e]);
                }
                bf16x8 pk;
#pragma unroll
                for (int e = 0; e < 8; ++e) pk[e] = fd_cvt_h(val[e]);
                if (jp < 2) {
                    const int m = m0 + 64 * wm + 16 * i + fr;
                    *(u32x4 *)(O + ((int64_t)b * OHW + m) * p.ldo + p.offo + n0) = __builtin_bit_cast(u32x4, pk);
                } else {
                    outq[(jp - 2) * 4 + i] = __builtin_bit_cast(u32x4, pk);
                }
            }
        }
        sb = b; sm0 = m0; stn = tn;
        pending = true;
        e3 = LOOK;
    }
    if (pending) {
#pragma unroll
        for (int u = 0; u < 8; ++u) store_q(u);
    }
}

}  // namespace

// 1 if `p` runs on the persistent pointwise GEMM: bf16, 1x1 / stride 1 / one source, K and Cout multiples of 256, whole
// 256-pixel row blocks per image, no GroupNorm sums, epilogue NONE / SILU_SPLIT / GATE_RES, enough tiles to fill the chip.
int fd_pwgemm_ok(const fd_conv_params &p) {
    const bool off = fd_dev(FD_DEV_NO_PWGEMM);            // development switch: the generic 256x256 tile
    if (off || p.dtype != FD_BF16 || p.out_f32 || p.ndir != 1 || p.prologue != FD_PRO_NONE) return 0;
    if (p.KH != 1 || p.KW != 1 || p.stride != 1 || p.pad_h != 0 || p.pad_w != 0 || p.upsample) return 0;
    if (p.OH != p.H || p.OW != p.W || p.c1 != 0 || p.in1) return 0;
    if (p.c0 % 256 || p.c0 < 256 || p.Cout % 256 || ((int64_t)p.OH * p.OW) % 256) return 0;
    if (p.stats_partial || p.weight_f8) return 0;
    if (p.epilogue != FD_EPI_NONE && p.epilogue != FD_EPI_SILU_SPLIT && p.epilogue != FD_EPI_GATE_RES) return 0;
    if (p.epilogue == FD_EPI_SILU_SPLIT && p.epi_split % 8) return 0;
    if (p.ld0 % 8 || p.off0 % 8 || p.ldo % 8 || p.offo % 8) return 0;
    if (p.epilogue == FD_EPI_GATE_RES && (!p.res || !p.gate || p.ld_res % 8 || p.off_res % 8)) return 0;
    if (p.Cout > PG_MAXC) return 0;     // the LDS bias table (the gate table holds B x Cout floats: larger batches launch in image groups)
    if ((int64_t)p.OH * p.OW * p.ld0 * 2 >= (1ll << 31) || (int64_t)p.Cout * p.c0 * 2 >= (1ll << 31)) return 0;   // 32-bit DMA offsets
    const int64_t tiles = (int64_t)p.B * ((int64_t)p.OH * p.OW / 256) * (p.Cout / 256);
    // K = 256 tiles with the SiLU epilogue are epilogue-bound (8 stages of MFMA against ~1000 VALU instructions per lane):
    // with 8 of them per workgroup the persistent form measured 4 % SLOWER than the generic tile (256 -> 1024 at 128x128,
    // batch 8: 146 vs 140 us); every other shape of the forward gains 2-13 %
    if (p.c0 == 256 && p.epilogue == FD_EPI_SILU_SPLIT && tiles > 1024) return 0;
    return tiles >= 192;
}

int fd_pwgemm_launch(const fd_conv_params &pp, hipStream_t s) {
    // GATE_RES stages gate[b][n] of the whole launch in LDS (PG_MAXG floats).  A batch beyond that (Cout 512 at sub-batch 16, Cout
    // 256 at 32 ...) used to leave this kernel for the generic tile without a word: it now runs as several launches
    // over groups of whole images -- every output element sees the same K order, the results are the same bits.
    if (pp.epilogue == FD_EPI_GATE_RES && (int64_t)pp.B * pp.Cout > PG_MAXG) {
        const int per = PG_MAXG / pp.Cout;               // >= 2: Cout <= PG_MAXC
        const int64_t hw = (int64_t)pp.OH * pp.OW;
        for (int b0 = 0; b0 < pp.B; b0 += per) {
            fd_conv_params q = pp;
            q.B = (pp.B - b0 < per) ? pp.B - b0 : per;
            q.in0 = (const bf16 *)pp.in0 + (int64_t)b0 * hw * pp.ld0;
            q.res = (const bf16 *)pp.res + (int64_t)b0 * hw * pp.ld_res;
            q.out = (bf16 *)pp.out + (int64_t)b0 * hw * pp.ldo;
            q.gate = pp.gate + (int64_t)b0 * pp.gate_ld;
            if (pp.w_batch_stride) q.weight = (const bf16 *)pp.weight + (int64_t)b0 * pp.w_batch_stride;
            fd_pwgemm_launch(q, s);
        }
        return 0;
    }
    const fd_conv_params &p = pp;
    const int TM = p.B * (int)((int64_t)p.OH * p.OW / 256), TN = p.Cout / 256, nst = p.c0 / 32;
    if (p.epilogue == FD_EPI_SILU_SPLIT) hipLaunchKernelGGL((pw_gemm_kernel<256, FD_EPI_SILU_SPLIT>), dim3(256), dim3(512), 0, s, p, TM, TN, nst);
    else if (p.epilogue == FD_EPI_GATE_RES) hipLaunchKernelGGL((pw_gemm_kernel<256, FD_EPI_GATE_RES>), dim3(256), dim3(512), 0, s, p, TM, TN, nst);
    else hipLaunchKernelGGL((pw_gemm_kernel<256, FD_EPI_NONE>), dim3(256), dim3(512), 0, s, p, TM, TN, nst);
    return 0;
}
