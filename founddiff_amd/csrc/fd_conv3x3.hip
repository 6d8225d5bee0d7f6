// fd_conv3x3.hip -- 3x3 / stride-1 / pad-1 convolution (62 % of the denoiser's FLOPs) as a
// halo-tiled implicit GEMM on MFMA, bf16.
//
// The generic kernel (fd_conv.hip) re-fetches the shifted A tile from L2 for each of the 9 taps
// and pays one global-load latency per K step.  Here a workgroup owns an 8 x 16 (or 16 x 16) pixel
// output tile; per 64-channel slab it loads the (TH+2) x (16+2) halo ONCE into LDS (coalesced 128-byte
// pixel rows, through the nearest-x2 up-sampling index map and the two-source concat when
// present) and all 9 taps read their A fragments straight from that halo tile: an MFMA fragment
// is "16 bytes of one pixel's channel vector", so a tap is just a different pixel offset -- no
// im2col copy exists anywhere.  Global->LDS traffic for A drops 9x -> 1.4x, the only per-tap
// traffic is the 8/16 KiB weight tile (L2-resident, register-prefetched one tap ahead), and the
// next slab's halo is in flight during the current slab's 9 taps.
// Epilogue straight from the accumulators (the weights are the MFMA's A operand: a lane holds 8 consecutive channels
// of a pixel, 16-byte stores, no LDS staging); deterministic per-tile GroupNorm partial sums.
#include "fd_common.h"
#include <type_traits>

namespace {

constexpr int TW = 16, HX = TW + 2;                // output tile width (pixels), halo width
constexpr int ROWB = 128;                          // bytes per pixel row of a 64-channel bf16 slab

// XOR swizzle of the 16-byte chunks of a 128-byte LDS row.  Enumerated against the lane groups ds_read_b128 is
// served in (MI355X_MICROARCH.md, LDS: {0-3,12-15,20-27}, {4-11,16-19,28-31}, ...): the generic kernel's
// (row >> 1) & 7 is conflict-free for rows that start on a multiple of 16 but costs 6.7 instead of 4 LDS cycles
// per read on the tap-shifted A fragments here (16 consecutive halo pixels starting at ANY row: 26 % of the LDS
// cycles were bank conflicts, PMC); row & 7 is conflict-free for every shift and for the weight tiles.  The fp8
// form reads chunk pairs (2 fg, 2 fg + 1): there the low three row bits rotated by one are conflict-free.
template <bool F8>
__device__ __forceinline__ int swz(int row, int chunk) {
    const int s = F8 ? (((row & 3) << 1) | ((row >> 2) & 1)) : (row & 7);
    return (chunk ^ s) << 4;
}

// Tile = TH x 16 output pixels x BN channels, 4 waves, every wave a 64-pixel x 64-channel sub-tile (4 tile
// rows x 16 px, MT = NT = 4: 8 fragment reads per 16 MFMAs -- a 32-channel wave tile needs 6 per 8 and
// saturates the LDS array):
//   <128, 8>   Cout > 64:  2 (row groups) x 2 (channel halves) waves
//   <64, 16>   Cout <= 64: 4 row groups, each wave all 64 channels (halo overhead 1.27x instead of 1.41x)
//   <64, 8>    Cout <= 64 when OH is not a multiple of 16: 2 x 2 waves of 64 px x 32 channels
// F8 (fp8 weights, BASELINE configs[4]): the same tile and ring with the K axis in 128-channel slabs.  A halo pixel
// row is still 128 bytes -- 128 e4m3 channels instead of 64 bf16 -- so the LDS image, its swizzle and the weight
// tiles ([BN rows][128 bytes] of the e4m3 matrix) keep their byte geometry.  The bf16 halo is converted (x act_scale)
// once per slab on its way into LDS; a lane's operand of v_mfma_scale_f32_16x16x128_f8f6f4 is 32 consecutive
// channels = two adjacent 16-byte chunks (same slot order for A and B: tools/probes/mfma_fp8_layout.hip), block
// scales 1.0 (e8m0 127); the per-output-channel weight scale is applied in the epilogue.  Per channel this is half
// the LDS fragment bytes, half the weight DMA and a quarter of the MFMA instructions of the bf16 form.
typedef __attribute__((ext_vector_type(8))) int i32x8;

// The weights are the MFMA's A operand (D = W . X^T: a lane's 4 accumulator values of a 16x16 tile are 4 ROWS = 4
// output channels of ONE pixel), and row R of a weight tile in LDS holds output channel
//   R = 32 a + 16 p + 4 b + e   ->   32 a + 8 b + 4 p + e
// so that the two tiles (p = 0, 1) of a 32-row group give lane group b = lane / 16 the 8 CONSECUTIVE channels
// 32 a + 8 b .. + 7 of its pixel: the epilogue stores 16 bytes per lane straight from the accumulators -- no LDS
// staging of the C tile, no barrier between the last MFMA and the stores.
__device__ __forceinline__ int tile_row_channel(int R) {
    return (R & ~31) | (((R >> 2) & 3) << 3) | (((R >> 4) & 1) << 2) | (R & 3);
}

// x summed over the 16 lanes of a DPP row, the total in every lane: xor 1, xor 2 (quad permutes), then the mirrored
// other quad of the 8-lane half and the mirrored other half of the row -- 4 v_add_f32_dpp, no LDS, a fixed order.
__device__ __forceinline__ float row16_sum(float x) {
    const auto dpp = [](float v, int ctrl) -> float {
        const int r = ctrl == 0xB1   ? __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true)
                      : ctrl == 0x4E ? __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true)
                      : ctrl == 0x141 ? __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true)
                                      : __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true);
        return __builtin_bit_cast(float, r);
    };
    x += dpp(x, 0xB1);        // quad_perm [1,0,3,2]
    x += dpp(x, 0x4E);        // quad_perm [2,3,0,1]
    x += dpp(x, 0x141);       // row_half_mirror
    x += dpp(x, 0x140);       // row_mirror
    return x;
}

// UP (round 5): the nearest x2 up-sampling convolutions (`Upsample`: nn.Upsample(scale_factor=2) -> Conv2d 3x3, src/DADiff.py:121-127)
// as FOUR 2x2 convolutions on the LOW-resolution input, one per output parity class (a, b) = (row & 1, column & 1): the three
// taps of a 3x3 row / column fall on only TWO source pixels (a = 0: kh = 0 -> source row i - 1, kh = 1, 2 -> i; a = 1: kh = 0, 1
// -> i, kh = 2 -> i + 1), so the taps that share a source pixel are summed into one weight at pack time (fp32, then one bf16
// rounding): 4 MACs per output pixel, input and output channel instead of 9 -- the same sums in exact arithmetic, borders
// included (a pixel the 3x3 reads from the zero padding is a source pixel outside the low-resolution image).  A workgroup owns a
// LOW-resolution tile; its units are (class, slab) with 4 taps each, all classes read the same (TH + 2) x 18 source halo; the
// weight matrix is [Cout][4 classes][2 x 2 taps][Cin] (fd_conv_params.weight_up2x).  With 4 taps per unit the 3-slot weight
// ring no longer starts every unit at slot 0: the slot offsets so[] rotate by one per unit.
// SPL (round 5; fp32 storage, the `fp32s` engine: the last step's outer levels and the parity-grade mode): the split-bf16
// contraction of fd_conv.hip on this kernel.  x = x_hi + x_lo, w = w_hi + w_lo in bf16; per 64-channel slab THREE units of 9
// taps accumulate x_hi.w_hi + x_hi.w_lo + x_lo.w_hi into the same fp32 accumulators (4.4e-6 per contraction against the
// exact-f32 MFMA): the fp32 halo (32 bytes per 8 channels: two loads per chunk, like the fp8 form) is converted to its hi or lo
// halves on its way into LDS, the weights come pre-split as two bf16 matrices (weight_split_hi / _lo), the LDS image, the ring
// and the tap loop are the bf16 ones.  fp32 output straight from the accumulators (a lane holds 8 consecutive channels of a
// pixel = 32 bytes; the four lane groups of a pixel fill a 128-byte line), GroupNorm partial sums as in the bf16 form.
template <int BN, int TH, bool F8, bool UP = false, bool SPL = false>
__global__ __launch_bounds__(256, 2) void conv3x3_halo_kernel(const fd_conv_params p, const int tpw, const int tiles_xy) {
    static_assert(!(UP && F8), "the up-sampling form is bf16 only");
    static_assert(!(SPL && F8), "the split form: fp32 storage, bf16 MFMAs");
    constexpr int NTAP = UP ? 4 : 9, NCLS = UP ? 4 : 1;
    constexpr int BM = TH * TW, HY = TH + 2, HP = HY * HX;
    constexpr int HL = (HP * 8 + 255) / 256;          // halo 16-byte LDS chunks per thread
    constexpr int SLABC = F8 ? 128 : 64;              // channels per K slab
    constexpr int HG = (F8 || SPL) ? 2 : 1;           // 16-byte global loads per LDS chunk
    constexpr int ESZ = F8 ? 1 : 2;                   // bytes per weight element
    constexpr int WMW = TH / 4, WNW = 4 / WMW;        // wave grid
    constexpr int NB = BN / 32, NT = BN / WNW / 16, MT = 4;
    constexpr int HALO_B = HP * ROWB;                 // 23040 (TH = 8) / 41472 (TH = 16)
    constexpr int WT_B = BN * ROWB;
    constexpr int NWB = 3;                         // weight-tile ring (LDS-DMA, two taps ahead)
    constexpr int LOOP_B = HALO_B + NWB * WT_B;    // ONE halo buffer (the next slab / tile waits in registers)
    constexpr int NPASS = BN / 64;                 // epilogue passes: the C tile is staged 64 channels at a time
    constexpr int RB = BM * 8 / 256;               // 16-byte chunks of a staged pass per thread
    static_assert(BM * 128 <= HALO_B, "the C staging of one pass lives in the (dead) halo buffer");
    __shared__ __attribute__((aligned(16))) unsigned char smem[LOOP_B];
    __shared__ float s_stat[4][BN][2];
    __shared__ __attribute__((aligned(16))) float s_bias[BN], s_wsc[BN];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tiles_x = (UP ? p.W : p.OW) / TW;
    // a workgroup owns tpw consecutive tiles (row-major: neighbours share halo columns through L2) of image b,
    // channel tile nt
    // UP: `tpw` is the number of parity classes per workgroup instead (4: one workgroup per tile; 1: a workgroup per (tile, class) --
    // the one-slice kernel set, where a low-resolution tile grid alone is 64..256 workgroups), one tile per workgroup
    const int cpw = UP ? tpw : 1, cls_begin = UP ? (int)(blockIdx.x % (NCLS / cpw)) * cpw : 0, cls_end = UP ? cls_begin + cpw : 1;
    const int t_begin = UP ? (int)(blockIdx.x / (NCLS / cpw)) : blockIdx.x * tpw, t_end = UP ? t_begin + 1 : min(t_begin + tpw, tiles_xy);
    const int nt = blockIdx.y, b = blockIdx.z;
    // (SPL: three units per 64-channel slab -- the "slab" index below is 3 * slab + part, part 0: x_hi.w_hi, 1: x_hi.w_lo, 2: x_lo.w_hi)
    const int Cin = p.c0 + p.c1, K = NTAP * NCLS * Cin, nslab = (SPL ? 3 : 1) * (Cin / SLABC);
    const int Hs = UP ? p.H : p.OH, Ws = UP ? p.W : p.OW;       // conv input grid == output grid (stride 1, pad 1); UP: the source grid
    constexpr int ISZ = SPL ? 2 : 1;                  // input elements are ISZ x 2 bytes
    const bf16 *in0 = (const bf16 *)p.in0 + ((int64_t)b * p.H * p.W * p.ld0 + p.off0) * ISZ;
    const bf16 *in1 = p.in1 ? (const bf16 *)p.in1 + ((int64_t)b * p.H * p.W * p.ld1 + p.off1) * ISZ : nullptr;
    // (UP && SPL: the sub-pixel matrix of an up-sampling convolution, split like the 9-tap one: weight_up2x_split_hi / _lo)
    const unsigned char *wgt = (const unsigned char *)(SPL ? (UP ? p.weight_up2x_split_hi : p.weight_split_hi)
                                                           : (UP ? p.weight_up2x : (F8 ? p.weight_f8 : p.weight)));
    const unsigned char *wgt_lo = (const unsigned char *)(SPL ? (UP ? p.weight_up2x_split_lo : p.weight_split_lo) : nullptr);

    // ---- halo loader: chunk ids hid = tid + 256*i -> (halo pixel, 16-byte channel chunk)
    // Every load is issued (from a clamped in-image address, zeroed on the LDS store where it was
    // padding): the s_waitcnt bookkeeping of the weight ring below counts wave-level VMEM instructions.
    // The geometry is recomputed per load / store from an OPAQUE copy of tid (~10 VALU per chunk against a slab's
    // 288 MFMAs): as loop invariants the HL source offsets and HL LDS addresses were kept in -- spilled -- registers,
    // and a scratch reload is a VMEM operation whose s_waitcnt vmcnt(0) also waits for the weight DMA just issued.
    // chunk i of a thread is halo pixel hp0 + 32 i: (row, column) advance by (1, 14) with a carry at 18 -- a division
    // only for chunk 0
    struct HaloPos { int hy, hx; };
    auto halo_first = [&](int tl) -> HaloPos {
        const int hp = tl >> 3;
        const int hy = hp / HX;
        return {hy, hp - hy * HX};
    };
    auto halo_next = [&](HaloPos &q) {
        q.hx += 32 - HX;
        q.hy += 1;
        if (q.hx >= HX) { q.hx -= HX; q.hy += 1; }
    };
    auto halo_geom = [&](int tl, int i, HaloPos q, int ty0, int tx0, bool interior, bool &valid) -> int {    // element offset of the (clamped) source pixel
        if (i == HL - 1 && (HL * 256) / 8 > HP) {        // chunks past the halo (never stored): any in-image address
            const bool over = (tl >> 3) + 32 * i > HP - 1;
            q.hy = over ? HY - 1 : q.hy;
            q.hx = over ? HX - 1 : q.hx;
        }
        int y = ty0 + q.hy - 1, x = tx0 + q.hx - 1;
        if (interior) {                                  // (wave-uniform) no halo pixel of this tile is padding
            valid = true;
        } else {
            valid = (unsigned)y < (unsigned)Hs && (unsigned)x < (unsigned)Ws;
            y = min(max(y, 0), Hs - 1);
            x = min(max(x, 0), Ws - 1);
        }
        if constexpr (!UP) { if (p.upsample) { y >>= 1; x >>= 1; } }
        return y * p.W + x;
    };
    u32x4 rh[HL][HG];
    uint32_t hvalid = 0;                               // validity bits of the halo waiting in rh
    int rh_part = 0;                                   // SPL: which half (0, 1: hi; 2: lo) the halo waiting in rh becomes
    auto halo_gload = [&](int slab, int ty0, int tx0) {
        if constexpr (SPL) { rh_part = slab % 3; slab /= 3; }
        int tl = tid;
        asm volatile("" : "+v"(tl));
        const int c = slab * SLABC + (tl & 7) * (SLABC / 8);      // a thread's 8 (16) channels come from ONE source
        const bool interior = !F8 && ty0 >= 1 && ty0 + TH < Hs && tx0 >= 1 && tx0 + TW < Ws;
        hvalid = 0;
        if constexpr (F8) {
            const bf16 *src;
            int ld, cc;
            if (c < p.c0) { src = in0; ld = p.ld0; cc = c; }
            else { src = in1; ld = p.ld1; cc = c - p.c0; }
            HaloPos hq = halo_first(tl);
#pragma unroll
            for (int i = 0; i < HL; ++i, halo_next(hq)) {
                bool v;
                const int ho = halo_geom(tl, i, hq, ty0, tx0, interior, v);
                if (v) hvalid |= 1u << i;
#pragma unroll
                for (int h = 0; h < HG; ++h) {
                    const bf16 *ap = src + (int64_t)ho * ld + cc + 8 * h;
                    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(rh[i][h]) : "v"(ap));      // (see below: hidden from the compiler's waits)
                }
            }
        } else {
            // a 64-channel slab lies in ONE source (c0 % 64 == 0): wave-uniform base + a 32-bit byte offset per load
            const bool first = slab * SLABC < p.c0;
            const char *src = (const char *)(first ? in0 : in1);
            const unsigned ld2 = 2u * ISZ * (unsigned)(first ? p.ld0 : p.ld1);
            const unsigned cc2 = 2u * ISZ * (unsigned)(first ? c : c - p.c0);
            HaloPos hq = halo_first(tl);
#pragma unroll
            for (int i = 0; i < HL; ++i, halo_next(hq)) {
                bool v;
                const unsigned h = (unsigned)halo_geom(tl, i, hq, ty0, tx0, interior, v);
                if (v) hvalid |= 1u << i;
                // The loads are issued through inline asm so that the compiler's s_waitcnt pass does not know them: it counted
                // only ITS outstanding operations, i.e. its wait in front of halo_lstore's first ds_write was vmcnt(HL - 1) ..
                // vmcnt(0) -- a drain of every weight DMA in flight (the tiles of the next unit's taps 1 and 2, requested a tap
                // or two earlier) at EVERY unit boundary; the ring's two-tap lead ended there.  The counted waits of the tap loop
                // already guarantee more than is needed: the last tap's barrier allows NB operations in flight, all of them
                // younger than these loads (issued at tap 0).
                // (not where the kernel spills -- the 128-channel split form, 20 bytes --: a register parked in scratch before its
                //  load has landed would park garbage)
                const unsigned go = h * ld2 + cc2;
                if constexpr (SPL && BN == 128) {
                    rh[i][0] = *(const u32x4 *)(src + go);
                    rh[i][1] = *(const u32x4 *)(src + (go + 16));
                } else {
                    asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(rh[i][0]) : "v"(go), "s"(src));
                    if constexpr (SPL) asm volatile("global_load_dwordx4 %0, %1, %2 offset:16" : "=v"(rh[i][1]) : "v"(go), "s"(src));       // 8 fp32 channels = 32 bytes
                }
            }
        }
    };
    // two bf16 (one dword) -> two e4m3 bytes in the low / high half of `acc`
    auto cvt2 = [&](uint32_t w, uint32_t acc, bool hi) -> uint32_t {
        // e4m3fn has no infinity: clamp to its largest finite value (448) instead of letting an outlier become NaN
        const float a = __builtin_amdgcn_fmed3f(fd_h_lo(w) * p.act_scale, -448.f, 448.f);
        const float b = __builtin_amdgcn_fmed3f(fd_h_hi(w) * p.act_scale, -448.f, 448.f);
        return hi ? (uint32_t)__builtin_amdgcn_cvt_pk_fp8_f32(a, b, (int)acc, true)
                  : (uint32_t)__builtin_amdgcn_cvt_pk_fp8_f32(a, b, (int)acc, false);
    };
    auto halo_lstore = [&]() {
        unsigned char *sH = smem;
        int tl = tid;
        asm volatile("" : "+v"(tl));
        // the halo registers are written by inline-asm loads the compiler's waitcnt pass does not see; every call of
        // this function sits behind a counted vmcnt wait (+ barrier) that covers them.  Tying each register to an (empty)
        // volatile asm HERE makes every consumer below depend on a statement that stays behind those waits -- a register-only
        // consumer (the hvalid select, the split / fp8 conversions) can no longer be scheduled above them.  What this cannot
        // forbid is a COPY of such a register before the wait (live-range split, AGPR park): tests/test_host_cpu.py pins
        // AGPRs = 0 and scratch = 0 for every instantiation that uses the asm loads.
#pragma unroll
        for (int i = 0; i < HL; ++i)
#pragma unroll
            for (int h = 0; h < HG; ++h) asm volatile("" : "+v"(rh[i][h]));
#pragma unroll
        for (int i = 0; i < HL; ++i) {
            const int hid = tl + 256 * i, hp = hid >> 3;
            const u32x4 z4 = {0, 0, 0, 0};
            u32x4 v;
            if constexpr (F8) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const u32x4 s4 = rh[i][h];
                    v[2 * h] = cvt2(s4[1], cvt2(s4[0], 0u, false), true);
                    v[2 * h + 1] = cvt2(s4[3], cvt2(s4[2], 0u, false), true);
                }
            } else if constexpr (SPL) {
                // hi = bf16(x) (round to nearest even), lo = bf16(x - hi): the operand halves of the split contraction
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    // (scalar copies: indexing the vector inside this lambda read element 0 four times -- hipcc 7.2, the pitfall
                    //  DESIGN.md section 3 lists under "compiler pitfalls" (4); found here with delta weights: channel k read as 4 (k / 4))
                    const u32x4 s4 = rh[i][h];
                    const uint32_t sw[4] = {s4.x, s4.y, s4.z, s4.w};
                    f32x2 a = {__builtin_bit_cast(float, sw[0]), __builtin_bit_cast(float, sw[1])};
                    f32x2 c2 = {__builtin_bit_cast(float, sw[2]), __builtin_bit_cast(float, sw[3])};
                    uint32_t pa = fd_pack_bf16(a), pc = fd_pack_bf16(c2);
                    if (rh_part == 2) {
                        pa = fd_pack_bf16(a - fd_unpack_bf16(pa));
                        pc = fd_pack_bf16(c2 - fd_unpack_bf16(pc));
                    }
                    v[2 * h] = pa;
                    v[2 * h + 1] = pc;
                }
            } else v = rh[i][0];
            if (hp < HP) *(u32x4 *)(sH + hp * ROWB + swz<F8>(hp, tl & 7)) = ((hvalid >> i) & 1) ? v : z4;
        }
    };
    // ---- weight tiles ([BN rows][64 k] of tap t, slab s) by LDS-DMA (global_load_lds_dwordx4): no VGPR
    // staging, no ds_write, and a ring of NWB = 3 tiles so that a tile is requested TWO taps before it is
    // read.  (With the register-staged double buffer the request preceded the use by one tap, ~800 cycles:
    // an L2 round trip under load is longer -- removing the per-tap weight traffic altogether made these
    // convolutions 30 % faster, which is the stall this ring goes after.)
    // A wave-instruction fills 1 KiB = 8 tile rows: lane -> (row 8m + lane/8, physical chunk lane%8); the XOR
    // swizzle of the LDS image is applied on the SOURCE side (the lane fetches the logical chunk that
    // belongs in its physical slot).  Rows beyond Cout read row Cout-1: columns the epilogue never stores.
    unsigned goff[NB];      // per-lane BYTE offset inside the weight matrix, computed once
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const int r = 8 * (wave * NB + i) + (lane >> 3);
        const int c = swz<F8>(r, lane & 7) >> 4;          // the logical chunk that belongs in this lane's physical slot
        const int n = min(nt * BN + tile_row_channel(r), p.Cout - 1);
        goff[i] = (unsigned)(n * K * ESZ + c * 16);
    }
    // Issued through inline asm on purpose: for __builtin_amdgcn_global_load_lds on a plain LDS array the
    // compiler's waitcnt pass (no alias scopes to tell the ring slots apart) inserts s_waitcnt vmcnt(0)
    // before the next ds_read of ANY LDS address, i.e. right after the request -- the opposite of a
    // prefetch.  The counted waits below are therefore manual.  (The compiler's own vmcnt accounting for the
    // register halo loads and the epilogue's stores stays safe: operations it does not know about only make its waits
    // stricter.)  M0 (the DMA's LDS base) is written inside the asm -- hipcc rejects "m0" in clobber lists as a
    // reserved register, so each request saves and restores it.  The counted waits assume exactly NB DMA
    // instructions per wave and tap and HL * HG register halo loads per wave and slab; loads retire in order, so a
    // store (or any other VMEM operation) still in flight only makes a counted wait longer, never shorter.
    const unsigned lds_w = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void *)smem + HALO_B +
                           __builtin_amdgcn_readfirstlane(wave) * NB * 1024;
    // weight-ring slot (byte offset) of tap ct of the current unit, ct continuing into the next unit (ct >= NTAP): 9 taps
    // start every unit at slot 0; with 4 taps per unit (UP) the offsets rotate by one slot per unit (rotated at the unit's end)
    int so[3] = {0, WT_B, 2 * WT_B};
    auto slot = [&](int ct) -> int {
        if constexpr (UP) return so[ct % 3];
        else return (ct % NWB) * WT_B;
    };
    auto w_dma = [&](int cls, int slab, int tap, int bufoff) {
        const unsigned char *wm = wgt;
        if constexpr (SPL) { wm = (slab % 3 == 1) ? wgt_lo : wgt; slab /= 3; }
        const char *wb = (const char *)(wm + (((UP ? 4 * cls : 0) + tap) * Cin + slab * SLABC) * ESZ);
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const unsigned dst = lds_w + bufoff + i * 1024;           // wave-uniform: M0
            unsigned m0_saved;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3\n\t"
                         "s_mov_b32 m0, %0"
                         : "=&s"(m0_saved) : "s"(dst), "v"(goff[i]), "s"(wb) : "memory");
        }
    };
    // s_waitcnt vmcnt(n) alone (expcnt / lgkmcnt left at their maxima); gfx9 encoding
#define FD_WAIT_VM(n) __builtin_amdgcn_s_waitcnt(((n) & 0xF) | (((n) >> 4) << 14) | (0x7 << 4) | (0xF << 8))
    // ... and lgkmcnt(l): LDS operations return in order (no scalar loads are in flight inside the tap loop)
#define FD_WAIT_VM_LGKM(n, l) __builtin_amdgcn_s_waitcnt(((n) & 0xF) | (((n) >> 4) << 14) | (0x7 << 4) | ((l) << 8))

    const int wm = wave / WNW, wn = wave % WNW;
    const int fr = lane & 15, fg = lane >> 4;
    f32x4 acc[MT][NT];

    // LDS byte offsets of the A / B fragments, computed ONCE: with the 9 taps unrolled a tap is the
    // compile-time pick aoff[i + kh][kw] (6 halo rows x 3 column shifts cover all 36 (tap, m-tile) pairs)
    // and the second K32 step of a 64-channel slab flips bit 6 of the swizzled chunk (chunk ^ 4).  Before,
    // the per-tap swizzle arithmetic cost 3.6 VALU instructions per MFMA (PMC) -- more than the 8 issue
    // cycles a 16x16x32 MFMA leaves free.
    int aoff[UP ? 5 : 6][UP ? 2 : 3], boff[NT];
    // UP: the four taps (r, c) of parity class (a, b) are the taps (a + r, b + c) of the 3x3 offset table: 5 x 2 offsets per class
    auto set_class = [&](int cls) {
        int tl = tid;
        asm volatile("" : "+v"(tl));                  // (recomputed per class: not a tile-loop invariant worth 10 live registers)
        const int a = cls >> 1, bq_ = cls & 1, fr_ = tl & 15, fg_ = (tl >> 4) & 3, wm_ = (tl >> 6) / WNW;
#pragma unroll
        for (int j = 0; j < (UP ? 5 : 6); ++j)
#pragma unroll
            for (int kw = 0; kw < (UP ? 2 : 3); ++kw) {
                const int hp = (4 * wm_ + j + a) * HX + fr_ + kw + bq_;
                aoff[j][kw] = hp * ROWB + swz<F8>(hp, F8 ? 2 * fg_ : fg_);  // F8: chunks 2 fg, 2 fg + 1 (= offset ^ 16)
            }
    };
    if constexpr (UP) set_class(cls_begin);
    else {
#pragma unroll
    for (int j = 0; j < 6; ++j)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
            const int hp = (4 * wm + j) * HX + fr + kw;
            aoff[j][kw] = hp * ROWB + swz<F8>(hp, F8 ? 2 * fg : fg);        // F8: chunks 2 fg, 2 fg + 1 (= offset ^ 16)
        }
    }
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int r = (BN / WNW) * wn + 16 * j + fr;
        boff[j] = r * ROWB + swz<F8>(r, F8 ? 2 * fg : fg);
    }

    // bf16 register sets: weight fragments double-buffered, pixel fragments rolling (fa[i] is reloaded for the next
    // half-step as soon as the NT MFMAs that read it are issued).  The second K32 step of a slab flips bit 6 of the
    // swizzled offsets: one v_xor per read, kept opaque so that the compiler does not keep a second copy of the
    // 18 + NT offsets in registers.
    bf16x8 fa[MT], fb[2][NT];
    auto koff = [&](int off, int ks) -> int {
        if (ks == 0) return off;
        int o;
        asm volatile("v_xor_b32 %0, 64, %1" : "=v"(o) : "v"(off));
        return o;
    };
    auto load_a = [&](int tap, int ks, int i) {
        const int kh = UP ? tap >> 1 : tap / 3, kw = UP ? tap & 1 : tap - kh * 3;
        fa[i] = *(const bf16x8 *)(smem + koff(aoff[i + kh][kw], ks));              // pixel tile i = tile row 4 wm + i
    };
    auto load_b = [&](int ct, int ks, bf16x8 (&bq)[NT]) {                          // ct: slot() above
        const unsigned char *sB = smem + HALO_B + slot(ct);
#pragma unroll
        for (int j = 0; j < NT; ++j) bq[j] = *(const bf16x8 *)(sB + koff(boff[j], ks));
    };
    // the MFMAs of one half-step; `next`: request the pixel fragments of half-step (ntap, nks) behind their last readers
    auto half_step = [&](const bf16x8 (&bq)[NT], bool next, int ntap, int nks) {
#pragma unroll
        for (int i = 0; i < MT; ++i) {
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[i][j] = FD_MFMA16(bq[j], fa[i], acc[i][j], 0, 0, 0);
            if (next) load_a(ntap, nks, i);
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // bias / fp8 weight scales of the channel tile: read from LDS by every tile's epilogue (a global load there would
    // sit behind the weight DMA and halo loads in flight: vmcnt retires in order)
    if (tid < BN) {
        const int n = blockIdx.y * BN + tid;
        s_bias[tid] = (p.bias && n < p.Cout) ? p.bias[n] : 0.f;
        s_wsc[tid] = (F8 && n < p.Cout) ? p.w_scale[n] / p.act_scale : 1.f;
    }
    int ty0 = (t_begin / tiles_x) * TH, tx0 = (t_begin % tiles_x) * TW;
    w_dma(cls_begin, 0, 0, 0);
    w_dma(cls_begin, 0, 1, WT_B);
    if constexpr (!F8) w_dma(cls_begin, 0, 2, 2 * WT_B);
    halo_gload(0, ty0, tx0);
    FD_WAIT_VM(0);                                   // (the halo loads are asm: no compiler wait in front of their first use)
    halo_lstore();
    FD_WAIT_VM(0);
    __syncthreads();
    if constexpr (!F8) {
#pragma unroll
        for (int i = 0; i < MT; ++i) load_a(0, 0, i);
        load_b(0, 0, fb[0]);
    }

    for (int t = t_begin; t < t_end; ++t) {
        const bool more_tiles_t = t + 1 < t_end;
        const int tn = more_tiles_t ? t + 1 : t;
        const int nty0_t = (tn / tiles_x) * TH, ntx0_t = (tn % tiles_x) * TW;
      for (int cls = cls_begin; cls < cls_end; ++cls) {  // UP: the workgroup's output parity classes of the (low-resolution) tile
        // what follows this (tile, class): the tile's next class (same halo), or the workgroup's next tile
        const bool more_cls = UP && cls + 1 < cls_end;
        const bool more_tiles = more_cls || more_tiles_t;                     // "another (tile, class) follows"
        const int nty0 = more_cls ? ty0 : nty0_t, ntx0 = more_cls ? tx0 : ntx0_t;
        const int ncls = more_cls ? cls + 1 : 0;
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

        for (int slab = 0; slab < nslab; ++slab) {
            // the (tile, slab) unit that follows this one: the next slab, or slab 0 of the workgroup's next tile -- its
            // halo is loaded into registers during this unit's taps, its first weight tiles ride the ring
            const bool last_slab = slab + 1 == nslab;
            const bool has_next = !last_slab || more_tiles;
            const int nslab_i = last_slab ? 0 : slab + 1;
            const int ncls_i = last_slab ? ncls : cls;
            // SPL: the unit behind x_hi.w_hi (part 0) is x_hi.w_lo on the SAME halo image: nothing to load or store for it
            const bool reload = !(SPL && !last_slab && slab % 3 == 0);
            if constexpr (F8) {
                const unsigned char *sH = smem;
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) {
                    // request tile q+2 into the buffer tile q-1 was read from (all waves are past that barrier)
                    const bool dma = tap + 2 < 9 || has_next;
                    if (tap == 0 && has_next) {                   // in flight during this slab's 9 taps
                        if (last_slab) halo_gload(0, nty0, ntx0); else halo_gload(slab + 1, ty0, tx0);
                    }
                    if (dma) w_dma(0, tap + 2 < 9 ? slab : nslab_i, (tap + 2) % 9, ((tap + 2) % NWB) * WT_B);
                    const unsigned char *sB = smem + HALO_B + (tap % NWB) * WT_B;
                    const int kh = tap / 3, kw = tap - kh * 3;
                    // operands are 8 VGPRs each: the pixel fragments of the wave's 4 m-tiles stay live (32 VGPRs), the
                    // weight fragments are read one n-tile at a time (a compiler fence keeps hipcc from hoisting all of
                    // them above the MFMAs, which spilled ~600 registers); 4 MFMAs x 32 cycles cover the next read
                    i32x8 af[MT];
#pragma unroll
                    for (int i = 0; i < MT; ++i) {
                        const u32x4 lo = *(const u32x4 *)(sH + aoff[i + kh][kw]), hi = *(const u32x4 *)(sH + (aoff[i + kh][kw] ^ 16));
                        af[i] = (i32x8){(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
                    }
#pragma unroll
                    for (int j = 0; j < NT; ++j) {
                        const u32x4 lo = *(const u32x4 *)(sB + boff[j]), hi = *(const u32x4 *)(sB + (boff[j] ^ 16));
                        const i32x8 bj = (i32x8){(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
#pragma unroll
                        for (int i = 0; i < MT; ++i)
                            // inline asm with the accumulator TIED (dst = src C): through the builtin hipcc 7.2 leaves the
                            // two untied and the register allocator spills ~600 VGPRs.  Hazards by hand: A / B come from
                            // ds_read (the compiler's s_waitcnt covers asm operands), the same accumulator recurs only
                            // 4 MFMAs (>= 128 cycles) later, s_nop 1 covers the v_mov of the scale register, and the
                            // epilogue's first read of the accumulators sits behind the s_nop block after the loop.
                            asm("s_nop 1\n\tv_mfma_scale_f32_16x16x128_f8f6f4 %0, %2, %1, %0, %3, %3 op_sel_hi:[0,0,0]"
                                : "+v"(acc[i][j]) : "v"(af[i]), "v"(bj), "v"(0x7f7f7f7f));
                    }
                    // tile q+1 (requested one tap ago) must have landed before the barrier publishes it.  vmcnt retires
                    // in order: allow exactly the operations issued AFTER that request -- this tap's DMA (NB
                    // instructions) and, during the first tap of a slab, the halo loads of the next unit.
                    const bool halo_young = tap < 1 && has_next;
                    if (dma) { if (halo_young) FD_WAIT_VM(NB + HL * HG); else FD_WAIT_VM(NB); }
                    else FD_WAIT_VM(0);
                    __syncthreads();
                    if (tap == 8 && !last_slab) {           // every wave is done with this slab's halo
                        halo_lstore();
                        __syncthreads();
                    }
                }
            } else {
                // bf16: a tap is two K32 half-steps; the fragments of half-step h + 1 are requested from LDS before /
                // between the 16 MFMAs of half-step h, so an LDS round trip is behind matrix work instead of in front of
                // it.  The per-tap barrier sits BETWEEN a tap's two half-steps: by then every wave has read both halves
                // of weight tile t (the (t, 1) reads were issued before the (t, 0) MFMAs), so B_t both publishes tile
                // t + 1 -- whose first fragments are requested right behind it -- and frees slot t % 3 for the request of
                // tile t + 3: three slots give a two-tap lead.
                //   per tap t:  B reads(t, 1) | MFMA(t, 0) + A reads(t, 1) | wait tile t+1, B_t | DMA tile t+3 |
                //               B reads(t+1, 0) | MFMA(t, 1) + A reads(t+1, 0)
                // vmcnt order of a unit: ... DMA(2) | B_0 | DMA(3), next unit's halo loads | B_1 | DMA(4) | B_2 | DMA(5) ...
#pragma unroll
                for (int tap = 0; tap < NTAP; ++tap) {
                    const bool next_tile = tap + 1 < NTAP || has_next;
                    load_b(tap, 1, fb[1]);
                    __builtin_amdgcn_sched_barrier(0);
                    half_step(fb[0], true, tap, 1);
                    if (next_tile) {
                        // tile t+1 landed: younger than its request are DMA(t+2) and, at B_1 / B_2, the next unit's halo
                        // loads; and this wave's reads of tile t have returned: all but the MT youngest LDS reads (the
                        // pixel fragments of (t, 1), from the halo).  A bare s_barrier: __syncthreads() would drain those
                        // as well, an LDS round trip in front of every barrier.
                        const bool t2 = tap + 2 < NTAP || has_next;
                        const bool halo_young = has_next && reload && (tap == 1 || tap == 2);
                        asm volatile("" ::: "memory");
                        // (the unit's last barrier drains them too: behind it the halo buffer is overwritten -- next slab / C staging)
                        if (tap == NTAP - 1) { if (t2) FD_WAIT_VM_LGKM(NB, 0); else FD_WAIT_VM_LGKM(0, 0); }
                        else if (t2) { if (halo_young) FD_WAIT_VM_LGKM(NB + HL * HG, MT); else FD_WAIT_VM_LGKM(NB, MT); }
                        else FD_WAIT_VM_LGKM(0, MT);
                        __builtin_amdgcn_s_barrier();
                        asm volatile("" ::: "memory");
                        if (tap + 3 < NTAP || has_next)
                            w_dma(tap + 3 < NTAP ? cls : ncls_i, tap + 3 < NTAP ? slab : nslab_i, (tap + 3) % NTAP, slot(tap + 3));
                        if (tap == 0 && has_next && reload) {     // in flight during this unit's taps
                            if (last_slab) halo_gload(0, nty0, ntx0); else halo_gload(slab + 1, ty0, tx0);
                        }
                    }
                    if (tap < NTAP - 1) {
                        load_b(tap + 1, 0, fb[0]);
                        __builtin_amdgcn_sched_barrier(0);
                        half_step(fb[1], true, tap + 1, 0);
                    } else if (!last_slab) {                      // every wave is done with this slab's halo
                        if (reload) {
                            halo_lstore();
                            __syncthreads();
                        }
                        load_b(NTAP, 0, fb[0]);                   // tap 0 of the next unit
                        __builtin_amdgcn_sched_barrier(0);
                        half_step(fb[1], true, 0, 0);
                    } else half_step(fb[1], false, 0, 0);         // the tile's last MFMAs; the epilogue follows
                }
                if constexpr (UP) {                               // 4 taps per unit: the ring's phase moves on by one slot
                    const int s0 = so[0];
                    so[0] = so[1];
                    so[1] = so[2];
                    so[2] = s0;
                }
            }
        }
        // F8: the last inline-asm MFMAs must have written their accumulators before the epilogue reads them (the
        // compiler's hazard recognizer does not see into asm); volatile + memory clobber keeps the LDS stores below it
        // (every accumulator is an in/out operand of the nop block, so no register-only consumer can be scheduled above it: fd_common.h)
        if constexpr (F8) {
            static_assert(MT == 4 && (NT == 4 || NT == 2), "accumulator ties below");
            if constexpr (NT == 4) asm volatile(FD_MFMA_ASM_DRAIN : FD_TIE4(acc[0]), FD_TIE4(acc[1]), FD_TIE4(acc[2]), FD_TIE4(acc[3]) :: "memory");
            else asm volatile(FD_MFMA_ASM_DRAIN : FD_TIE2(acc[0]), FD_TIE2(acc[1]), FD_TIE2(acc[2]), FD_TIE2(acc[3]) :: "memory");
        }

        // ---- epilogue.  acc[i][2 q + h][e] of lane (fr, fg) = pixel (tile row 4 wm + i, column fr), output channel
        // cb + 32 q + 4 h + e with cb = the wave's channel base + 8 fg (tile_row_channel above).  Every wave is past its
        // last halo read (B_8), so the halo buffer stages the bf16 C tile 64 channels at a time -- 16-byte LDS
        // stores, then every store instruction of a wave writes 8 whole 128-byte rows -- while the next tile's halo
        // waits in rh and its first weight tiles land in the ring (storing 64-byte pieces straight from the
        // accumulators measured 8 % slower than this round trip).
        // (lane geometry from an opaque copy of tid: as tile-loop invariants the bias values, staging addresses and
        // output offsets of the epilogue would stay live -- spilled -- across the tap loop)
        constexpr int NQ = NT / 2;
        if constexpr (SPL) {
            // ---- fp32 epilogue of the split form: bias, GroupNorm partial sums, 32-byte stores straight from the accumulators
            int te = tid;
            asm volatile("" : "+v"(te));
            const int fr = te & 15, fg = (te >> 4) & 3, wv = te >> 6;
            const int wm = wv / WNW, wn = wv % WNW;
            const int cb = (BN / WNW) * wn + 8 * fg;
            float *outp = (float *)p.out + (int64_t)b * p.OH * p.OW * p.ldo + p.offo;
            if (!more_tiles) __syncthreads();    // (s_stat below; every wave is past its last LDS read of the unit)
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const int n0 = nt * BN + cb + 32 * q;
                float bias[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) bias[e] = s_bias[cb + 32 * q + e];
                f32x2 su[4] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}}, sq[4] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
#pragma unroll
                for (int i = 0; i < MT; ++i) {
                    float val[8];
#pragma unroll
                    for (int e = 0; e < 8; e += 2) {
                        const f32x4 a4 = acc[i][2 * q + (e >> 2)];
                        const f32x2 a = (e & 2) ? f32x2{a4[2], a4[3]} : f32x2{a4[0], a4[1]};
                        const f32x2 v2 = a + f32x2{bias[e], bias[e + 1]};
                        su[e >> 1] += v2;
                        sq[e >> 1] = v2 * v2 + sq[e >> 1];
                        val[e] = v2.x;
                        val[e + 1] = v2.y;
                    }
                    if (n0 < p.Cout) {
                        // (UP: tile pixel (i, j) of class (a, b) = (cls >> 1, cls & 1) is output pixel (2 i + a, 2 j + b))
                        float *op = UP ? outp + ((int64_t)(2 * (ty0 + 4 * wm + i) + (cls >> 1)) * p.OW + 2 * (tx0 + fr) + (cls & 1)) * p.ldo + n0
                                       : outp + ((int64_t)(ty0 + 4 * wm + i) * p.OW + tx0 + fr) * p.ldo + n0;
                        *(f32x4 *)op = f32x4{val[0], val[1], val[2], val[3]};
                        *(f32x4 *)(op + 4) = f32x4{val[4], val[5], val[6], val[7]};
                    }
                }
                if (p.stats_partial) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float sm = row16_sum(e & 1 ? su[e >> 1].y : su[e >> 1].x), sv = row16_sum(e & 1 ? sq[e >> 1].y : sq[e >> 1].x);
                        if (fr == 0) *(f32x2 *)&s_stat[wm][cb + 32 * q + e][0] = f32x2{sm, sv};
                    }
                }
            }
            if (p.stats_partial) {
                __syncthreads();
                if (te < BN) {
                    const int n = nt * BN + te;
                    if (n < p.Cout) {
                        float sm = 0.f, sq1 = 0.f;
#pragma unroll
                        for (int w = 0; w < WMW; ++w) { sm += s_stat[w][te][0]; sq1 += s_stat[w][te][1]; }
                        constexpr int EPT = BM / 64;
                        float *sp = p.stats_partial + (((int64_t)b * EPT * tiles_xy + EPT * t) * p.Cout + n) * 2;
                        sp[0] = sm;
                        sp[1] = sq1;
#pragma unroll
                        for (int e = 1; e < EPT; ++e) {
                            sp[2 * e * p.Cout] = 0.f;
                            sp[2 * e * p.Cout + 1] = 0.f;
                        }
                    }
                }
            }
            if (more_tiles) {
                __syncthreads();
                halo_lstore();
            }
        } else {
        if (!more_tiles) __syncthreads();        // no B_8 behind the workgroup's last unit: halo reads of other waves
        int te = tid;
        asm volatile("" : "+v"(te));
        const int fr = te & 15, fg = (te >> 4) & 3, wv = te >> 6;
        const int wm = wv / WNW, wn = wv % WNW;
        const int cb = (BN / WNW) * wn + 8 * fg;
        __attribute__((aligned(8))) float ssum[4 * NT], ssq[4 * NT];
#pragma unroll
        for (int k = 0; k < 4 * NT; ++k) ssum[k] = ssq[k] = 0.f;
        const bool relu = p.epilogue == FD_EPI_RELU;
        bf16 *outp = (bf16 *)p.out + (int64_t)b * p.OH * p.OW * p.ldo + p.offo;
        const int v8 = te & 7, r0 = te >> 3;
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps) {
            // (two copies, ReLU or not: as a per-value select it was 128 of the epilogue's ~450 VALU instructions)
            auto stage = [&](auto relu_c) {
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    float bias[8], wsc[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        bias[e] = s_bias[cb + 32 * q + e];
                        wsc[e] = F8 ? s_wsc[cb + 32 * q + e] : 1.f;
                    }
                    // 16-byte chunk of the staged 128-byte row: the wave's channels of this pass, octet by octet
                    const int ch = (NPASS == 1 ? (BN / WNW) * wn / 8 : 0) + fg + 4 * q;
#pragma unroll
                    for (int i = 0; i < MT; ++i) {
                        float val[8];
#pragma unroll
                        for (int e = 0; e < 8; e += 2) {         // channel pairs: v_pk_add / v_pk_fma
                            const f32x4 a4 = acc[i][2 * q + (e >> 2)];
                            const f32x2 a = (e & 2) ? f32x2{a4[2], a4[3]} : f32x2{a4[0], a4[1]};
                            const f32x2 bi = {bias[e], bias[e + 1]};
                            f32x2 v2 = F8 ? a * f32x2{wsc[e], wsc[e + 1]} + bi : a + bi;
                            if (decltype(relu_c)::value == 1 || (decltype(relu_c)::value == 2 && relu))
                                v2 = f32x2{fmaxf(v2.x, 0.f), fmaxf(v2.y, 0.f)};
                            if constexpr (!UP) {             // (the up-sampling convolutions emit no GroupNorm sums: fd_conv3x3_up2x_ok)
                                f32x2 &su = *(f32x2 *)&ssum[8 * q + e], &sq2 = *(f32x2 *)&ssq[8 * q + e];
                                su += v2;
                                sq2 = v2 * v2 + sq2;
                            }
                            val[e] = v2.x;
                            val[e + 1] = v2.y;
                        }
                        const int r = 16 * (4 * wm + i) + fr;
                        store8((bf16 *)(smem + r * 128 + ((ch ^ (r & 7)) << 4)), val);
                    }
                }
            };
            if (NPASS == 1 || wn == ps) {
                if constexpr (F8) stage(std::integral_constant<int, 2>{});      // one copy (register budget), per-value select
                else if (relu) stage(std::integral_constant<int, 1>{});
                else stage(std::integral_constant<int, 0>{});
            }
            if (ps == NPASS - 1 && p.stats_partial) {
                // per-channel sums over the wave's 64 pixels: slot k = 8 q + e of lane (fr, fg) is channel cb + 32 q + e;
                // lane fr = 0 of each row writes its group's 4 NT channels
#pragma unroll
                for (int k = 0; k < 4 * NT; ++k) {
                    ssum[k] = row16_sum(ssum[k]);
                    ssq[k] = row16_sum(ssq[k]);
                }
                if (fr == 0) {
#pragma unroll
                    for (int k = 0; k < 4 * NT; ++k)
                        *(f32x2 *)&s_stat[wm][cb + 32 * (k >> 3) + (k & 7)][0] = f32x2{ssum[k], ssq[k]};
                }
            }
            __syncthreads();
            u32x4 cv[RB];
#pragma unroll
            for (int k = 0; k < RB; ++k) {
                const int r = r0 + 32 * k;
                cv[k] = *(const u32x4 *)(smem + r * 128 + ((v8 ^ (r & 7)) << 4));
            }
            if (ps == NPASS - 1 && p.stats_partial && te < BN) {
                const int n = nt * BN + te;
                if (n < p.Cout) {
                    float sm = 0.f, sq = 0.f;
#pragma unroll
                    for (int w = 0; w < WMW; ++w) { sm += s_stat[w][te][0]; sq += s_stat[w][te][1]; }
                    // the workspace holds one entry per 64 output pixels (fd_conv_mtiles): this BM-pixel
                    // tile fills its first entry and zeroes the other BM/64 - 1
                    constexpr int EPT = BM / 64;
                    float *sp = p.stats_partial + (((int64_t)b * EPT * tiles_xy + EPT * t) * p.Cout + n) * 2;
                    sp[0] = sm;
                    sp[1] = sq;
#pragma unroll
                    for (int e = 1; e < EPT; ++e) {
                        sp[2 * e * p.Cout] = 0.f;
                        sp[2 * e * p.Cout + 1] = 0.f;
                    }
                }
            }
            __syncthreads();                             // the staging area is free again
            if (ps == NPASS - 1 && more_tiles) halo_lstore();      // the next tile's halo, before the stores below
            const int n0 = nt * BN + 64 * ps + 8 * v8;
            if (n0 < p.Cout) {
                // row r0 + 32 k is pixel (ty0 + r0 / 16 + 2 k, tx0 + r0 % 16): one 32-bit byte offset + a uniform step
                // from the image's (wave-uniform) base
                // (UP: tile pixel (i, j) of class (a, b) = (cls >> 1, cls & 1) is output pixel (2 i + a, 2 j + b))
                const unsigned step = (UP ? 8u : 4u) * (unsigned)p.OW * (unsigned)p.ldo;
                unsigned ob = UP ? 2u * ((unsigned)((2 * (ty0 + (r0 >> 4)) + (cls >> 1)) * p.OW + 2 * (tx0 + (r0 & 15)) + (cls & 1)) * (unsigned)p.ldo + (unsigned)n0)
                                 : 2u * ((unsigned)((ty0 + (r0 >> 4)) * p.OW + tx0 + (r0 & 15)) * (unsigned)p.ldo + (unsigned)n0);
#pragma unroll
                for (int k = 0; k < RB; ++k, ob += step) *(u32x4 *)((char *)outp + ob) = cv[k];
            }
        }
        }
        if (more_tiles) {
            __syncthreads();                             // the next tile's halo is published
            ty0 = nty0;
            tx0 = ntx0;
            if constexpr (UP) set_class(ncls);
            if constexpr (!F8) {
#pragma unroll
                for (int i = 0; i < MT; ++i) load_a(0, 0, i);
                load_b(0, 0, fb[0]);
            }
        }
      }
    }
#undef FD_WAIT_VM
#undef FD_WAIT_VM_LGKM
}

}  // namespace

// 1 if `p` runs on the halo-tiled 3x3 kernel.
int fd_conv3x3_ok(const fd_conv_params &p) {
    const int Cin = p.c0 + p.c1;
    if (p.dtype != FD_BF16 || p.out_f32 || p.ndir != 1) return 0;
    if (p.KH != 3 || p.KW != 3 || p.stride != 1 || p.pad_h != 1 || p.pad_w != 1) return 0;
    if (p.epilogue != FD_EPI_NONE && p.epilogue != FD_EPI_RELU) return 0;
    if (p.prologue != FD_PRO_NONE) return 0;
    if (Cin % 64 || p.c0 % 64 || p.Cout % 8) return 0;
    if (p.ld0 % 8 || p.off0 % 8 || (p.in1 && (p.ld1 % 8 || p.off1 % 8)) || p.ldo % 8 || p.offo % 8) return 0;
    if (p.OH % 8 || p.OW % TW) return 0;
    if (p.OH != (p.upsample ? 2 * p.H : p.H) || p.OW != (p.upsample ? 2 * p.W : p.W)) return 0;
    // at <= 16384 pixels per image the generic tiles (64-row / 8-wave 128x256) fill the chip better
    if ((int64_t)p.OH * p.OW < 4096) return 0;
    if ((int64_t)p.H * p.W * (p.ld0 > p.ld1 ? p.ld0 : p.ld1) >= (1ll << 30)) return 0;   // 32-bit byte offsets
    if ((int64_t)p.Cout * 16 * Cin >= (1ll << 30)) return 0;
    if ((int64_t)p.OH * p.OW * p.ldo >= (1ll << 30)) return 0;
    return 1;
}

// 1 if `p` carries fp8 weights and runs on the fp8 form of the halo kernel: K axis in 128-channel slabs, each
// thread's 16 halo channels from one source.
int fd_conv3x3_fp8_ok(const fd_conv_params &p) {
    if (!p.weight_f8 || !p.w_scale || !(p.act_scale > 0.f)) return 0;
    if (!fd_conv3x3_ok(p)) return 0;
    return (p.c0 + p.c1) % 128 == 0 && p.c0 % 16 == 0 && (p.in1 == nullptr || p.c1 % 16 == 0);
}

// 1 if `p` (fp32 storage, f32_split, the pre-split bf16 weight matrices set) runs on the split-bf16 form of the halo kernel
int fd_conv3x3_split_ok(const fd_conv_params &p) {
    const bool off = fd_dev(FD_DEV_NO_CONV3_SPLIT);     // development: the generic split implicit GEMM
    const int Cin = p.c0 + p.c1;
    if (off || p.dtype != FD_F32 || !p.f32_split || !p.weight_split_hi || !p.weight_split_lo || p.ndir != 1) return 0;
    if (p.KH != 3 || p.KW != 3 || p.stride != 1 || p.pad_h != 1 || p.pad_w != 1) return 0;
    if (p.epilogue != FD_EPI_NONE || p.prologue != FD_PRO_NONE) return 0;
    if (Cin % 64 || p.c0 % 64 || p.Cout % 8) return 0;
    if (p.ld0 % 4 || p.off0 % 4 || (p.in1 && (p.ld1 % 4 || p.off1 % 4)) || p.ldo % 4 || p.offo % 4) return 0;
    if (p.OH % 8 || p.OW % TW) return 0;
    if (p.OH != (p.upsample ? 2 * p.H : p.H) || p.OW != (p.upsample ? 2 * p.W : p.W)) return 0;
    if ((int64_t)p.OH * p.OW < 4096) return 0;
    if ((int64_t)p.H * p.W * (p.ld0 > p.ld1 ? p.ld0 : p.ld1) >= (1ll << 29)) return 0;   // 32-bit byte offsets, 4-byte elements
    if ((int64_t)p.Cout * 9 * Cin >= (1ll << 30)) return 0;
    return 1;
}

// 1 if `p` (fp32 storage, f32_split, an up-sampling 3x3 with its pre-split sub-pixel matrices) runs as four 2x2 convolutions on
// the source grid in the split-bf16 form: both of the above at once (the fp32s engine's three Upsample convolutions)
int fd_conv3x3_up2x_split_ok(const fd_conv_params &p) {
    const bool off = (fd_dev(FD_DEV_NO_CONV3_UP2X) | fd_dev(FD_DEV_NO_CONV3_SPLIT)) != 0;
    if (off || !p.weight_up2x_split_hi || !p.weight_up2x_split_lo || !p.upsample || p.stats_partial) return 0;
    if (p.H % 8 || p.W % TW) return 0;
    if ((int64_t)p.Cout * 16 * (p.c0 + p.c1) >= (1ll << 30)) return 0;
    fd_conv_params q = p;                   // every other condition is the 9-tap split form's
    q.weight_split_hi = p.weight_up2x_split_hi;
    q.weight_split_lo = p.weight_up2x_split_lo;
    return fd_conv3x3_split_ok(q);
}

// 1 if `p` (an up-sampling 3x3 with its sub-pixel weight matrix, weight_up2x) runs as four 2x2 convolutions on the source grid
int fd_conv3x3_up2x_ok(const fd_conv_params &p) {
    const bool off = fd_dev(FD_DEV_NO_CONV3_UP2X);    // development: the 9-tap form through the up-sampling index map
    return !off && p.weight_up2x && p.upsample && !p.weight_f8 && !p.stats_partial && p.epilogue == FD_EPI_NONE &&
           p.H % 8 == 0 && p.W % TW == 0 && fd_conv3x3_ok(p);
}

int fd_conv3x3_launch(const fd_conv_params &p, hipStream_t s) {
    const bool wide = p.Cout > 64;
    if (fd_conv3x3_up2x_split_ok(p)) {
        const int cpw_env = fd_dev(FD_DEV_CONV3_UP_CPW);
        const int cpw = (cpw_env == 1 || cpw_env == 2 || cpw_env == 4) ? cpw_env : (p.upsample == 2 ? 1 : 4);
        const int tiles_xy = (p.H / 8) * (p.W / TW), gy = cdiv(p.Cout, wide ? 128 : 64);
        dim3 grid(tiles_xy * (4 / cpw), gy, p.B), block(256);
        if (wide) hipLaunchKernelGGL((conv3x3_halo_kernel<128, 8, false, true, true>), grid, block, 0, s, p, cpw, tiles_xy);
        else hipLaunchKernelGGL((conv3x3_halo_kernel<64, 8, false, true, true>), grid, block, 0, s, p, cpw, tiles_xy);
        return 0;
    }
    if (fd_conv3x3_split_ok(p)) {
        // 8-row tiles at every width: the fp32 halo registers of a 16-row tile spill (152 bytes of scratch; 64 -> 64 at 512^2
        // 639 -> 535 us with <64, 8>)
        const int tiles_xy = (p.OH / 8) * (p.OW / TW), gy = cdiv(p.Cout, wide ? 128 : 64);
        dim3 grid(tiles_xy, gy, p.B), block(256);
        if (wide) hipLaunchKernelGGL((conv3x3_halo_kernel<128, 8, false, false, true>), grid, block, 0, s, p, 1, tiles_xy);
        else hipLaunchKernelGGL((conv3x3_halo_kernel<64, 8, false, false, true>), grid, block, 0, s, p, 1, tiles_xy);
        return 0;
    }
    if (fd_conv3x3_up2x_ok(p)) {
        // tiles over the SOURCE grid; one tile per workgroup = 4 classes x Cin / 64 slabs of 4 taps
        const int th = (!wide && p.H % 16 == 0) ? 16 : 8;              // (8-row tiles at Cout <= 64 measured 221 -> 224 us: no)
        const int tiles_xy = (p.H / th) * (p.W / TW), gy = cdiv(p.Cout, wide ? 128 : 64);
        // classes per workgroup: 4 (the four classes of a tile share its workgroup), or 1 when the caller asks for the
        // class-parallel grid (`upsample` = 2: the one-slice kernel set; any split gives the same bits)
        const int cpw_env = fd_dev(FD_DEV_CONV3_UP_CPW);     // development
        const int cpw = (cpw_env == 1 || cpw_env == 2 || cpw_env == 4) ? cpw_env : (p.upsample == 2 ? 1 : 4);
        dim3 grid(tiles_xy * (4 / cpw), gy, p.B), block(256);
        if (wide) hipLaunchKernelGGL((conv3x3_halo_kernel<128, 8, false, true>), grid, block, 0, s, p, cpw, tiles_xy);
        else if (th == 16) hipLaunchKernelGGL((conv3x3_halo_kernel<64, 16, false, true>), grid, block, 0, s, p, cpw, tiles_xy);
        else hipLaunchKernelGGL((conv3x3_halo_kernel<64, 8, false, true>), grid, block, 0, s, p, cpw, tiles_xy);
        return 0;
    }
    const bool th8 = fd_dev(FD_DEV_CONV3_TH8);        // development: 8-row tiles for Cout <= 64 too
    const int th = (!wide && p.OH % 16 == 0 && !th8) ? 16 : 8;
    const int tiles_xy = (p.OH / th) * (p.OW / TW), gy = cdiv(p.Cout, wide ? 128 : 64);
    // consecutive tiles per workgroup: the next tile's halo is in flight during the current tile's MFMAs (a workgroup
    // per tile has nothing in flight while it computes: the 64 -> 64 convolutions of level 0 ran at 2.1 TB/s of a
    // latency-bound load -> compute -> store sequence).  Chosen from the launch size only: any split gives the same
    // bits (a tile's result does not depend on which workgroup computes it).
    const int tpw_env = fd_dev(FD_DEV_CONV_TPW);
    // Round 4: ONE tile per workgroup by default.  Alone on the chip several tiles per workgroup win (64 -> 64 at 512x512:
    // 265 -> 238 us at 4 tiles), inside the forward they lose: FD_CONV_TPW=1 / 2 / 4 for every layer gave 13.09 / 13.11 /
    // 13.37 ms per batch-8 forward against 13.13 with the launch-size rule (interleaved runs, one box).
    int tpw = tpw_env > 0 ? tpw_env : 1;
    tpw = tpw < 1 ? 1 : (tpw > 8 ? 8 : tpw);
    dim3 grid(cdiv(tiles_xy, tpw), gy, p.B), block(256);
    const size_t pad = fd_occ_pad(FD_DEV_PAD_CONV3);
    if (fd_conv3x3_fp8_ok(p)) {
        if (wide) hipLaunchKernelGGL((conv3x3_halo_kernel<128, 8, true>), grid, block, 0, s, p, tpw, tiles_xy);
        else if (th == 16) hipLaunchKernelGGL((conv3x3_halo_kernel<64, 16, true>), grid, block, 0, s, p, tpw, tiles_xy);
        else hipLaunchKernelGGL((conv3x3_halo_kernel<64, 8, true>), grid, block, 0, s, p, tpw, tiles_xy);
        return 0;
    }
    if (wide) hipLaunchKernelGGL((conv3x3_halo_kernel<128, 8, false>), grid, block, pad, s, p, tpw, tiles_xy);
    else if (th == 16) hipLaunchKernelGGL((conv3x3_halo_kernel<64, 16, false>), grid, block, pad, s, p, tpw, tiles_xy);
    else hipLaunchKernelGGL((conv3x3_halo_kernel<64, 8, false>), grid, block, pad, s, p, tpw, tiles_xy);
    return 0;
}
