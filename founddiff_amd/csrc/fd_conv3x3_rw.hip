// fd_conv3x3_rw.hip -- 3x3 / stride-1 / pad-1 convolution 64 -> 64 channels (bf16) with the WEIGHTS RESIDENT IN REGISTERS
// (round 4): the first convolution of the 64-channel ResnetBlocks at 512x512 / 256x256 (src/DADiff.py:139-154, 213-229).
//
// On these single-slab layers the halo-tiled kernel (fd_conv3x3.hip) is issue-bound, not matrix-bound: per tile a wave
// issues 288 MFMAs (4608 cycles) and ~1000 VALU instructions (4000 cycles) on one port -- halo geometry, the swizzle
// arithmetic of the second K32 step, the bookkeeping of a weight ring that re-streams all nine 8 KB tap tiles from L2 for
// every 256 pixels, one barrier per tap.  With K = 9 x 64 = 576 the whole weight matrix is small: a wave that owns 32
// output channels holds its 32 x 576 slice as 36 MFMA A fragments = 144 VGPRs for the lifetime of a persistent
// workgroup.  Per 8 x 16 pixel tile: the 10 x 18 halo goes to LDS once (the NEXT tile's halo is requested into
// registers before this tile's MFMAs), then 18 K32 steps of {4 fragment reads, 8 MFMAs} with no barrier, no DMA wait and
// no address arithmetic beyond one XOR per second-half read; epilogue straight from the accumulators (8 consecutive
// channels per lane: 16-byte stores) with the deterministic per-tile GroupNorm partial sums of fd_conv2d.
#include "fd_common.h"

namespace {

constexpr int RW_TH = 8, RW_TW = 16, RW_HY = RW_TH + 2, RW_HX = RW_TW + 2, RW_HP = RW_HY * RW_HX;      // 180 halo pixels
constexpr int RW_HL = (RW_HP * 8 + 255) / 256;                                                           // 6 chunks per thread

__device__ __forceinline__ int rw_off(int hp, int chunk) { return hp * 128 + ((chunk ^ (hp & 7)) << 4); }

__device__ __forceinline__ float rw_row16_sum(float x) {
    x = fd_dpp_add<0xB1>(x);
    x = fd_dpp_add<0x4E>(x);
    x = fd_dpp_add<0x141>(x);
    return fd_dpp_add<0x140>(x);
}

__global__ __launch_bounds__(256, 2) void conv3x3_rw_kernel(const fd_conv_params p, const int tpw, const int tiles_xy) {
    __shared__ __attribute__((aligned(16))) unsigned char halo[RW_HP * 128];
    __shared__ float s_stat[2][64][2];
    __shared__ __attribute__((aligned(16))) float s_bias[64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fg = lane >> 4;
    const int wn = wave & 1, wm = wave >> 1;               // channel half (32 channels), tile rows 4 wm .. 4 wm + 3
    const int b = blockIdx.y;
    const int tiles_x = p.OW / RW_TW;
    const bf16 *in0 = (const bf16 *)p.in0 + (int64_t)b * p.H * p.W * p.ld0 + p.off0;
    bf16 *outp = (bf16 *)p.out + (int64_t)b * p.OH * p.OW * p.ldo + p.offo;

    // ---- once per workgroup: the wave's 32 x 576 weight slice.  Rows permuted (fd_gemm_rows.hip) so that the two 16-row
    // fragments a = 0, 1 leave lane group fg with the 8 consecutive channels 32 wn + 8 fg .. + 7 of its pixel
    bf16x8 wf[2][18];
    {
        const bf16 *w = (const bf16 *)p.weight;
        const int row = 32 * wn + 8 * (fr >> 2) + (fr & 3);
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int q = 0; q < 18; ++q) wf[a][q] = *(const bf16x8 *)(w + (int64_t)(row + 4 * a) * 576 + q * 32 + fg * 8);
    }
    if (tid < 64) s_bias[tid] = p.bias ? p.bias[tid] : 0.f;       // read by every tile's epilogue (8 registers fewer across the MFMAs)
    // LDS byte offsets of the pixel fragments: halo row j = 0 .. 5 of this wave (tile row 4 wm + i, tap row kh: j = i + kh),
    // column fr + kw, first K32 step (the second flips bit 6 of the swizzled offset)
    int aoff[6][3];
#pragma unroll
    for (int j = 0; j < 6; ++j)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
            const int hp = (4 * wm + j) * RW_HX + fr + kw;
            aoff[j][kw] = rw_off(hp, fg);
        }

    u32x4 rh[RW_HL];
    uint32_t hvalid = 0;
    auto halo_gload = [&](int ty0, int tx0) {
        int tl = tid;
        asm volatile("" : "+v"(tl));                        // per-chunk geometry recomputed, not kept live across the MFMAs
        hvalid = 0;
#pragma unroll
        for (int i = 0; i < RW_HL; ++i) {
            const int id = min(tl + 256 * i, RW_HP * 8 - 1);
            const int hp = id >> 3;
            const int hy = (int)(__umul24((unsigned)hp, 3641u) >> 16);       // hp / 18 for hp < 512 (18 * 3641 = 65538)
            const int hx = hp - hy * RW_HX;
            const int y = ty0 + hy - 1, x = tx0 + hx - 1;
            if ((unsigned)y < (unsigned)p.OH && (unsigned)x < (unsigned)p.OW) hvalid |= 1u << i;
            const int yc = min(max(y, 0), p.OH - 1), xc = min(max(x, 0), p.OW - 1);
            rh[i] = *(const u32x4 *)(in0 + ((__umul24(yc, p.W) + xc) * p.ld0 + (id & 7) * 8));
        }
    };
    auto halo_lstore = [&]() {
        int tl = tid;
        asm volatile("" : "+v"(tl));
#pragma unroll
        for (int i = 0; i < RW_HL; ++i) {
            const int id = tl + 256 * i;
            const u32x4 z4 = {0, 0, 0, 0};
            if (id < RW_HP * 8) *(u32x4 *)(halo + rw_off(id >> 3, id & 7)) = ((hvalid >> i) & 1) ? rh[i] : z4;
        }
    };

    const int t_begin = blockIdx.x * tpw, t_end = min(t_begin + tpw, tiles_xy);
    if (t_begin >= t_end) return;
    halo_gload((t_begin / tiles_x) * RW_TH, (t_begin % tiles_x) * RW_TW);
    halo_lstore();
    __syncthreads();
    for (int t = t_begin; t < t_end; ++t) {
        const int ty0 = (t / tiles_x) * RW_TH, tx0 = (t % tiles_x) * RW_TW;
        const bool more = t + 1 < t_end;
        if (more) halo_gload(((t + 1) / tiles_x) * RW_TH, ((t + 1) % tiles_x) * RW_TW);      // in flight during the MFMAs
        f32x4 acc[2][4];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[a][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    bf16x8 bq[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        int o = aoff[i + kh][kw];
                        if (ks) asm volatile("v_xor_b32 %0, 64, %1" : "=v"(o) : "v"(aoff[i + kh][kw]));
                        bq[i] = *(const bf16x8 *)(halo + o);
                    }
                    // accumulators TIED through inline asm (the builtin leaves destination and C operand apart: spills)
#pragma unroll
                    for (int a = 0; a < 2; ++a)
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                            asm(FD_MFMA16_ASM " %0, %1, %2, %0" : "+v"(acc[a][i]) : "v"(wf[a][(kh * 3 + kw) * 2 + ks]), "v"(bq[i]));
                    __builtin_amdgcn_sched_barrier(0);
                }
        asm volatile(FD_MFMA_ASM_DRAIN : FD_TIE4(acc[0]), FD_TIE4(acc[1]) :: "memory");      // the last MFMAs' results (fd_common.h)

        // ---- epilogue: lane (fr, fg) holds channels n0 .. n0 + 7 (a = 0: + 0..3, a = 1: + 4..7) of pixel (ty0 + 4 wm + i, tx0 + fr)
        const int n0 = 32 * wn + 8 * fg;
        float bias[8];
        load8(s_bias + n0, bias);
        f32x2 ssum[4] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}}, ssq[4] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            uint32_t o[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f32x4 a4 = acc[j >> 1][i];
                const f32x2 v2 = ((j & 1) ? f32x2{a4[2], a4[3]} : f32x2{a4[0], a4[1]}) + f32x2{bias[2 * j], bias[2 * j + 1]};
                o[j] = fd_pack_bf16(v2);
                ssum[j] += v2;                               // (fp32 values, as in fd_conv3x3.hip)
                ssq[j] = v2 * v2 + ssq[j];
            }
            *(u32x4 *)(outp + ((__umul24(ty0 + 4 * wm + i, p.OW) + tx0 + fr) * p.ldo + n0)) = (u32x4){o[0], o[1], o[2], o[3]};
        }
        if (p.stats_partial) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                ssum[j].x = rw_row16_sum(ssum[j].x); ssum[j].y = rw_row16_sum(ssum[j].y);
                ssq[j].x = rw_row16_sum(ssq[j].x); ssq[j].y = rw_row16_sum(ssq[j].y);
            }
            if (fr == 0) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    *(f32x2 *)&s_stat[wm][n0 + 2 * j][0] = f32x2{ssum[j].x, ssq[j].x};
                    *(f32x2 *)&s_stat[wm][n0 + 2 * j + 1][0] = f32x2{ssum[j].y, ssq[j].y};
                }
            }
        }
        __syncthreads();                                    // every wave is done with the halo; s_stat is complete
        if (p.stats_partial && tid < 64) {
            // the workspace holds one entry per 64 output pixels (fd_conv_mtiles): this 128-pixel tile fills its first and
            // zeroes its second
            float *sp = p.stats_partial + (((int64_t)b * 2 * tiles_xy + 2 * t) * 64 + tid) * 2;
            sp[0] = s_stat[0][tid][0] + s_stat[1][tid][0];
            sp[1] = s_stat[0][tid][1] + s_stat[1][tid][1];
            sp[2 * 64] = 0.f;
            sp[2 * 64 + 1] = 0.f;
        }
        if (more) {
            halo_lstore();
            __syncthreads();
        }
    }
}


}  // namespace

int fd_conv3x3_rw_ok(const fd_conv_params &p) {
    const bool off = fd_dev(FD_DEV_NO_CONV3_RW);       // development switch
    if (off || p.dtype != FD_BF16 || p.out_f32 || p.ndir != 1 || p.upsample || p.weight_f8) return 0;
    if (p.KH != 3 || p.KW != 3 || p.stride != 1 || p.pad_h != 1 || p.pad_w != 1) return 0;
    if (p.epilogue != FD_EPI_NONE || p.prologue != FD_PRO_NONE) return 0;
    if (p.c0 != 64 || p.c1 != 0 || p.Cout != 64) return 0;
    if (p.ld0 % 8 || p.off0 % 8 || p.ldo % 8 || p.offo % 8 || p.w_batch_stride) return 0;
    if (p.OH != p.H || p.OW != p.W || p.OH % RW_TH || p.OW % RW_TW) return 0;
    if ((int64_t)p.OH * p.OW < 131072) return 0;            // fewer tiles: the weight load of a workgroup is not amortised (256x256: 52.5 vs 51 us)
    if ((int64_t)p.H * p.W * (p.ld0 > p.ldo ? p.ld0 : p.ldo) >= (1ll << 31)) return 0;      // 32-bit element offsets
    return 1;
}

int fd_conv3x3_rw_launch(const fd_conv_params &p, hipStream_t s) {
    const int tiles_xy = (p.OH / RW_TH) * (p.OW / RW_TW);
    // persistent: one wave of resident workgroups (2 per CU), consecutive tiles each.  The partition does not touch the
    // results (every tile is computed by itself, one GroupNorm partial per tile)
    const int tpw_env = fd_dev(FD_DEV_CONV3_RW_TPW);      // development
    int tpw = tpw_env > 0 ? tpw_env : (int)(((int64_t)tiles_xy * p.B + 511) / 512);
    if (tpw < 1) tpw = 1;
    dim3 grid((tiles_xy + tpw - 1) / tpw, p.B);
    // (a 16-channel-per-wave form -- 72 VGPRs of weights, three workgroups per CU, fragment reads 1:1 with the MFMAs -- measured
    //  246 us against 218: the LDS reads, not the tiles in flight, then set the pace)
    hipLaunchKernelGGL(conv3x3_rw_kernel, grid, dim3(256), 0, s, p, tpw, tiles_xy);
    return 0;
}
