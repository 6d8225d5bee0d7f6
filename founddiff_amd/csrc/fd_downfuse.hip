// fd_downfuse.hip -- the tail of an identity-residual ResnetBlock and the down-sampling convolution behind it in ONE
// pass (round 4): skip = x + SiLU(GroupNorm(h)) (src/DADiff.py:213-229, 418-430) and
// down = Conv2d(C, Cout, 4, stride 2, padding 1)(skip) (src/DADiff.py:128-131, 578-584), bf16, C = 64.
//
// Unfused, gn_silu_apply_kernel reads h and x and writes the block output (3 tensor passes, at the HBM rate) and the
// 4x4 / stride-2 convolution reads it back through the generic implicit GEMM, whose 128 x 64 tile re-streams the
// 128 KB weight matrix from L2 for every 128 output pixels: 0.44 PFLOP/s, bound by the L2-hit load rate of a CU.  Here
// a workgroup owns an 8 x 16 OUTPUT tile, i.e. the 16 x 32 block of input pixels it alone covers plus a one-pixel halo:
//   phase 1  load h and x of the 18 x 34 input pixels (all loads of a batch in flight before the first use), apply
//            GroupNorm + SiLU + residual in registers -- the arithmetic of gn_silu_apply_kernel, bit for bit -- write the
//            16 x 32 interior to HBM (that IS the block output / skip tensor: every input pixel is interior to exactly
//            one tile) and park the bf16 tile in LDS, out-of-image pixels as the convolution's zero padding;
//   phase 2  the convolution on MFMA with the WEIGHTS RESIDENT IN REGISTERS: wave w owns 16 output channels, i.e. the
//            16 x 1024 slice of the weight matrix = 32 K32 fragments = 128 VGPRs, loaded once per (persistent)
//            workgroup; a tap is a pixel offset into the LDS tile (stride-2 columns: even and odd columns stored apart),
//            256 ds_read_b128 + 256 MFMAs per wave and tile;
//   epilogue bias, bf16, 8-byte stores straight from the accumulators (the output is a quarter of the input).
// 78 KB of LDS: two 4-wave workgroups per CU (Cout = 64) whose phases overlap; Cout = 128 runs 8 waves per workgroup.
#include "fd_common.h"

namespace {

constexpr int DT_OH = 8, DT_OW = 16;                       // output tile
constexpr int DT_IH = 2 * DT_OH + 2, DT_IW = 2 * DT_OW + 2;      // 18 x 34 input pixels
constexpr int DT_PIX = DT_IH * DT_IW;                      // 612
constexpr int DT_LDS = DT_PIX * 128;                       // 78336 bytes
constexpr int DT_NCH = DT_PIX * 8;                         // 16-byte chunks of the tile

struct DownParams {
    const bf16 *h, *x;                // [B,H,W,64]
    const float *mean_rstd, *gamma, *beta; int groups;
    bf16 *skip;                       // [B,H,W,64]
    const bf16 *w; const float *bias; // [Cout][16 taps x 64], K order (kh, kw, c)
    bf16 *out; int Cout;              // [B,H/2,W/2,Cout]
    int H, W, tiles_x, ntiles, tpw;
};

// LDS image of the tile: the pixels of a row are stored EVEN columns first, then the odd ones (slot = (c & 1) * 17 + c / 2):
// the 16 lanes of a fragment read columns 2 fr + kw, i.e. 16 CONSECUTIVE slots, and with the 16-byte chunks XORed by
// slot & 7 a ds_read_b128 lane group ({0-3, 12-15, 20-27} ...: MI355X_MICROARCH.md, LDS) hits 16 distinct 16-byte bank
// groups of the 64-bank array for every tap -- the layout of the 3x3 halo kernel.  (Stored in column order, all 16 lanes
// sit on pixels of one parity and use half of the banks: 47 % of the LDS cycles were conflicts, PMC.)
__device__ __forceinline__ int dt_slot(int c) { return (c & 1) * (DT_IW / 2) + (c >> 1); }
__device__ __forceinline__ int dt_off(int row, int slot, int chunk) { return (row * DT_IW + slot) * 128 + ((chunk ^ (slot & 7)) << 4); }

template <int NW, int BATCH>
__global__ __launch_bounds__(64 * NW, 2) void down_fused_kernel(const DownParams p) {
    constexpr int NT = 64 * NW, NLD = (DT_NCH + NT - 1) / NT;
    extern __shared__ __attribute__((aligned(16))) unsigned char tile[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fg = lane >> 4;
    const int64_t img = blockIdx.y;
    const int OH = p.H / 2, OW = p.W / 2;
    const bf16 *hin = p.h + img * p.H * p.W * 64, *xin = p.x + img * p.H * p.W * 64;
    bf16 *skip = p.skip + img * p.H * p.W * 64;
    bf16 *outp = p.out + img * OH * OW * p.Cout;

    // ---- once per workgroup: this wave's 16 x 1024 weight slice as 32 MFMA A fragments (row = channel, 8 consecutive k
    // per lane), the GroupNorm scale / shift of this thread's 8 channels, the bias of this lane's 4 output channels
    const int cb = 16 * wave;
    bf16x8 wf[32];
    {
        const bf16 *wr = p.w + (int64_t)(cb + fr) * 1024 + fg * 8;
#pragma unroll
        for (int q = 0; q < 32; ++q) wf[q] = *(const bf16x8 *)(wr + 32 * q);
    }
    const int v = tid & 7;                                  // channel chunk of every load of this thread (NT % 8 == 0)
    f32x2 sc[4], sh[4];
    {
        const int cpg = 64 / p.groups;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float s2[2], h2[2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int c = v * 8 + 2 * j + e, g = c / cpg;
                const float mean = p.mean_rstd[(img * p.groups + g) * 2], rstd = p.mean_rstd[(img * p.groups + g) * 2 + 1];
                s2[e] = rstd * p.gamma[c];
                h2[e] = p.beta[c] - mean * s2[e];
            }
            sc[j] = f32x2{s2[0], s2[1]};
            sh[j] = f32x2{h2[0], h2[1]};
        }
    }
    float bias[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) bias[e] = p.bias ? p.bias[cb + 4 * fg + e] : 0.f;
    // LDS byte offsets of this lane's B fragments: column 2 fr + kw, K32 step ks (the row is an immediate)
    int boff[4][2];
#pragma unroll
    for (int kw = 0; kw < 4; ++kw)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) boff[kw][ks] = dt_off(0, dt_slot(2 * fr + kw), ks * 4 + fg);

    const int tile0 = blockIdx.x * p.tpw;
    for (int tt = 0; tt < p.tpw; ++tt) {
        const int t = tile0 + tt;
        if (t >= p.ntiles) break;                           // workgroup-uniform
        const int oy0 = (t / p.tiles_x) * DT_OH, ox0 = (t % p.tiles_x) * DT_OW;
        const int iy0 = 2 * oy0 - 1, ix0 = 2 * ox0 - 1;
        // ---- phase 1: h, x -> x + SiLU(GN(h)) -> skip (interior) and the LDS tile
        int tl = tid;
        asm volatile("" : "+v"(tl));                        // keep the per-chunk geometry inside the loop (registers)
#pragma unroll 1
        for (int i0 = 0; i0 < NLD; i0 += BATCH) {
            u32x4 hr[BATCH], xr[BATCH];
#pragma unroll
            for (int k = 0; k < BATCH; ++k) {
                const int id = min(tl + NT * (i0 + k), DT_NCH - 1);
                const int pix = id >> 3;
                const int r = (int)(__umul24((unsigned)pix, 1928u) >> 16);      // pix / 34 for pix < 1024 (34 * 1928 = 65552)
                const int c = pix - r * DT_IW;
                const int iy = min(max(iy0 + r, 0), p.H - 1), ix = min(max(ix0 + c, 0), p.W - 1);
                const int off = (__umul24(iy, p.W) + ix) * 64 + v * 8;
                hr[k] = *(const u32x4 *)(hin + off);
                xr[k] = *(const u32x4 *)(xin + off);
            }
#pragma unroll
            for (int k = 0; k < BATCH; ++k) {
                const int idr = tl + NT * (i0 + k);
                if (i0 + k >= NLD || idr >= DT_NCH) continue;
                const int pix = idr >> 3;
                const int r = (int)(__umul24((unsigned)pix, 1928u) >> 16);
                const int c = pix - r * DT_IW;
                const int iy = iy0 + r, ix = ix0 + c;
                const uint32_t hw4[4] = {hr[k].x, hr[k].y, hr[k].z, hr[k].w}, xw4[4] = {xr[k].x, xr[k].y, xr[k].z, xr[k].w};
                uint32_t o[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const f32x2 y = fd_silu2(fd_unpack_bf16(hw4[j]) * sc[j] + sh[j]) + fd_unpack_bf16(xw4[j]);
                    o[j] = fd_pack_bf16(y);
                }
                u32x4 ov = {o[0], o[1], o[2], o[3]};
                if (r >= 1 && r <= 2 * DT_OH && c >= 1 && c <= 2 * DT_OW)       // interior: this tile owns the pixel
                    *(u32x4 *)(skip + ((__umul24(iy, p.W) + ix) * 64 + v * 8)) = ov;
                const bool inimg = (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
                if (!inimg) ov = (u32x4){0, 0, 0, 0};        // the convolution's zero padding
                *(u32x4 *)(tile + dt_off(r, dt_slot(c), v)) = ov;
            }
        }
        __syncthreads();
        // ---- phase 2: 16 taps x 2 K32 steps, 8 output rows (m-tiles of 16 pixels) per wave
        f32x4 acc[DT_OH];
#pragma unroll
        for (int m = 0; m < DT_OH; ++m) acc[m] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kh = 0; kh < 4; ++kh)
#pragma unroll
            for (int kw = 0; kw < 4; ++kw)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const bf16x8 a = wf[(kh * 4 + kw) * 2 + ks];
                    bf16x8 bq[DT_OH];
#pragma unroll
                    for (int m = 0; m < DT_OH; ++m) bq[m] = *(const bf16x8 *)(tile + (2 * m + kh) * DT_IW * 128 + boff[kw][ks]);
                    // accumulator TIED through inline asm (hipcc leaves the builtin's destination and C operand apart and
                    // spills, see fd_pwgemm.hip); the fence keeps the next step's eight LDS reads from being hoisted over
                    // this step's (32 more live registers per step hoisted)
#pragma unroll
                    for (int m = 0; m < DT_OH; ++m)
                        asm(FD_MFMA16_ASM " %0, %1, %2, %0" : "+v"(acc[m]) : "v"(a), "v"(bq[m]));
                    __builtin_amdgcn_sched_barrier(0);
                }
        static_assert(DT_OH == 8, "FD_TIE8");
        asm volatile(FD_MFMA_ASM_DRAIN : FD_TIE8(acc) :: "memory");    // the last MFMAs' results before the epilogue reads them (fd_common.h)
        // ---- epilogue: lane (fr, fg) holds channels cb + 4 fg .. + 3 of output pixel (oy0 + m, ox0 + fr)
#pragma unroll
        for (int m = 0; m < DT_OH; ++m) {
            const f32x2 lo = {acc[m][0] + bias[0], acc[m][1] + bias[1]}, hi = {acc[m][2] + bias[2], acc[m][3] + bias[3]};
            uint2 ov;
            ov.x = fd_pack_bf16(lo);
            ov.y = fd_pack_bf16(hi);
            *(uint2 *)(outp + ((__umul24(oy0 + m, OW) + ox0 + fr) * p.Cout + cb + 4 * fg)) = ov;
        }
        __syncthreads();                                    // the tile is rewritten by the next phase 1
    }
}

}  // namespace

// `dtype | FD_OPT_LOW_LATENCY`: a lone slice has too few tiles per workgroup to amortise the 128 KB of weights a
// workgroup loads; the one-slice kernel set keeps the apply pass + implicit GEMM.
extern "C" int fd_gn_apply_down4x4_ok(int dtype_opts, int C, int Cout, int H, int W) {
    const bool off = fd_dev(FD_DEV_NO_DOWNFUSE);       // development switch
    const int dtype = dtype_opts & 0xff;
    return !off && !(dtype_opts & FD_OPT_LOW_LATENCY) && dtype == FD_BF16 && C == 64 && (Cout == 64 || Cout == 128) &&
           H % (2 * DT_OH) == 0 && W % (2 * DT_OW) == 0 && (int64_t)H * W >= 16384 && (int64_t)H * W * 128 < (1ll << 31);
}

extern "C" int fd_gn_apply_down4x4(int dtype, const void *h, const void *x, const float *mean_rstd, const float *gamma,
                                   const float *beta, int groups, void *skip, const void *w, const float *bias, void *out,
                                   int B, int H, int W, int C, int Cout, void *stream) {
    FD_REQUIRE(fd_gn_apply_down4x4_ok(dtype & 0xff, C, Cout, H, W),
               "fd_gn_apply_down4x4: unsupported shape (bf16, C = 64, Cout in {64, 128}, H %% 16, W %% 32): C=%d Cout=%d H=%d W=%d",
               C, Cout, H, W);
    FD_REQUIRE(h && x && mean_rstd && gamma && beta && skip && w && out, "fd_gn_apply_down4x4: null pointer");
    FD_REQUIRE(groups > 0 && 64 % groups == 0 && (64 / groups) % 2 == 0, "fd_gn_apply_down4x4: groups must divide 64 into even sizes");
    DownParams p;
    p.h = (const bf16 *)h; p.x = (const bf16 *)x; p.mean_rstd = mean_rstd; p.gamma = gamma; p.beta = beta; p.groups = groups;
    p.skip = (bf16 *)skip; p.w = (const bf16 *)w; p.bias = bias; p.out = (bf16 *)out; p.Cout = Cout;
    p.H = H; p.W = W;
    p.tiles_x = (W / 2) / DT_OW;
    p.ntiles = p.tiles_x * ((H / 2) / DT_OH);
    // persistent: one wave of resident workgroups, consecutive tiles each (row-major: neighbours share halo columns in L2;
    // the partition does not touch the results: every tile is computed by itself)
    const int tpw_env = fd_dev(FD_DEV_DOWN_TPW);      // development
    const int wgs = Cout == 64 ? 512 : 256;                 // resident workgroups of the chip (2 / 1 per CU)
    int tpw = tpw_env > 0 ? tpw_env : (int)(((int64_t)p.ntiles * B + wgs - 1) / wgs);
    if (tpw < 1) tpw = 1;
    p.tpw = tpw;
    dim3 grid((p.ntiles + tpw - 1) / tpw, B);
    // (loads per thread and round trip: 5 / 7 / 10 measured 208 / 210 / 215 us at level 0 -- the kernel runs at the ~4.2 TB/s
    //  the streaming row-GEMMs reach, its phases alternate between the two workgroups of a CU)
    if (Cout == 64) {
        (void)hipFuncSetAttribute((const void *)down_fused_kernel<4, 5>, hipFuncAttributeMaxDynamicSharedMemorySize, DT_LDS);
        hipLaunchKernelGGL((down_fused_kernel<4, 5>), grid, dim3(256), DT_LDS, (hipStream_t)stream, p);
    } else {
        (void)hipFuncSetAttribute((const void *)down_fused_kernel<8, 5>, hipFuncAttributeMaxDynamicSharedMemorySize, DT_LDS);
        hipLaunchKernelGGL((down_fused_kernel<8, 5>), grid, dim3(512), DT_LDS, (hipStream_t)stream, p);
    }
    FD_LAUNCH_OK("fd_gn_apply_down4x4");
    return FD_OK;
}
