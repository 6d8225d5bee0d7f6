// fd_initconv.hip -- the UNet's init_conv: 7x7, pad 3, 2 (3 with input_condition) input planes ->
// dim channels (src/DADiff.py:558, 704), bf16 mode.
//
// Through the generic implicit GEMM this layer costs 305 us at batch 8: its K axis is 49 taps x 8
// zero-padded channels (6 of 8 are padding), every K tile straddles taps (integer-division decode per
// tile) and the A operand is an NHWC copy of the two fp32 image planes made by a separate kernel.  The
// layer itself is output-bound: 0.27 GB written per launch, ~55 us at the HBM rate.
// Here a workgroup owns a 16 x 16 pixel tile:
//   * the (16+6) x (16+6+2) halo of the 2-3 fp32 planes goes straight into LDS as bf16 "pixel = 4
//     channel slots" words (8 bytes) -- no packed NHWC copy of the input exists;
//   * K is ordered (kh, kw, c) with kw padded 7 -> 8 and c -> 4: one filter row is exactly one K32 MFMA
//     step, and a lane's 8 consecutive k are two neighbouring pixels = two ds_read_b64;
//   * the MFMA is issued transposed (D^T = W . X^T, weight rows permuted as in fd_gemm_rows.hip), so a
//     lane ends with 8 consecutive output channels of one pixel: bias add and one 16-byte store;
//   * all 7 x (Cout/16) weight fragments live in registers for the whole tile (loaded once, L2).
#include "fd_common.h"

namespace {

constexpr int IT = 16, IHY = IT + 6, IHX = IT + 8;      // tile, halo rows, halo columns (22 + 1 for kw = 7 + 1 pad)

template <int NG>      // NG = Cout / 32
__global__ __launch_bounds__(256, 2) void initconv7_kernel(const float *__restrict__ p0, const float *__restrict__ p1,
                                                          const float *__restrict__ p2, const bf16 *__restrict__ wk,
                                                          const float *__restrict__ bias, bf16 *__restrict__ out,
                                                          int H, int W) {
    __shared__ __attribute__((aligned(16))) uint2 sx[IHY * IHX];     // [row][col] -> 4 bf16 channel slots
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fg = lane >> 4;
    const int tiles_x = W / IT;
    const int ty0 = (blockIdx.x / tiles_x) * IT, tx0 = (blockIdx.x % tiles_x) * IT;
    const int64_t img = blockIdx.y, plane = (int64_t)H * W;
    // weight fragments: row = output channel (permuted), 8 consecutive k of filter row kh
    const int rperm = 8 * (fr >> 2) + (fr & 3);
    bf16x8 wa[NG][7], wb[NG][7];
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
        for (int kh = 0; kh < 7; ++kh) {
            wa[g][kh] = *(const bf16x8 *)(wk + (int64_t)(32 * g + rperm) * 224 + kh * 32 + fg * 8);
            wb[g][kh] = *(const bf16x8 *)(wk + (int64_t)(32 * g + rperm + 4) * 224 + kh * 32 + fg * 8);
        }
    // halo -> LDS
    for (int i = tid; i < IHY * IHX; i += 256) {
        const int hy = i / IHX, hx = i - hy * IHX;
        const int y = ty0 + hy - 3, x = tx0 + hx - 3;
        float v0 = 0.f, v1 = 0.f, v2 = 0.f;
        if (y >= 0 && y < H && x >= 0 && x < W) {
            const int64_t o = img * plane + (int64_t)y * W + x;
            v0 = p0[o];
            if (p1) v1 = p1[o];
            if (p2) v2 = p2[o];
        }
        const bf16 b0 = (bf16)v0, b1 = (bf16)v1;
        uint2 wv;
        wv.x = (uint32_t)__builtin_bit_cast(uint16_t, b0) | ((uint32_t)__builtin_bit_cast(uint16_t, b1) << 16);
        if (p2) {
            const bf16 b2 = (bf16)v2;
            wv.y = (uint32_t)__builtin_bit_cast(uint16_t, b2);
        } else {
            // two planes leave two of the four channel slots free: they carry the bf16 ROUNDING RESIDUALS of the
            // planes (x = hi + lo, weights duplicated on the host), so the image enters the network with ~16
            // mantissa bits instead of 8 at no extra MFMA work (input rounding alone was 6e-3 of output drift)
            const bf16 l0 = (bf16)(v0 - (float)b0), l1 = (bf16)(v1 - (float)b1);
            wv.y = (uint32_t)__builtin_bit_cast(uint16_t, l0) | ((uint32_t)__builtin_bit_cast(uint16_t, l1) << 16);
        }
        sx[i] = wv;
    }
    __syncthreads();
    float bs[NG][8];
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        if (bias) load8(bias + 32 * g + 8 * fg, bs[g]);
        else {
#pragma unroll
            for (int e = 0; e < 8; ++e) bs[g][e] = 0.f;
        }
    }
    constexpr int CO = 32 * NG;
#pragma unroll 1
    for (int i = 0; i < 4; ++i) {
        const int ty = 4 * wave + i;                       // tile row = one m-tile of 16 pixels
        f32x4 a0[NG], a1[NG];
#pragma unroll
        for (int g = 0; g < NG; ++g) { a0[g] = (f32x4){0.f, 0.f, 0.f, 0.f}; a1[g] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int kh = 0; kh < 7; ++kh) {
            // lane (pixel fr, k-group fg): kw = 2 fg, 2 fg + 1 -> halo columns fr + 2 fg, + 1
            const uint2 *src = sx + (ty + kh) * IHX + fr + 2 * fg;
            const uint2 q0 = src[0], q1 = src[1];
            const u32x4 xv = {q0.x, q0.y, q1.x, q1.y};
            const bf16x8 xb = __builtin_bit_cast(bf16x8, xv);
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                a0[g] = FD_MFMA16(wa[g][kh], xb, a0[g], 0, 0, 0);
                a1[g] = FD_MFMA16(wb[g][kh], xb, a1[g], 0, 0, 0);
            }
        }
        const int y = ty0 + ty, x = tx0 + fr;
        bf16 *op = out + ((img * H + y) * W + x) * CO + 8 * fg;
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            float val[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) { val[e] = a0[g][e] + bs[g][e]; val[4 + e] = a1[g][e] + bs[g][4 + e]; }
            store8(op + 32 * g, val);
        }
    }
}

// The same layer for the fp32s engine (fp32 storage, split-bf16 contractions), two input planes: the LDS image is the one
// above -- slots (x0_hi, x1_hi, x0_lo, x1_lo) -- and the weights come as TWO packed sets: `wk` = bf16(w) in all four slots
// (w_hi . x_hi + w_hi . x_lo) and `wl` = bf16(w - bf16(w)) in slots 0, 1, zero in 2, 3 (w_lo . x_hi): the three terms of
// every other fp32s product, two MFMAs per filter row.  Output channels in groups of 32 with that group's 28 fragments in
// registers; fp32 stores.  The generic split implicit GEMM took 610-670 us per batch-8 launch at 512 x 512 for this layer
// (K = 49 taps x 8 zero-padded channels, an NHWC copy of the planes in front of it).
template <int NG>
__global__ __launch_bounds__(256, 2) void initconv7_f32s_kernel(const float *__restrict__ p0, const float *__restrict__ p1,
                                                               const bf16 *__restrict__ wk, const bf16 *__restrict__ wl,
                                                               const float *__restrict__ bias, float *__restrict__ out,
                                                               int H, int W) {
    __shared__ __attribute__((aligned(16))) uint2 sx[IHY * IHX];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fg = lane >> 4;
    const int tiles_x = W / IT;
    const int ty0 = (blockIdx.x / tiles_x) * IT, tx0 = (blockIdx.x % tiles_x) * IT;
    const int64_t img = blockIdx.y, plane = (int64_t)H * W;
    const int rperm = 8 * (fr >> 2) + (fr & 3);
    for (int i = tid; i < IHY * IHX; i += 256) {
        const int hy = i / IHX, hx = i - hy * IHX;
        const int y = ty0 + hy - 3, x = tx0 + hx - 3;
        float v0 = 0.f, v1 = 0.f;
        if (y >= 0 && y < H && x >= 0 && x < W) {
            const int64_t o = img * plane + (int64_t)y * W + x;
            v0 = p0[o];
            v1 = p1[o];
        }
        const bf16 b0 = (bf16)v0, b1 = (bf16)v1;
        const bf16 l0 = (bf16)(v0 - (float)b0), l1 = (bf16)(v1 - (float)b1);
        uint2 wv;
        wv.x = (uint32_t)__builtin_bit_cast(uint16_t, b0) | ((uint32_t)__builtin_bit_cast(uint16_t, b1) << 16);
        wv.y = (uint32_t)__builtin_bit_cast(uint16_t, l0) | ((uint32_t)__builtin_bit_cast(uint16_t, l1) << 16);
        sx[i] = wv;
    }
    __syncthreads();
    constexpr int CO = 32 * NG;
#pragma unroll 1
    for (int g = 0; g < NG; ++g) {
        bf16x8 wa[7], wb[7], la[7], lb[7];
#pragma unroll
        for (int kh = 0; kh < 7; ++kh) {
            const int64_t oa = (int64_t)(32 * g + rperm) * 224 + kh * 32 + fg * 8, ob = oa + 4 * 224;
            wa[kh] = *(const bf16x8 *)(wk + oa);
            wb[kh] = *(const bf16x8 *)(wk + ob);
            la[kh] = *(const bf16x8 *)(wl + oa);
            lb[kh] = *(const bf16x8 *)(wl + ob);
        }
        float bs[8];
        if (bias) load8(bias + 32 * g + 8 * fg, bs);
        else {
#pragma unroll
            for (int e = 0; e < 8; ++e) bs[e] = 0.f;
        }
#pragma unroll 1
        for (int i = 0; i < 4; ++i) {
            const int ty = 4 * wave + i;
            f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kh = 0; kh < 7; ++kh) {
                const uint2 *src = sx + (ty + kh) * IHX + fr + 2 * fg;
                const uint2 q0 = src[0], q1 = src[1];
                const u32x4 xv = {q0.x, q0.y, q1.x, q1.y};
                const bf16x8 xb = __builtin_bit_cast(bf16x8, xv);
                a0 = FD_MFMA16(la[kh], xb, a0, 0, 0, 0);
                a1 = FD_MFMA16(lb[kh], xb, a1, 0, 0, 0);
                a0 = FD_MFMA16(wa[kh], xb, a0, 0, 0, 0);
                a1 = FD_MFMA16(wb[kh], xb, a1, 0, 0, 0);
            }
            const int y = ty0 + ty, x = tx0 + fr;
            float *op = out + ((img * H + y) * W + x) * CO + 32 * g + 8 * fg;
            float val[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) { val[e] = a0[e] + bs[e]; val[4 + e] = a1[e] + bs[4 + e]; }
            store8(op, val);
        }
    }
}

}  // namespace

// fp32s form (fp32 storage, split-bf16 contraction), two input planes: w_hi_packed / w_lo_packed are fd_init_conv7's packing of
// bf16(w) (all four channel slots: the planes and their rounding residuals) and of bf16(w - bf16(w)) (slots 0, 1 only).
extern "C" int fd_init_conv7_f32s(const float *p0, const float *p1, const void *w_hi_packed, const void *w_lo_packed,
                                  const float *bias, void *out, int B, int H, int W, int Cout, void *stream) {
    FD_REQUIRE((Cout == 32 || Cout == 64) && H % IT == 0 && W % IT == 0, "fd_init_conv7_f32s: needs Cout in {32, 64}, H, W multiples "
               "of 16 (Cout=%d H=%d W=%d)", Cout, H, W);
    FD_REQUIRE(p0 && p1 && w_hi_packed && w_lo_packed && out, "fd_init_conv7_f32s: null pointer");
    FD_REQUIRE((((uintptr_t)bias | (uintptr_t)out | (uintptr_t)w_hi_packed | (uintptr_t)w_lo_packed) & 15) == 0, "fd_init_conv7_f32s: 16-byte alignment");
    dim3 grid((H / IT) * (W / IT), B), block(256);
    if (Cout == 64)
        hipLaunchKernelGGL(initconv7_f32s_kernel<2>, grid, block, 0, (hipStream_t)stream, p0, p1, (const bf16 *)w_hi_packed,
                           (const bf16 *)w_lo_packed, bias, (float *)out, H, W);
    else
        hipLaunchKernelGGL(initconv7_f32s_kernel<1>, grid, block, 0, (hipStream_t)stream, p0, p1, (const bf16 *)w_hi_packed,
                           (const bf16 *)w_lo_packed, bias, (float *)out, H, W);
    FD_LAUNCH_OK("fd_init_conv7_f32s");
    return FD_OK;
}

extern "C" int fd_init_conv7_ok(int dtype, int Cout, int H, int W) {
    return dtype == FD_BF16 && (Cout == 32 || Cout == 64) && H % IT == 0 && W % IT == 0;
}

extern "C" int fd_init_conv7(int dtype, const float *p0, const float *p1, const float *p2, const void *w_packed,
                             const float *bias, void *out, int B, int H, int W, int Cout, void *stream) {
    FD_REQUIRE(fd_init_conv7_ok(dtype, Cout, H, W), "fd_init_conv7: needs bf16, Cout in {32, 64}, H, W multiples of 16 "
               "(Cout=%d H=%d W=%d)", Cout, H, W);
    FD_REQUIRE(p0 && w_packed && out, "fd_init_conv7: null pointer");
    FD_REQUIRE(!bias || ((uintptr_t)bias & 15) == 0, "fd_init_conv7: bias must be 16-byte aligned");
    dim3 grid((H / IT) * (W / IT), B), block(256);
    if (Cout == 64)
        hipLaunchKernelGGL(initconv7_kernel<2>, grid, block, 0, (hipStream_t)stream, p0, p1, p2, (const bf16 *)w_packed, bias,
                           (bf16 *)out, H, W);
    else
        hipLaunchKernelGGL(initconv7_kernel<1>, grid, block, 0, (hipStream_t)stream, p0, p1, p2, (const bf16 *)w_packed, bias,
                           (bf16 *)out, H, W);
    FD_LAUNCH_OK("fd_init_conv7");
    return FD_OK;
}
