// fd_scan_ref.hip -- the reference's ONE native-op interface at the C ABI:
//     selective_scan_cuda_core.fwd(u, delta, A, B, C, D, delta_bias, delta_softplus, nrows) -> (out, x, ...)
// (/root/reference/src/emamba2.py:154; tensor layout asserted at 124-149: everything fp32, contiguous in the
// last dim, B / C four-dimensional (b, K, N, L)).  A maintainer who keeps the reference's emamba2.py and only
// swaps the extension binds this entry point (founddiff_amd/selective_scan_cuda_core.py is that shim); the
// engine itself uses the fused NHWC form in fd_scan.hip, which never materialises u / delta / B / C per
// direction.
//
//     dt   = delta + delta_bias[d];  softplus(dt) when delta_softplus            (threshold 20, as torch)
//     h_t  = exp(dt A[d,n]) h_{t-1} + dt B[b,g,n,t] u[b,d,t]                      g = d / (KD / K)
//     out  = sum_n h_t C[b,g,n,t] + D[d] u[b,d,t]
//
// Layout is L-contiguous per (batch, channel) row, so the parallel axis inside a workgroup is the SEQUENCE:
// one workgroup owns one row; a tile is 256 threads x 4 consecutive positions (16-byte coalesced loads of u,
// delta, B_n, C_n).  Per state n: every thread composes the affine maps h -> a h + b of its 4 positions, a
// wave-level inclusive scan of the (a, b) pairs (6 lane shifts) plus one LDS exchange across the 4 waves
// yields each thread's carry-in, the thread replays its 4 positions from it and accumulates C_n h.  The
// tile-to-tile carry of every state lives in LDS.  fp32 throughout (SURVEY: chunked composition == sequential
// fp64 to 1.3e-7).  HBM bound: (2 + 2N/Dg) reads + 1 write of 4 bytes per (row, position), Dg = KD / K rows
// sharing one (B, C) group through L2.
#include "fd_common.h"

namespace {

constexpr int SR_T = 256, SR_E = 4, SR_TILE = SR_T * SR_E;

// (a2, b2) o (a1, b1): first map 1, then map 2
__device__ __forceinline__ void compose(float &a, float &b, float a_prev, float b_prev) {
    b = a * b_prev + b;
    a = a * a_prev;
}

__global__ __launch_bounds__(SR_T) void scan_ref_kernel(const float *__restrict__ u, const float *__restrict__ delta,
                                                       const float *__restrict__ A, const float *__restrict__ Bm,
                                                       const float *__restrict__ Cm, const float *__restrict__ Dv,
                                                       const float *__restrict__ dbias, int softplus,
                                                       float *__restrict__ out, float *__restrict__ x_last, int KD,
                                                       int K, int N, int64_t L) {
    __shared__ float s_carry[2][256];            // h of every state at the end of the previous tile, double-buffered
                                                 // over tiles: a tile reads [tp] and writes [tp ^ 1]
    __shared__ float s_wa[2][4], s_wb[2][4];     // wave totals, double-buffered over n: one barrier per state
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t row = blockIdx.x;              // b * KD + d
    const int b = (int)(row / KD), d = (int)(row - (int64_t)b * KD);
    const int g = d / (KD / K);
    const float *ur = u + row * L, *dr = delta + row * L;
    float *yr = out + row * L;
    const float *Br = Bm + ((int64_t)b * K + g) * N * L, *Cr = Cm + ((int64_t)b * K + g) * N * L;
    const float bias = dbias ? dbias[d] : 0.f, Dd = Dv ? Dv[d] : 0.f;
    for (int n = tid; n < N; n += SR_T) s_carry[0][n] = 0.f;
    const bool vec = (L & 3) == 0;               // rows are 16-byte aligned when L % 4 == 0
    __syncthreads();
    int par = 0, tp = 0;
    for (int64_t t0 = 0; t0 < L; t0 += SR_TILE, tp ^= 1) {
        const int64_t l0 = t0 + (int64_t)tid * SR_E;
        float uu[SR_E], dt[SR_E], y[SR_E];
        auto load4 = [&](const float *p, float (&v)[SR_E]) {
            if (vec && l0 + SR_E <= L) {
                const f32x4 q = *(const f32x4 *)(p + l0);
                v[0] = q[0]; v[1] = q[1]; v[2] = q[2]; v[3] = q[3];
            } else {
#pragma unroll
                for (int i = 0; i < SR_E; ++i) v[i] = l0 + i < L ? p[l0 + i] : 0.f;
            }
        };
        load4(ur, uu);
        load4(dr, dt);
#pragma unroll
        for (int i = 0; i < SR_E; ++i) {
            float v = dt[i] + bias;
            if (softplus) v = fd_softplus(v);
            dt[i] = l0 + i < L ? v : 0.f;        // positions past the end are the identity map (a = 1, b = 0)
            y[i] = Dd * uu[i];
        }
        for (int n = 0; n < N; ++n, par ^= 1) {
            const float An = A[(int64_t)d * N + n];
            float Bv[SR_E], Cv[SR_E], a[SR_E], bb[SR_E];
            load4(Br + (int64_t)n * L, Bv);
            load4(Cr + (int64_t)n * L, Cv);
            float Pa = 1.f, Pb = 0.f;            // this thread's 4 positions composed
#pragma unroll
            for (int i = 0; i < SR_E; ++i) {
                a[i] = __expf(dt[i] * An);
                bb[i] = dt[i] * Bv[i] * uu[i];
                Pb = a[i] * Pb + bb[i];
                Pa = a[i] * Pa;
            }
            // inclusive scan over the lanes of the wave
            float sa = Pa, sb = Pb;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const float pa = __shfl_up(sa, o, 64), pb = __shfl_up(sb, o, 64);
                if (lane >= o) compose(sa, sb, pa, pb);
            }
            if (lane == 63) { s_wa[par][wave] = sa; s_wb[par][wave] = sb; }
            // exclusive prefix inside the wave
            float ea = __shfl_up(sa, 1, 64), eb = __shfl_up(sb, 1, 64);
            if (lane == 0) { ea = 1.f; eb = 0.f; }
            __syncthreads();
            // carry-in of this thread = tile carry, then the waves before this one, then the lanes before it
            float h = s_carry[tp][n];
            for (int w = 0; w < wave; ++w) h = s_wa[par][w] * h + s_wb[par][w];
            h = ea * h + eb;
#pragma unroll
            for (int i = 0; i < SR_E; ++i) {
                h = a[i] * h + bb[i];
                y[i] += Cv[i] * h;
            }
            if (tid == SR_T - 1) s_carry[tp ^ 1][n] = h;     // the last position of the tile
        }
        __syncthreads();                         // s_carry[tp ^ 1] complete before the next tile reads it
        if (vec && l0 + SR_E <= L) {
            *(f32x4 *)(yr + l0) = (f32x4){y[0], y[1], y[2], y[3]};
        } else {
#pragma unroll
            for (int i = 0; i < SR_E; ++i)
                if (l0 + i < L) yr[l0 + i] = y[i];
        }
    }
    if (x_last)
        for (int n = tid; n < N; n += SR_T) x_last[row * N + n] = s_carry[tp][n];
}

}  // namespace

extern "C" int fd_selective_scan_fwd_f32(const float *u, const float *delta, const float *A, const float *B,
                                         const float *C, const float *D, const float *delta_bias,
                                         int delta_softplus, int nrows, int batch, int KD, int K, int N, int64_t L,
                                         float *out, float *x_last, void *stream) {
    FD_REQUIRE(u && delta && A && B && C && out, "fd_selective_scan_fwd_f32: null pointer");
    // the reference's own asserts (src/emamba2.py:129-130)
    FD_REQUIRE(nrows >= 1 && nrows <= 4, "fd_selective_scan_fwd_f32: nrows=%d not in 1..4", nrows);
    FD_REQUIRE(batch > 0 && KD > 0 && K > 0 && L > 0 && KD % (K * nrows) == 0,
               "fd_selective_scan_fwd_f32: u.shape[1]=%d must be a multiple of B.shape[1]*nrows=%d*%d", KD, K, nrows);
    FD_REQUIRE(N >= 1 && N <= 256, "fd_selective_scan_fwd_f32: d_state=%d not in 1..256", N);
    FD_REQUIRE((((uintptr_t)u | (uintptr_t)delta | (uintptr_t)B | (uintptr_t)C | (uintptr_t)out) & 15) == 0,
               "fd_selective_scan_fwd_f32: tensors must be 16-byte aligned");
    hipLaunchKernelGGL(scan_ref_kernel, dim3((unsigned)((int64_t)batch * KD)), dim3(SR_T), 0, (hipStream_t)stream, u, delta, A, B,
                       C, D, delta_bias, delta_softplus, out, x_last, KD, K, N, L);
    FD_LAUNCH_OK("fd_selective_scan_fwd_f32");
    return FD_OK;
}
