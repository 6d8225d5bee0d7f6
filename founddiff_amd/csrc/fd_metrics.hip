// fd_metrics.hip -- image-quality metrics of the evaluation loop on the device (no D2H per slice):
// PSNR (max_val 1), RMSE and SSIM (11x11 Gaussian, sigma 1.5, reflect padding, clamp to [0,1],
// mean) as in /root/reference/src/util.py:188-236 (kornia get_gaussian_kernel2d + filter2d).
#include "fd_common.h"

namespace {

constexpr int MT = 16;          // output tile
constexpr int MR = 5;           // window radius (11 taps)
constexpr int MH = MT + 2 * MR; // halo tile

__device__ __forceinline__ int reflect(int i, int n) {
    if (i < 0) i = -i;
    if (i >= n) i = 2 * n - 2 - i;
    // halo positions of tiles that hang over a small image (n < 11) can reflect twice: they only feed
    // masked-out outputs, but the load must stay inside the buffer
    return min(max(i, 0), n - 1);
}

__global__ __launch_bounds__(256) void metrics_partial_kernel(const float *__restrict__ pred,
                                                             const float *__restrict__ tgt, int H, int W,
                                                             float *__restrict__ partial) {
    __shared__ float sp[MH][MH + 1], st[MH][MH + 1];
    __shared__ float g[11];
    __shared__ float red[2][4];
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    const int b = blockIdx.z, x0 = blockIdx.x * MT, y0 = blockIdx.y * MT;
    const float *p = pred + (int64_t)b * H * W, *t = tgt + (int64_t)b * H * W;
    if (tid < 11) {
        float s = 0.f;
        for (int i = 0; i < 11; ++i) s += expf(-(float)((i - 5) * (i - 5)) / (2.f * 1.5f * 1.5f));
        g[tid] = expf(-(float)((tid - 5) * (tid - 5)) / (2.f * 1.5f * 1.5f)) / s;
    }
    for (int i = tid; i < MH * MH; i += 256) {
        const int hy = i / MH, hx = i - hy * MH;
        const int yy = reflect(y0 + hy - MR, H), xx = reflect(x0 + hx - MR, W);
        sp[hy][hx] = p[(int64_t)yy * W + xx];
        st[hy][hx] = t[(int64_t)yy * W + xx];
    }
    __syncthreads();
    float se = 0.f, ss = 0.f;
    const int x = x0 + tx, y = y0 + ty;
    if (x < W && y < H) {
        float m1 = 0.f, m2 = 0.f, s11 = 0.f, s22 = 0.f, s12 = 0.f;
        for (int dy = 0; dy < 11; ++dy)
#pragma unroll
            for (int dx = 0; dx < 11; ++dx) {
                const float w = g[dy] * g[dx];
                const float a = sp[ty + dy][tx + dx], c = st[ty + dy][tx + dx];
                m1 += w * a; m2 += w * c;
                s11 += w * a * a; s22 += w * c * c; s12 += w * a * c;
            }
        const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
        const float v1 = s11 - m1 * m1, v2 = s22 - m2 * m2, v12 = s12 - m1 * m2;
        float s = ((2.f * m1 * m2 + C1) * (2.f * v12 + C2)) / ((m1 * m1 + m2 * m2 + C1) * (v1 + v2 + C2));
        ss = fminf(fmaxf(s, 0.f), 1.f);
        const float d = sp[ty + MR][tx + MR] - st[ty + MR][tx + MR];
        se = d * d;
    }
    se = wave_sum(se);
    ss = wave_sum(ss);
    if ((tid & 63) == 0) { red[0][tid >> 6] = se; red[1][tid >> 6] = ss; }
    __syncthreads();
    if (tid == 0) {
        float *o = partial + (((int64_t)b * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 2;
        o[0] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
        o[1] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    }
}

__global__ __launch_bounds__(256) void metrics_finalize_kernel(const float *__restrict__ partial, int nblk,
                                                              double inv_n, float *__restrict__ out) {
    __shared__ double sh[2][256];
    const int b = blockIdx.x, tid = threadIdx.x;
    double se = 0.0, ss = 0.0;
    for (int i = tid; i < nblk; i += 256) {
        se += partial[((int64_t)b * nblk + i) * 2];
        ss += partial[((int64_t)b * nblk + i) * 2 + 1];
    }
    sh[0][tid] = se;
    sh[1][tid] = ss;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) { sh[0][tid] += sh[0][tid + o]; sh[1][tid] += sh[1][tid + o]; }
        __syncthreads();
    }
    if (tid == 0) {
        const double mse = sh[0][0] * inv_n;
        out[b * 3 + 0] = (float)(10.0 * log10(1.0 / mse));
        out[b * 3 + 1] = (float)(sh[1][0] * inv_n);
        out[b * 3 + 2] = (float)sqrt(mse);
    }
}

}  // namespace

extern "C" int fd_metrics_nblk(int H, int W) { return cdiv(H, MT) * cdiv(W, MT); }

extern "C" int fd_metrics(const float *pred, const float *target, int B, int H, int W, float *partial,
                          float *out, void *stream) {
    FD_REQUIRE(pred && target && partial && out && H > MR && W > MR, "fd_metrics: bad args (need H,W > 5)");
    dim3 grid(cdiv(W, MT), cdiv(H, MT), B);
    hipLaunchKernelGGL(metrics_partial_kernel, grid, dim3(256), 0, (hipStream_t)stream, pred, target, H, W, partial);
    hipLaunchKernelGGL(metrics_finalize_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, partial,
                       fd_metrics_nblk(H, W), 1.0 / ((double)H * W), out);
    FD_LAUNCH_OK("fd_metrics");
    return FD_OK;
}
