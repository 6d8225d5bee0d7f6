// fd_sched.hip -- the ancestral sampler's per-step scheduler work without a host in the loop, and its step noise
// as a counter-based stream keyed per SLICE (src/DADiff.py:1222-1273: p_sample_loop / p_sample / q_posterior).
//
// The reference draws `torch.randn_like(x)` from the device generator at every step (1228): the noise a slice
// receives depends on its position in the batch and on everything drawn before it, so a volume sharded over ranks
// (BASELINE configs[3]: 64 slices over 8 GPUs) would depend on the world size.  Here the noise of (slice, step,
// pixel) is a pure function of (seed of the slice, t, pixel index): Philox4x32-10 keyed by the slice's 64-bit seed,
// counter = (pixel / 4, t, domain tag, 0), Box-Muller on the four 32-bit outputs.  Nothing is stored, nothing is
// drawn on the host, and the same slice gets the same stream on any rank, in any batch, on either HIP stream.
// oracle/keyed_noise.py restates the generator in numpy (tests/).
//
// fd_ancestral_begin / fd_res_posterior_step_keyed read the timestep from a device counter and the posterior
// coefficients from a device table, so a chunk of steps can be captured in one HIP graph and replayed.
#include "fd_common.h"

namespace {

struct u32q { uint32_t x, y, z, w; };

__device__ __forceinline__ u32q philox4x32_10(u32q c, uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c.x), lo0 = 0xD2511F53u * c.x;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c.z), lo1 = 0xCD9E8D57u * c.z;
        c = u32q{hi1 ^ c.y ^ k0, lo1, hi0 ^ c.w ^ k1, lo0};
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return c;
}

// four standard normals of (seed, t, group g of four consecutive pixels)
__device__ __forceinline__ void keyed_normal4(uint64_t seed, uint32_t t, uint32_t g, float z[4]) {
    const u32q r = philox4x32_10(u32q{g, t, 0x46444e5au, 0u}, (uint32_t)seed, (uint32_t)(seed >> 32));
    // 23-bit uniforms (k + 0.5) * 2^-23, k < 2^23: k + 0.5 has 24 significant bits, so every value is exactly
    // representable, strictly inside (0, 1) and the grid is uniform over the whole range
    const float u0 = ((float)(r.x >> 9) + 0.5f) * (1.f / 8388608.f), u1 = ((float)(r.y >> 9) + 0.5f) * (1.f / 8388608.f);
    const float u2 = ((float)(r.z >> 9) + 0.5f) * (1.f / 8388608.f), u3 = ((float)(r.w >> 9) + 0.5f) * (1.f / 8388608.f);
    const float ra = sqrtf(-2.f * logf(u0)), rb = sqrtf(-2.f * logf(u2));
    float s0, c0, s1, c1;
    sincospif(2.f * u1, &s0, &c0);
    sincospif(2.f * u3, &s1, &c1);
    z[0] = ra * c0; z[1] = ra * s0; z[2] = rb * c1; z[3] = rb * s1;
}

__device__ __forceinline__ float clamp1s(float v) { return fd_clamp1(v); }

__global__ void keyed_normal_kernel(const int64_t *__restrict__ seeds, int t, float *__restrict__ out, int64_t npix) {
    const int b = blockIdx.y;
    const uint64_t seed = (uint64_t)seeds[b];
    const int64_t ng = (npix + 3) / 4;
    for (int64_t g = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; g < ng; g += (int64_t)gridDim.x * blockDim.x) {
        float z[4];
        keyed_normal4(seed, (uint32_t)t, (uint32_t)g, z);
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (4 * g + e < npix) out[(int64_t)b * npix + 4 * g + e] = z[e];
    }
}

// start of an ancestral step: t <- t - 1, time[b] <- times[t] (the UNet's time input alphas_cumsum[t] * T)
__global__ void ancestral_begin_kernel(int *t_dev, const float *__restrict__ times, float *__restrict__ time_buf, int B) {
    const int t = max(*t_dev - 1, 0);          // a caller that runs more steps than the table has rows repeats t = 0
    __syncthreads();
    if (threadIdx.x == 0) *t_dev = t;
    for (int b = threadIdx.x; b < B; b += blockDim.x) time_buf[b] = times[t];
}

// xt / out may be the same buffer (in-place update): no __restrict__ on them; a thread reads its pixels before it
// writes them
__global__ void res_posterior_keyed_kernel(const float *__restrict__ mo, const float *xt, const float *__restrict__ xin,
                                           const float *__restrict__ coef_table, const int *__restrict__ t_dev,
                                           const int64_t *__restrict__ seeds, float *out, float *__restrict__ xs_out,
                                           int64_t npix) {
    const int b = blockIdx.y;
    const int t = max(*t_dev, 0);
    const float c1 = coef_table[t * 4], c2 = coef_table[t * 4 + 1], c3 = coef_table[t * 4 + 2];
    const float sd = t > 0 ? expf(0.5f * coef_table[t * 4 + 3]) : 0.f;       // no noise at t = 0 (src/DADiff.py:1228)
    const uint64_t seed = (uint64_t)seeds[b];
    const int64_t ng = (npix + 3) / 4;
    for (int64_t g = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; g < ng; g += (int64_t)gridDim.x * blockDim.x) {
        float z[4] = {0.f, 0.f, 0.f, 0.f};
        if (t > 0) keyed_normal4(seed, (uint32_t)t, (uint32_t)g, z);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int64_t i = 4 * g + e;
            if (i < npix) {
                const int64_t j = (int64_t)b * npix + i;
                const float pr = clamp1s(mo[j]);
                const float xs = clamp1s(xin[j] - pr);
                out[j] = c1 * xt[j] + c2 * pr + c3 * xs + sd * z[e];
                if (xs_out) xs_out[j] = xs;
            }
        }
    }
}

inline int g4(int64_t npix) { int64_t b = ((npix + 3) / 4 + 255) / 256; return (int)(b > 2048 ? 2048 : (b < 1 ? 1 : b)); }

}  // namespace

extern "C" int fd_keyed_normal(const int64_t *seeds, int t, float *out, int B, int64_t npix, void *stream) {
    FD_REQUIRE(seeds && out && B > 0 && npix > 0 && npix < (1ll << 33), "fd_keyed_normal: bad args");
    hipLaunchKernelGGL(keyed_normal_kernel, dim3(g4(npix), B), dim3(256), 0, (hipStream_t)stream, seeds, t, out, npix);
    FD_LAUNCH_OK("fd_keyed_normal");
    return FD_OK;
}

extern "C" int fd_ancestral_begin(int *t_dev, const float *times, float *time_buf, int B, void *stream) {
    FD_REQUIRE(t_dev && times && time_buf && B > 0, "fd_ancestral_begin: bad args");
    hipLaunchKernelGGL(ancestral_begin_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, t_dev, times, time_buf, B);
    FD_LAUNCH_OK("fd_ancestral_begin");
    return FD_OK;
}

extern "C" int fd_res_posterior_step_keyed(const float *model_out, const float *x_t, const float *x_in,
                                           const float *coef_table, const int *t_dev, const int64_t *seeds,
                                           float *img_out, float *x_start_out, int B, int64_t npix, void *stream) {
    FD_REQUIRE(model_out && x_t && x_in && coef_table && t_dev && seeds && img_out, "fd_res_posterior_step_keyed: null pointer");
    FD_REQUIRE(npix < (1ll << 33), "fd_res_posterior_step_keyed: image too large for the 32-bit pixel-group counter");
    hipLaunchKernelGGL(res_posterior_keyed_kernel, dim3(g4(npix), B), dim3(256), 0, (hipStream_t)stream, model_out, x_t,
                       x_in, coef_table, t_dev, seeds, img_out, x_start_out, npix);
    FD_LAUNCH_OK("fd_res_posterior_step_keyed");
    return FD_OK;
}
