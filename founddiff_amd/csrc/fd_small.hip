// fd_small.hip -- fp32 conditioning path (time MLP, adaLN vectors, prompt path, DA-CLIP heads),
// scheduler math on fp32 images, and the library's error state.
#include <stdarg.h>
#include <stdio.h>
#include "fd_common.h"

static thread_local char g_err[512] = "";

void fd_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char *fd_last_error(void) { return g_err; }

// which 16-bit type this build stores and multiplies in (fd_common.h): 0 = bfloat16 (libfounddiff_hip.so), 1 = IEEE binary16
// (libfounddiff_hip_f16.so).  The host checks it against the torch dtype it is about to hand over.
extern "C" int fd_half_format(void) {
#ifdef FD_HALF_F16
    return 1;
#else
    return 0;
#endif
}

// ---- development switches (fd_common.h: FD_DEV_SWITCHES)
#ifdef FD_RELEASE
extern "C" const char *fd_dev_options(void) { return "release build: development switches compiled out (every one at its default)"; }
#else
namespace {
struct DevTable {
    int v[FD_DEV_COUNT];
    char text[2048];
    DevTable() {
        static const struct { const char *name; char kind; int dflt; } tab[] = {
#define FD_DEV_ROW(name, kind, dflt) {"FD_" #name, #kind[0], dflt},
            FD_DEV_SWITCHES(FD_DEV_ROW)
#undef FD_DEV_ROW
        };
        size_t n = (size_t)snprintf(text, sizeof text, "development build; switches set:");
        for (int i = 0; i < FD_DEV_COUNT; ++i) {
            const char *e = getenv(tab[i].name);
            v[i] = e ? (tab[i].kind == 'F' ? 1 : atoi(e)) : tab[i].dflt;
            if (e && n < sizeof text) n += (size_t)snprintf(text + n, sizeof text - n, " %s=%d", tab[i].name, v[i]);
        }
    }
};
const DevTable &dev_table() {
    static const DevTable t;        // read once; thread-safe initialisation
    return t;
}
}  // namespace
int fd_dev(int id) { return dev_table().v[id]; }
extern "C" const char *fd_dev_options(void) { return dev_table().text; }
#endif
extern "C" int fd_version(void) { return 100; }

namespace {

__device__ __forceinline__ float act_fn(float v, int act) {
    if (act == 1) return fd_silu(v);
    if (act == 2) return 0.5f * v * (1.f + erff(v * 0.70710678118654752f));   // nn.GELU() (erf form)
    if (act == 3) return fmaxf(v, 0.f);
    return v;
}

// one wave per output feature n, up to 8 rows of x at a time
__global__ __launch_bounds__(256) void linear_kernel(const float *__restrict__ x, const float *__restrict__ w,
                                                    const float *__restrict__ bias, float *__restrict__ out, int M,
                                                    int N, int K, int act, int pre_silu) {
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    const float *wr = w + (int64_t)n * K;
    for (int m0 = 0; m0 < M; m0 += 8) {
        float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int k = lane; k < K; k += 64) {
            const float wv = wr[k];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (m0 + i < M) {
                    float xv = x[(int64_t)(m0 + i) * K + k];
                    if (pre_silu) xv = fd_silu(xv);
                    acc[i] += wv * xv;
                }
            }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            float s = wave_sum(acc[i]);
            if (lane == 0 && m0 + i < M) out[(int64_t)(m0 + i) * N + n] = act_fn(s + (bias ? bias[n] : 0.f), act);
        }
    }
}

__global__ void sinusoidal_kernel(const float *__restrict__ time, float *__restrict__ out, int B, int dim) {
    const int half = dim / 2;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * half) return;
    const int b = i / half, j = i % half;
    const float e = logf(10000.f) / (float)(half - 1);
    const float f = expf((float)j * -e);
    const float a = time[b] * f;
    out[b * dim + j] = sinf(a);
    out[b * dim + half + j] = cosf(a);
}

__global__ __launch_bounds__(256) void softmax_mul_kernel(const float *__restrict__ x, const float *__restrict__ p,
                                                         float *__restrict__ out, int N) {
    __shared__ float red[4];
    const int m = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float mx = -3.4e38f;
    for (int n = tid; n < N; n += 256) mx = fmaxf(mx, x[(int64_t)m * N + n]);
    mx = wave_max(mx);
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float den = 0.f;
    for (int n = tid; n < N; n += 256) den += expf(x[(int64_t)m * N + n] - mx);
    den = wave_sum(den);
    if (lane == 0) red[wave] = den;
    __syncthreads();
    den = red[0] + red[1] + red[2] + red[3];
    for (int n = tid; n < N; n += 256) out[(int64_t)m * N + n] = expf(x[(int64_t)m * N + n] - mx) / den * p[n];
}

__global__ __launch_bounds__(256) void l2norm_rows_kernel(const float *__restrict__ x, float *__restrict__ out, int N,
                                                         float eps) {
    __shared__ float red[4];
    const int m = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float s = 0.f;
    for (int n = tid; n < N; n += 256) { float v = x[(int64_t)m * N + n]; s += v * v; }
    s = wave_sum(s);
    if (lane == 0) red[wave] = s;
    __syncthreads();
    const float nrm = fmaxf(sqrtf(red[0] + red[1] + red[2] + red[3]), eps);
    for (int n = tid; n < N; n += 256) out[(int64_t)m * N + n] = x[(int64_t)m * N + n] / nrm;
}

__global__ void add_kernel(const float *a, const float *b, float *out, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = a[i] + b[i];
}
__global__ void affine_kernel(const float *x, float a, float b, float *out, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = x[i] * a + b;
}
__global__ void axpy_kernel(const float *x, const float *nz, float s, float *out, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = x[i] + s * nz[i];
}

__device__ __forceinline__ float clamp1(float v) { return fd_clamp1(v); }

__global__ void res_predictions_kernel(const float *__restrict__ mo, const float *__restrict__ xt,
                                       const float *__restrict__ xin, const float *__restrict__ ac,
                                       const float *__restrict__ bc, float *pred_res, float *pred_noise,
                                       float *x_start, int64_t npix) {
    const int b = blockIdx.y;
    const float a = ac[b], bb = bc[b];
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < npix; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t j = (int64_t)b * npix + i;
        const float pr = clamp1(mo[j]);
        if (pred_res) pred_res[j] = pr;
        if (pred_noise) pred_noise[j] = (xt[j] - xin[j] - (a - 1.f) * pr) / bb;
        if (x_start) x_start[j] = clamp1(xin[j] - pr);
    }
}

// img / out (and xt / out below) may be the SAME buffer -- the samplers update the image in place -- so
// neither carries __restrict__; every thread reads index i before it writes index i.
__global__ void res_ddim_kernel(const float *__restrict__ mo, const float *img,
                                const float *__restrict__ xin, const float *__restrict__ noise, float alpha,
                                float sigma, int last, float *out, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float pr = clamp1(mo[i]);
        float v;
        if (last) v = clamp1(xin[i] - pr);
        else {
            v = img[i] - alpha * pr;
            if (noise) v += sigma * noise[i];
        }
        out[i] = v;
    }
}

__global__ void res_posterior_kernel(const float *__restrict__ mo, const float *xt,
                                     const float *__restrict__ xin, const float *__restrict__ noise,
                                     const float *__restrict__ coef, float *out,
                                     float *__restrict__ xs_out, int64_t npix) {
    const int b = blockIdx.y;
    const float c1 = coef[b * 4], c2 = coef[b * 4 + 1], c3 = coef[b * 4 + 2];
    const float sd = expf(0.5f * coef[b * 4 + 3]);
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < npix; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t j = (int64_t)b * npix + i;
        const float pr = clamp1(mo[j]);
        const float xs = clamp1(xin[j] - pr);
        float v = c1 * xt[j] + c2 * pr + c3 * xs;
        if (noise) v += sd * noise[j];
        out[j] = v;
        if (xs_out) xs_out[j] = xs;
    }
}

// General form of the three kernels above for the other objectives / the dual-UNet model
// (src/DADiff.py:1168-1207).  mode: 0 o0 = residual; 1 o1 = noise (x_start through
// predict_start_from_xinput_noise); 2 o0 = residual and o1 = noise; 3 o0 = x_0 and o1 = noise.
// par[b] = {alphas_cumsum[t], betas_cumsum[t], one_minus_alphas_cumsum[t], k0, k1, k2, k3, flag};
// step 0: predictions only; 1: DDIM update img - k0*pred_res + k1*noise, or x_start when flag != 0
// (1317-1318, 1344); 2: posterior k0 x_t + k1 pred_res + k2 x_start + exp(k3/2) noise (1142-1151, 1226-1229).
__global__ void res_step_obj_kernel(int mode, int step, const float *__restrict__ o0, const float *__restrict__ o1,
                                    const float *xt, const float *__restrict__ xin,
                                    const float *__restrict__ noise, const float *__restrict__ par, float *pred_res,
                                    float *pred_noise, float *x_start, float *img_out, int64_t npix) {
    const int b = blockIdx.y;
    const float a = par[b * 8], bb = par[b * 8 + 1], oma = par[b * 8 + 2];
    const float k0 = par[b * 8 + 3], k1 = par[b * 8 + 4], k2 = par[b * 8 + 5], k3 = par[b * 8 + 6];
    const bool flag = par[b * 8 + 7] != 0.f;
    const float sd = step == 2 ? expf(0.5f * k3) : 0.f;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < npix; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t j = (int64_t)b * npix + i;
        const float x = xt[j], xi = xin[j];
        float pr, pn, xs;
        if (mode == 0) {
            pr = clamp1(o0[j]);
            pn = (x - xi - (a - 1.f) * pr) / bb;
            xs = clamp1(xi - pr);
        } else if (mode == 1) {
            pn = o1[j];
            xs = clamp1((x - a * xi - bb * pn) / oma);
            pr = clamp1(xi - xs);
        } else if (mode == 2) {
            pr = clamp1(o0[j]);
            pn = o1[j];
            xs = clamp1(x - a * pr - bb * pn);
        } else {
            const float m0 = o0[j];
            pr = clamp1(xi - m0);
            pn = o1[j];
            xs = clamp1(m0);
        }
        if (pred_res) pred_res[j] = pr;
        if (pred_noise) pred_noise[j] = pn;
        if (x_start) x_start[j] = xs;
        if (step == 1) {
            float v = xs;
            if (!flag) {
                v = x - k0 * pr;
                if (noise) v += k1 * noise[j];
            }
            img_out[j] = v;
        } else if (step == 2) {
            float v = k0 * x + k1 * pr + k2 * xs;
            if (noise) v += sd * noise[j];
            img_out[j] = v;
        }
    }
}

unsigned g1(int64_t n) {
    int64_t g = (n + 255) / 256;
    return (unsigned)(g > 4096 ? 4096 : (g < 1 ? 1 : g));
}

}  // namespace

extern "C" int fd_linear(const float *x, const float *w, const float *b, float *out, int M, int N, int K, int act,
                         int pre_silu, void *stream) {
    FD_REQUIRE(x && w && out && M > 0 && N > 0 && K > 0, "fd_linear: bad args");
    hipLaunchKernelGGL(linear_kernel, dim3((N + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, w, b, out, M, N, K, act, pre_silu);
    FD_LAUNCH_OK("fd_linear");
    return FD_OK;
}
extern "C" int fd_sinusoidal(const float *time, float *out, int B, int dim, void *stream) {
    FD_REQUIRE(time && out && dim >= 4 && dim % 2 == 0, "fd_sinusoidal: bad args");
    hipLaunchKernelGGL(sinusoidal_kernel, dim3((B * dim / 2 + 255) / 256), dim3(256), 0, (hipStream_t)stream, time, out, B, dim);
    FD_LAUNCH_OK("fd_sinusoidal");
    return FD_OK;
}
extern "C" int fd_softmax_mul(const float *x, const float *p, float *out, int M, int N, void *stream) {
    FD_REQUIRE(x && p && out, "fd_softmax_mul: null pointer");
    hipLaunchKernelGGL(softmax_mul_kernel, dim3(M), dim3(256), 0, (hipStream_t)stream, x, p, out, N);
    FD_LAUNCH_OK("fd_softmax_mul");
    return FD_OK;
}
extern "C" int fd_l2norm_rows(const float *x, float *out, int M, int N, float eps, void *stream) {
    FD_REQUIRE(x && out, "fd_l2norm_rows: null pointer");
    hipLaunchKernelGGL(l2norm_rows_kernel, dim3(M), dim3(256), 0, (hipStream_t)stream, x, out, N, eps);
    FD_LAUNCH_OK("fd_l2norm_rows");
    return FD_OK;
}
extern "C" int fd_add_f32(const float *a, const float *b, float *out, int64_t n, void *stream) {
    hipLaunchKernelGGL(add_kernel, dim3(g1(n)), dim3(256), 0, (hipStream_t)stream, a, b, out, n);
    FD_LAUNCH_OK("fd_add_f32");
    return FD_OK;
}
extern "C" int fd_affine_f32(const float *x, float a, float b, float *out, int64_t n, void *stream) {
    hipLaunchKernelGGL(affine_kernel, dim3(g1(n)), dim3(256), 0, (hipStream_t)stream, x, a, b, out, n);
    FD_LAUNCH_OK("fd_affine_f32");
    return FD_OK;
}
extern "C" int fd_axpy_f32(const float *x, const float *noise, float s, float *out, int64_t n, void *stream) {
    hipLaunchKernelGGL(axpy_kernel, dim3(g1(n)), dim3(256), 0, (hipStream_t)stream, x, noise, s, out, n);
    FD_LAUNCH_OK("fd_axpy_f32");
    return FD_OK;
}
extern "C" int fd_res_predictions(const float *model_out, const float *x_t, const float *x_in, const float *ac,
                                  const float *bc, float *pred_res, float *pred_noise, float *x_start, int B,
                                  int64_t npix, void *stream) {
    FD_REQUIRE(model_out && x_t && x_in && ac && bc, "fd_res_predictions: null pointer");
    hipLaunchKernelGGL(res_predictions_kernel, dim3(g1(npix), B), dim3(256), 0, (hipStream_t)stream, model_out, x_t,
                       x_in, ac, bc, pred_res, pred_noise, x_start, npix);
    FD_LAUNCH_OK("fd_res_predictions");
    return FD_OK;
}
extern "C" int fd_res_step_obj(int mode, int step, const float *o0, const float *o1, const float *x_t, const float *x_in,
                               const float *noise, const float *par, float *pred_res, float *pred_noise,
                               float *x_start, float *img_out, int B, int64_t npix, void *stream) {
    FD_REQUIRE(mode >= 0 && mode <= 3 && step >= 0 && step <= 2, "fd_res_step_obj: bad mode/step %d/%d", mode, step);
    FD_REQUIRE(x_t && x_in && par, "fd_res_step_obj: null pointer");
    FD_REQUIRE((mode == 1 || o0) && (mode == 0 || o1), "fd_res_step_obj: mode %d needs o0=%p o1=%p", mode, (const void *)o0, (const void *)o1);
    FD_REQUIRE(step == 0 || img_out, "fd_res_step_obj: step %d needs img_out", step);
    hipLaunchKernelGGL(res_step_obj_kernel, dim3(g1(npix), B), dim3(256), 0, (hipStream_t)stream, mode, step, o0, o1, x_t,
                       x_in, noise, par, pred_res, pred_noise, x_start, img_out, npix);
    FD_LAUNCH_OK("fd_res_step_obj");
    return FD_OK;
}
extern "C" int fd_res_ddim_step(const float *model_out, const float *img, const float *x_in, const float *noise,
                                float alpha, float sigma, int last, float *img_out, int64_t n, void *stream) {
    FD_REQUIRE(model_out && img && x_in && img_out, "fd_res_ddim_step: null pointer");
    hipLaunchKernelGGL(res_ddim_kernel, dim3(g1(n)), dim3(256), 0, (hipStream_t)stream, model_out, img, x_in, noise,
                       alpha, sigma, last, img_out, n);
    FD_LAUNCH_OK("fd_res_ddim_step");
    return FD_OK;
}
extern "C" int fd_res_posterior_step(const float *model_out, const float *x_t, const float *x_in, const float *noise,
                                     const float *coef, float *img_out, float *x_start_out, int B, int64_t npix,
                                     void *stream) {
    FD_REQUIRE(model_out && x_t && x_in && coef && img_out, "fd_res_posterior_step: null pointer");
    hipLaunchKernelGGL(res_posterior_kernel, dim3(g1(npix), B), dim3(256), 0, (hipStream_t)stream, model_out, x_t, x_in,
                       noise, coef, img_out, x_start_out, npix);
    FD_LAUNCH_OK("fd_res_posterior_step");
    return FD_OK;
}
