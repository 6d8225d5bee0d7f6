// fd_common.h -- shared device helpers for libfounddiff_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/founddiff_hip.h"

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

void fd_set_error(const char *fmt, ...);

#define FD_REQUIRE(cond, ...)            \
    do {                                 \
        if (!(cond)) {                   \
            fd_set_error(__VA_ARGS__);   \
            return FD_ERR_ARG;           \
        }                                \
    } while (0)

#define FD_LAUNCH_OK(name)                                                     \
    do {                                                                       \
        hipError_t e__ = hipGetLastError();                                    \
        if (e__ != hipSuccess) {                                               \
            fd_set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
            return FD_ERR_LAUNCH;                                              \
        }                                                                      \
    } while (0)

// bf16-mode softplus: log(1 + e) directly.  For e < 0.018 the rounding of 1 + e costs up to 4e-4 relative
// in a dt that is itself < 0.018 -- far below the bf16 resolution of the activations it multiplies.
__device__ __forceinline__ float fd_softplus_bf16(float x) {
    const float e = __builtin_amdgcn_exp2f(x * 1.4426950408889634f);
    const float lg = __builtin_amdgcn_logf(1.0f + e) * 0.6931471805599453f;
    return x > 20.0f ? x : lg;
}

template <typename T> struct TT;
template <> struct TT<float> { static constexpr int CH = 4; };   // elements per 16-byte chunk
template <> struct TT<bf16> { static constexpr int CH = 8; };

// x * sigmoid(x) with the hardware reciprocal (v_rcp_f32, ~1 ulp) instead of an IEEE division
// (v_div_scale/fmas/fixup: ~10 instructions and 4 live registers per element).
__device__ __forceinline__ float fd_silu(float x) {
    return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}
__device__ __forceinline__ float fd_softplus(float x) { return x > 20.0f ? x : log1pf(__expf(x)); }
// softplus without libm: for x < -4 the series e - e^2/2 + e^3/3 of log1p(e), e = exp(x) < 0.0184
// (truncation < 3e-8 absolute, ~1.5e-6 relative); else log(1 + e) with 1 + e >= 1.018 so the
// rounding of the sum costs <= 3e-6 relative.  v_exp_f32 / v_log_f32 are the base-2 forms.
__device__ __forceinline__ float fd_softplus_fast(float x) {
    if (x > 20.0f) return x;
    const float e = __builtin_amdgcn_exp2f(x * 1.4426950408889634f);
    const float series = e * (1.0f + e * (-0.5f + e * 0.33333334f));
    const float lg = __builtin_amdgcn_logf(1.0f + e) * 0.6931471805599453f;
    return x < -4.0f ? series : lg;
}

// 8 consecutive elements <-> 8 floats (16-byte aligned for bf16, 32-byte span for f32)
__device__ __forceinline__ void load8(const float *p, float v[8]) {
    f32x4 a = *(const f32x4 *)p, b = *(const f32x4 *)(p + 4);
    v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3];
    v[4] = b[0]; v[5] = b[1]; v[6] = b[2]; v[7] = b[3];
}
__device__ __forceinline__ void load8(const bf16 *p, float v[8]) {
    bf16x8 a = *(const bf16x8 *)p;
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (float)a[i];
}
__device__ __forceinline__ void store8(float *p, const float v[8]) {
    f32x4 a = {v[0], v[1], v[2], v[3]}, b = {v[4], v[5], v[6], v[7]};
    *(f32x4 *)p = a;
    *(f32x4 *)(p + 4) = b;
}
__device__ __forceinline__ void store8(bf16 *p, const float v[8]) {
    bf16x8 a;
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = (bf16)v[i];
    *(bf16x8 *)p = a;
}
__device__ __forceinline__ float ld1(const float *p) { return *p; }
__device__ __forceinline__ float ld1(const bf16 *p) { return (float)*p; }
__device__ __forceinline__ void st1(float *p, float v) { *p = v; }
__device__ __forceinline__ void st1(bf16 *p, float v) { *p = (bf16)v; }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

static inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }
