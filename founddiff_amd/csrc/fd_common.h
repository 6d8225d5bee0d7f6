// fd_common.h -- shared device helpers for libfounddiff_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include "../../include/founddiff_hip.h"

// The library's 16-bit storage and MFMA-operand type.  The default build is bfloat16 (libfounddiff_hip.so).  -DFD_HALF_F16
// compiles the SAME sources with IEEE binary16 in its place (libfounddiff_hip_f16.so, the host's precision='fp16'): three more
// mantissa bits in every stored activation and weight at the same bytes, MFMA rate and instruction counts, in exchange for
// binary16's range (largest finite value 65504, nothing below 6e-8) -- DESIGN.md section 5.  The type keeps the name `bf16`, and
// dtype code FD_BF16 means "the 16-bit type of this build", in both; fd_half_format() tells a caller which build it loaded.
#ifdef FD_HALF_F16
typedef _Float16 bf16;
#define FD_MFMA16 __builtin_amdgcn_mfma_f32_16x16x32_f16
#define FD_MFMA16_ASM "v_mfma_f32_16x16x32_f16"
#define FD_H_ONES 0x3C003C00u          // the pair (1, 1)
#else
typedef __bf16 bf16;
#define FD_MFMA16 __builtin_amdgcn_mfma_f32_16x16x32_bf16
#define FD_MFMA16_ASM "v_mfma_f32_16x16x32_bf16"
#define FD_H_ONES 0x3F803F80u
#endif
typedef __attribute__((ext_vector_type(8))) bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) bf16 bf16x2;
// the low / high 16-bit element of a dword as a float (bfloat16: a shift / a mask; binary16: v_cvt_f32_f16, SDWA for the high one)
__device__ __forceinline__ float fd_h_lo(uint32_t w) {
#ifdef FD_HALF_F16
    return (float)__builtin_bit_cast(_Float16, (uint16_t)w);
#else
    return __builtin_bit_cast(float, w << 16);
#endif
}
__device__ __forceinline__ float fd_h_hi(uint32_t w) {
#ifdef FD_HALF_F16
    return (float)__builtin_bit_cast(_Float16, (uint16_t)(w >> 16));
#else
    return __builtin_bit_cast(float, w & 0xffff0000u);
#endif
}
// fp32 -> the 16-bit type, round to nearest even, for STORED values.  binary16 build: the fp32 value is pinned before it is
// converted.  hipcc folds fptrunc(fma(..)) into v_fma_mixlo_f16 where it sees one -- a single rounding to binary16 instead of
// fp32's and then binary16's -- and it saw one in the persistent pointwise GEMM's gated-residual epilogue and not in the generic
// tile's, which has the same source lines: 6e-5 of a layer's elements came out one binary16 ulp apart depending on WHICH of the
// two kernels the batch size had selected (tools/probes/pwg_vs_igemm.py), and a slice's bits depended on its batch.  The empty
// asm costs no instruction.
__device__ __forceinline__ bf16 fd_cvt_h(float v) {
#ifdef FD_HALF_F16
    asm("" : "+v"(v));
#endif
    return (bf16)v;
}
// c + a.x b.x + a.y b.y on a pair of 16-bit elements, fp32 accumulation (v_dot2_f32_bf16 / v_dot2_f32_f16)
__device__ __forceinline__ float fd_dot2(bf16x2 a, bf16x2 b, float c) {
#ifdef FD_HALF_F16
    return __builtin_amdgcn_fdot2(a, b, c, false);
#else
    return __builtin_amdgcn_fdot2_f32_bf16(a, b, c, false);
#endif
}
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

// clamp(v, -1, 1) of the samplers (src/DADiff.py:1206, 1318: torch.clamp), NaN in -> NaN out like torch.clamp.  fminf / fmaxf alone
// return the other operand for a NaN: a forward that went non-finite (binary16 overflow in the fp16 build, a broken checkpoint in
// any build) came back as a plausible image of -1s.
__device__ __forceinline__ float fd_clamp1(float v) { return v != v ? v : fminf(fmaxf(v, -1.f), 1.f); }

template <typename T> struct TT;
template <> struct TT<float> { static constexpr int CH = 4; };   // elements per 16-byte chunk
template <> struct TT<bf16> { static constexpr int CH = 8; };

// x * sigmoid(x) with the hardware reciprocal (v_rcp_f32, ~1 ulp) instead of an IEEE division
// (v_div_scale/fmas/fixup: ~10 instructions and 4 live registers per element).
__device__ __forceinline__ float fd_silu(float x) {
    return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}
// the same on a PAIR (v_pk_mul / v_pk_add around the four transcendentals): 11 instead of 14 issue slots per two
// elements, bit-identical to fd_silu per element (packed fp32 ops round like the scalar ones)
__device__ __forceinline__ f32x2 fd_silu2(f32x2 x) {
    const f32x2 t = x * -1.4426950408889634f;
    const f32x2 e = f32x2{__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y)} + 1.0f;
    return x * f32x2{__builtin_amdgcn_rcpf(e.x), __builtin_amdgcn_rcpf(e.y)};
}
__device__ __forceinline__ void fd_silu8(float (&v)[8]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const f32x2 r = fd_silu2(f32x2{v[2 * j], v[2 * j + 1]});
        v[2 * j] = r.x;
        v[2 * j + 1] = r.y;
    }
}
// val += silu((h - mean) * rstd * gamma + beta) on 8 channels: the GroupNorm + SiLU + residual tail of a ResnetBlock
// (src/DADiff.py:213-229, 397-430) in packed fp32
__device__ __forceinline__ void fd_gn_silu_add8(float (&val)[8], const float (&h)[8], float mean, float rstd,
                                                const float (&gamma)[8], const float (&beta)[8]) {
    const float nm = -mean * rstd;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const f32x2 t = f32x2{h[2 * j], h[2 * j + 1]} * rstd + nm;
        const f32x2 r = fd_silu2(t * f32x2{gamma[2 * j], gamma[2 * j + 1]} + f32x2{beta[2 * j], beta[2 * j + 1]});
        val[2 * j] += r.x;
        val[2 * j + 1] += r.y;
    }
}
__device__ __forceinline__ float fd_softplus(float x) { return x > 20.0f ? x : log1pf(__expf(x)); }
// softplus without libm: for x < -4 the series e - e^2/2 + e^3/3 of log1p(e), e = exp(x) < 0.0184
// (truncation < 3e-8 absolute, ~1.5e-6 relative); else log(1 + e) with 1 + e >= 1.018 so the
// rounding of the sum costs <= 3e-6 relative.  v_exp_f32 / v_log_f32 are the base-2 forms.
__device__ __forceinline__ float fd_softplus_fast(float x) {
    // (branch-free: with an early return for x > 20 the unrolled scan steps became control flow the register allocator
    //  could not keep inside an occupancy step: 450-820 bytes of scratch in the fp32 phase-A kernels)
    const float e = __builtin_amdgcn_exp2f(x * 1.4426950408889634f);      // (x > 20: inf at worst, discarded below; NaN stays NaN)
    const float series = e * (1.0f + e * (-0.5f + e * 0.33333334f));
    const float lg = __builtin_amdgcn_logf(1.0f + e) * 0.6931471805599453f;
    const float r = x < -4.0f ? series : lg;
    return x > 20.0f ? x : r;
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
    for (int i = 0; i < 8; ++i) a[i] = fd_cvt_h(v[i]);
    *(bf16x8 *)p = a;
}
__device__ __forceinline__ float ld1(const float *p) { return *p; }
__device__ __forceinline__ float ld1(const bf16 *p) { return (float)*p; }
__device__ __forceinline__ void st1(float *p, float v) { *p = v; }
__device__ __forceinline__ void st1(bf16 *p, float v) { *p = fd_cvt_h(v); }

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

// v + (v of the lane the DPP control selects): one v_add_f32_dpp.  0xB1 / 0x4E = lane ^ 1 / lane ^ 2 inside a quad,
// 0x141 = row_half_mirror (lane 7 - i of each 8), 0x140 = row_mirror (lane 15 - i of each 16): after the two quad steps
// every lane of a quad holds the quad's sum, so the mirrors pair whole quads -- sums over 8 / 16 neighbouring lanes
// without ds_bpermute (an LDS round trip per step) and with the same bits in every lane of the group.
template <int CTRL>
__device__ __forceinline__ float fd_dpp_add(float v) {
    return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
template <int LANES>
__device__ __forceinline__ float fd_group_sum(float v) {
    static_assert(LANES == 4 || LANES == 8 || LANES == 16, "fd_group_sum: 4, 8 or 16 neighbouring lanes");
    v = fd_dpp_add<0xB1>(v);
    v = fd_dpp_add<0x4E>(v);
    if (LANES >= 8) v = fd_dpp_add<0x141>(v);
    if (LANES >= 16) v = fd_dpp_add<0x140>(v);
    return v;
}

__device__ __forceinline__ uint32_t fd_pack_bf16(f32x2 v) {
#ifdef FD_HALF_F16
    float a = v.x, b = v.y;                      // (fd_cvt_h: the fp32 values are final before the conversion)
    asm("" : "+v"(a), "+v"(b));
    v = f32x2{a, b};
#endif
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ f32x2 fd_unpack_bf16(uint32_t w) {
    return f32x2{fd_h_lo(w), fd_h_hi(w)};
}

// LayerNorm + modulate of one 16-byte chunk (8 bf16 channels) of a pixel whose C = 8 * LANES channels sit in LANES
// neighbouring lanes (src/DADiff.py:450-451, 477-488: x_n = LN(x) * g + b with g = gamma (1 + scale), b = beta (1 + scale)
// + shift folded by the caller).  One pass (sum and sum of squares, fp32), packed fp32 arithmetic, DPP group sums:
// ~40 issue slots per chunk (the two-pass scalar form with ds_bpermute shuffles took ~120 and six LDS round trips).
// Why one pass is enough HERE (and only here: ln_rows_kernel, which also serves the fp32 modes, centres its rows): the
// inputs are bf16.  var = E[x^2] - mean^2 in fp32 carries a relative error of ~2^-24 (1 + (mean / std)^2) (the products of
// bf16 values are exact in fp32); a bf16 row cannot hold |mean| / std above ~2^8 at all (the spacing of bf16 at |mean| is
// |mean| / 2^8: beyond that ratio the row is its own quantisation noise), so the error of var stays below 2^-24 2^16 =
// 0.4 % at that extreme and below 1e-4 for |mean| / std <= 40.  tests/test_gpu_round5.py::test_ln_rows_with_common_offset
// pins both forms against F.layer_norm on rows with a large common offset.
template <int LANES>
__device__ __forceinline__ u32x4 fd_ln_mod_chunk(u32x4 raw, const f32x2 (&g)[4], const f32x2 (&b)[4], float eps) {
    const uint32_t rw[4] = {raw.x, raw.y, raw.z, raw.w};
    f32x2 x[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) x[j] = fd_unpack_bf16(rw[j]);
    const f32x2 s = (x[0] + x[1]) + (x[2] + x[3]);
    f32x2 q = x[0] * x[0];
    q = x[1] * x[1] + q;
    q = x[2] * x[2] + q;
    q = x[3] * x[3] + q;
    const float s1 = fd_group_sum<LANES>(s.x + s.y), s2 = fd_group_sum<LANES>(q.x + q.y);
    constexpr float inv = 1.f / (8 * LANES);
    const float mean = s1 * inv;
    const float var = fmaxf(s2 * inv - mean * mean, 0.f);
    const float rstd = __builtin_amdgcn_rsqf(var + eps), nm = -mean * rstd;
    uint32_t o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const f32x2 t = x[j] * rstd + nm;
        o[j] = fd_pack_bf16(t * g[j] + b[j]);
    }
    return u32x4{o[0], o[1], o[2], o[3]};
}

// Accumulators written by inline-asm MFMAs: the compiler's hazard recognizer does not see into asm, so the last MFMAs'
// results need wait states before the first VALU read.  A bare `asm volatile("s_nop ..." ::: "memory")` only orders MEMORY
// operations: a register-only consumer (acc + bias) has no dependence on it and may legally be scheduled above the nops.
// These macros make every accumulator an in/out operand of the nop block (or of an empty volatile asm that follows it --
// volatile asm statements keep their order): any read of an accumulator now depends on the wait.  (tools/kres.py checks
// the ISA: no VALU read of an accumulator between the last v_mfma of a loop and the s_nop block.)
#define FD_TIE2(a) "+v"((a)[0]), "+v"((a)[1])
#define FD_TIE4(a) FD_TIE2(a), "+v"((a)[2]), "+v"((a)[3])
#define FD_TIE8(a) FD_TIE4(a), "+v"((a)[4]), "+v"((a)[5]), "+v"((a)[6]), "+v"((a)[7])
#define FD_MFMA_ASM_DRAIN "s_nop 15\n\ts_nop 15\n\ts_nop 15"

// ---- Development switches: ONE table, read from the environment ONCE, absent from the release build.
// Until round 6 the dispatch code held 29 getenv() calls behind function-local statics: the behaviour of a library that
// include/founddiff_hip.h calls stateless depended on the environment at first use.  Now: X(name, kind, default) below is the
// whole list (environment variable FD_<name>); fd_dev(FD_DEV_<name>) is the only accessor; under -DFD_RELEASE (what founddiff_amd.build.build() ships) it is a
// compile-time constant -- the default -- and no environment variable reaches the library.  A development build
// (FOUNDDIFF_DEV_BUILD=1 python -m founddiff_amd.build) reads the variables once, in fd_dev_options() / the first fd_dev().
// kind F: flag (set = 1); kind I: integer value.  What each one does is documented where it is used.
#define FD_DEV_SWITCHES(X)                                                                            \
    X(CONV_KID, I, -1)          /* fd_conv.hip: force an implicit-GEMM tile for stat-free 1x1 layers */ \
    X(CONV_BIG_TILE, F, 0)      /* fd_conv.hip: the 128x256 tile below its workgroup-count threshold */ \
    X(NO_CONV3_SPLIT, F, 0)     /* fd_conv3x3.hip: fp32s 3x3 on the generic split implicit GEMM */     \
    X(NO_CONV3_UP2X, F, 0)      /* fd_conv3x3.hip: up-sampling 3x3 in its 9-tap form */                \
    X(CONV3_UP_CPW, I, 0)       /* fd_conv3x3.hip: parity classes per workgroup of the four-2x2 form */ \
    X(CONV3_TH8, F, 0)          /* fd_conv3x3.hip: 8-row tiles for Cout <= 64 too */                   \
    X(CONV_TPW, I, 0)           /* fd_conv3x3.hip: tiles per workgroup */                              \
    X(NO_CONV3_RW, F, 0)        /* fd_conv3x3_rw.hip: off */                                           \
    X(CONV3_RW_TPW, I, 0)       /* fd_conv3x3_rw.hip: tiles per workgroup */                           \
    X(NO_DOWNFUSE, F, 0)        /* fd_downfuse.hip: off */                                             \
    X(DOWN_TPW, I, 0)           /* fd_downfuse.hip: tiles per workgroup */                             \
    X(ROWS_PER_CU_MAX, I, 0)    /* fd_gemm_rows.hip: persistent workgroups per CU, upper limit */      \
    X(ROWS_PER_CU, I, 0)        /* fd_gemm_rows.hip: persistent workgroups per CU, cap */              \
    X(ROWS32_PER_CU, I, 3)         /* fd_gemm_rows32.hip: persistent workgroups per CU, upper limit */   \
    X(ZRE_NOPF, F, 0)           /* fd_gemm_rows.hip: z-recompute out_proj without the next-tile prefetch */ \
    X(NO_PWDW128, F, 0)         /* fd_pwdw.hip: the C = 128 fused in_proj kernel off */                \
    X(PWDW_MINPIX, I, 32768)    /* fd_pwdw.hip: smallest image the fused kernels take */               \
    X(PWDW_TPW, I, 0)           /* fd_pwdw.hip: tiles per workgroup */                                 \
    X(NO_PWDW_PROJ, F, 0)       /* fd_pwdw.hip: v-recompute project_out kernel off */                  \
    X(GRAM_TPW, I, 0)           /* fd_pwdw.hip: tiles per workgroup of the Gram form (changes the partial-sum order) */ \
    X(NO_GRAM_FUSE, F, 0)       /* fd_pwdw.hip: Gram-fused qkv kernel off */                           \
    X(NO_DWGRAM, F, 0)          /* fd_pwdw.hip: dwconv_gram_kernel off */                              \
    X(NO_PWGEMM, F, 0)          /* fd_pwgemm.hip: the generic 256x256 tile instead */                  \
    X(SCAN_NO_SEQ, F, 0)        /* fd_scan.hip: chunked form for short sequences too */                \
    X(SCAN_NO_CPL2, F, 0)       /* fd_scan.hip: one channel per lane at level 0 */                     \
    X(SCAN_CL, I, 0)            /* fd_scan.hip: chunk length */                                        \
    X(PAD_CONV3, I, 0)          /* extra dynamic LDS (KiB) per workgroup: a cap on workgroups per CU (tools/probes/corun.py) */ \
    X(PAD_PWDW, I, 0)                                                                              \
    X(PAD_SCAN, I, 0)
enum fd_dev_id {
#define FD_DEV_ENUM(name, kind, dflt) FD_DEV_##name,
    FD_DEV_SWITCHES(FD_DEV_ENUM)
#undef FD_DEV_ENUM
    FD_DEV_COUNT
};
#ifdef FD_RELEASE
static constexpr int fd_dev_defaults[FD_DEV_COUNT] = {
#define FD_DEV_DFLT(name, kind, dflt) dflt,
    FD_DEV_SWITCHES(FD_DEV_DFLT)
#undef FD_DEV_DFLT
};
static constexpr int fd_dev(int id) { return fd_dev_defaults[id]; }
#else
int fd_dev(int id);            // fd_small.hip: the table, filled from the environment on first use
#endif
// extra dynamic LDS per workgroup of a kernel family (FD_PAD_<family>, KiB); 0 in the release build
static inline size_t fd_occ_pad(int id) { return (size_t)fd_dev(id) * 1024; }

static inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }
