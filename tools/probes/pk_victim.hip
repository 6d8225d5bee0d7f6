// Probe (round 6): which packed-fp32 instruction pattern goes wrong when an MFMA kernel shares the SIMD?
// Every lane runs the same arithmetic twice -- once on float pairs (v_pk_*_f32; this file is compiled WITH packed-fp32
// instructions) and once on plain floats -- and counts the iterations in which the two disagree, per component.
// Alone the two are bit-identical (packed fp32 ops round like the scalar ones).  hipcc --offload-arch=gfx950 -O3
// -fno-slp-vectorize -shared -fPIC tools/probes/pk_victim.hip -o tools/probes/bin/libpkvictim.so
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(2))) float f32x2;

template <int MODE>
__global__ __launch_bounds__(256) void pk_victim(uint32_t *out, int iters, float a0, float b0, const float *grow) {
    const float *grow2 = grow + 128;
    const int gid = blockIdx.x * 256 + threadIdx.x;
    f32x2 x = {0.5f + 1e-3f * (gid & 1023), 0.25f + 2e-3f * (gid & 511)};
    float sx = x.x, sy = x.y;
    f32x2 acc = {0.f, 0.f};
    float sa = 0.f, sb = 0.f;
    const f32x2 a = {a0, a0 * 0.999f}, b = {b0, b0 * 1.001f};
    uint32_t bad_lo = 0, bad_hi = 0;
    __shared__ __attribute__((aligned(16))) float srow[128];
    if (threadIdx.x < 128) srow[threadIdx.x] = grow[threadIdx.x];      // the LDS row = a copy of a global array: the reference path reads the global one
    __syncthreads();
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {                       // dependent v_pk_fma_f32 chain
            x = x * a + b;
            sx = sx * a.x + b.x;
            sy = sy * a.y + b.y;
        } else if (MODE == 1) {                // v_pk_mul_f32, low / high result consumed by plain VALU adds
            const f32x2 t = x * a;
            acc.x += t.x; acc.y += t.y;
            x = x + b * 1e-3f;
            const float tx = sx * a.x, ty = sy * a.y;
            sa += tx; sb += ty;
            sx = sx + b.x * 1e-3f; sy = sy + b.y * 1e-3f;
        } else if (MODE == 2) {                // v_exp_f32 x 2 -> v_pk_mul_f32 -> v_pk_fma_f32 (the scan step's shape)
            const f32x2 e = {__builtin_amdgcn_exp2f(-x.x), __builtin_amdgcn_exp2f(-x.y)};
            acc = e * acc + x * b;
            x = x * a + b * 1e-2f;
            const float ex = __builtin_amdgcn_exp2f(-sx), ey = __builtin_amdgcn_exp2f(-sy);
            sa = ex * sa + sx * b.x; sb = ey * sb + sy * b.y;
            sx = sx * a.x + b.x * 1e-2f; sy = sy * a.y + b.y * 1e-2f;
        } else if (MODE >= 4) {                // the scan step's shape: an LDS broadcast row (ds_read_b128, same address in all lanes) feeds the
                                               // arithmetic; the REFERENCE path reads the same values from global memory (no LDS)
            const int ri = (i & 31) * 4;
            float4 r;
            if (MODE == 8) {                   // 8: the row from GLOBAL memory (a second copy; asm: a real global_load per iteration)
                typedef __attribute__((ext_vector_type(4))) float f4;
                f4 g;
                asm volatile("global_load_dwordx4 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(g) : "v"(grow2 + ri) : "memory");
                r.x = g[0]; r.y = g[1]; r.z = g[2]; r.w = g[3];
            } else r = *(const float4 *)&srow[ri];
            const float4 cg = *(const float4 *)&grow[ri];
            const float c0 = cg.x, c1 = cg.y, c2 = cg.z, c3 = cg.w;
            if (MODE == 6) asm volatile("v_mov_b32 %0, %0\n\tv_mov_b32 %1, %1\n\tv_mov_b32 %2, %2\n\tv_mov_b32 %3, %3" : "+v"(r.x), "+v"(r.y), "+v"(r.z), "+v"(r.w));
            if (MODE == 7) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_nop 7" : "+v"(r.x), "+v"(r.y), "+v"(r.z), "+v"(r.w));
            if (MODE == 5) {                   // 5: plain (non-packed) arithmetic on the LDS values
                float ux = x.x, uy = x.y, va = acc.x, vb = acc.y;
                const float tx = a.x * r.x, ty = a.y * r.x;
                const float ex = __builtin_amdgcn_exp2f(tx * 0.01f - 1.f), ey = __builtin_amdgcn_exp2f(ty * 0.01f - 1.f);
                va = ex * va + ux * r.y; vb = ey * vb + uy * r.y;
                ux = ux * r.z + b.x * r.w; uy = uy * r.z + b.y * r.w;
                asm volatile("" : "+v"(ux), "+v"(uy), "+v"(va), "+v"(vb));
                x = f32x2{ux, uy}; acc = f32x2{va, vb};
            } else {                           // 4, 6, 7: packed arithmetic on the LDS values
                const f32x2 t = a * r.x;
                const f32x2 e = {__builtin_amdgcn_exp2f(t.x * 0.01f - 1.f), __builtin_amdgcn_exp2f(t.y * 0.01f - 1.f)};
                acc = e * acc + x * r.y;
                x = x * r.z + b * r.w;
            }
            const float tx = a.x * c0, ty = a.y * c0;
            const float ex = __builtin_amdgcn_exp2f(tx * 0.01f - 1.f), ey = __builtin_amdgcn_exp2f(ty * 0.01f - 1.f);
            sa = ex * sa + sx * c1; sb = ey * sb + sy * c1;
            sx = sx * c2 + b.x * c3; sy = sy * c2 + b.y * c3;
        } else {                               // v_pk_add_f32 only
            x = x + b;
            sx = sx + b.x; sy = sy + b.y;
        }
        // keep the two paths apart (no CSE between them) and compare as bits
        asm volatile("" : "+v"(sx), "+v"(sy), "+v"(sa), "+v"(sb));
        // (scalar copies pinned through an empty asm: hipcc 7.2 compiled __builtin_bit_cast(uint32_t, x.y) to a read of ELEMENT 0 --
        //  the first version of this probe "found" a mismatch in every lane and iteration that way)
        float px = x.x, py = x.y, qx = acc.x, qy = acc.y;
        asm volatile("" : "+v"(px), "+v"(py), "+v"(qx), "+v"(qy));
        const bool l = __builtin_bit_cast(uint32_t, px) != __builtin_bit_cast(uint32_t, sx) ||
                       __builtin_bit_cast(uint32_t, qx) != __builtin_bit_cast(uint32_t, sa);
        const bool h = __builtin_bit_cast(uint32_t, py) != __builtin_bit_cast(uint32_t, sy) ||
                       __builtin_bit_cast(uint32_t, qy) != __builtin_bit_cast(uint32_t, sb);
        bad_lo += l; bad_hi += h;
        if (l) { x.x = sx; acc.x = sa; }       // resynchronise: count events, not their propagation
        if (h) { x.y = sy; acc.y = sb; }
        if (MODE == 0 && (i & 63) == 63) { x = f32x2{0.5f, 0.25f} + x * 1e-6f; sx = 0.5f + sx * 1e-6f; sy = 0.25f + sy * 1e-6f; }
    }
    out[2 * gid] = bad_lo;
    out[2 * gid + 1] = bad_hi;
}

// An aggressor of our own: bf16 MFMAs on registers (WHAT = 1), LDS traffic (2), both (3); 64 threads per workgroup and few
// registers, so that its waves share SIMDs with the victim's.
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
template <int WHAT>
__global__ __launch_bounds__(64) void pk_aggressor(float *out, int iters) {
    __shared__ __attribute__((aligned(16))) float sl[64 * 4 * 4];
    const int l = threadIdx.x;
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(0.01f * (l + e)); b[e] = (__bf16)(0.02f * (l - e)); }
    f32x4 c0 = {0, 0, 0, 0}, c1 = {0, 0, 0, 0}, c2 = {0, 0, 0, 0}, c3 = {0, 0, 0, 0};
    f32x4 v = {1.f * l, 2.f, 3.f, 4.f};
    for (int i = 0; i < iters; ++i) {
        if (WHAT & 1) {
            c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c3, 0, 0, 0);
        }
        if (WHAT & 2) {
            *(f32x4 *)&sl[((l + i) & 63) * 16 + 4 * (i & 3)] = v;
            v += *(const f32x4 *)&sl[((l * 7 + i) & 63) * 16 + 4 * ((i + 1) & 3)];
        }
    }
    const f32x4 r = c0 + c1 + c2 + c3 + v;
    if (r[0] == 123.456f) out[l] = r[1] + r[2] + r[3];
}

extern "C" int pk_aggressor_launch(int what, float *out, int iters, int blocks, void *stream) {
    hipStream_t s = (hipStream_t)stream;
    if (what == 1) hipLaunchKernelGGL(pk_aggressor<1>, dim3(blocks), dim3(64), 0, s, out, iters);
    else if (what == 2) hipLaunchKernelGGL(pk_aggressor<2>, dim3(blocks), dim3(64), 0, s, out, iters);
    else hipLaunchKernelGGL(pk_aggressor<3>, dim3(blocks), dim3(64), 0, s, out, iters);
    return (int)hipGetLastError();
}

extern "C" int pk_victim_launch(int mode, uint32_t *out, int iters, int blocks, const float *grow, void *stream) {
    hipStream_t s = (hipStream_t)stream;
    switch (mode) {
    case 0: hipLaunchKernelGGL(pk_victim<0>, dim3(blocks), dim3(256), 0, s, out, iters, 0.97f, 0.013f, grow); break;
    case 1: hipLaunchKernelGGL(pk_victim<1>, dim3(blocks), dim3(256), 0, s, out, iters, 0.97f, 0.013f, grow); break;
    case 2: hipLaunchKernelGGL(pk_victim<2>, dim3(blocks), dim3(256), 0, s, out, iters, 0.97f, 0.013f, grow); break;
    case 4: hipLaunchKernelGGL(pk_victim<4>, dim3(blocks), dim3(256), 0, s, out, iters, 0.97f, 0.013f, grow); break;
    case 5: hipLaunchKernelGGL(pk_victim<5>, dim3(blocks), dim3(256), 0, s, out, iters, 0.97f, 0.013f, grow); break;
    case 6: hipLaunchKernelGGL(pk_victim<6>, dim3(blocks), dim3(256), 0, s, out, iters, 0.97f, 0.013f, grow); break;
    case 7: hipLaunchKernelGGL(pk_victim<7>, dim3(blocks), dim3(256), 0, s, out, iters, 0.97f, 0.013f, grow); break;
    case 8: hipLaunchKernelGGL(pk_victim<8>, dim3(blocks), dim3(256), 0, s, out, iters, 0.97f, 0.013f, grow); break;
    default: hipLaunchKernelGGL(pk_victim<3>, dim3(blocks), dim3(256), 0, s, out, iters, 0.97f, 0.013f, grow); break;
    }
    return (int)hipGetLastError();
}
