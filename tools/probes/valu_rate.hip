// Probe (round 5): what does a gfx950 SIMD actually issue per clock?  The scan / fused LN -> 1x1 -> depthwise kernels are
// "VALU-bound" against a model of ONE wave64 VALU instruction per 4 cycles per SIMD (16 lanes / clk) with packed fp32
// counting double and transcendentals half rate.  MI355X_MICROARCH.md says a wave64 v_fma_f32 takes 2 cycles of a SIMD once
// more than one wave issues (4 for a wave alone) and v_exp_f32 8.  This probe measures, per instruction kind and per
// waves-per-SIMD (1, 2, 4, 8), the cycles one SIMD spends per wave64 instruction (s_memtime around an unrolled block of
// independent instructions on random data), and the same for MIXES inside a wave and across the waves of a SIMD:
// does a transcendental of one wave overlap the FMAs of another?
//   build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 -o /tmp/vr tools/probes/valu_rate.hip && /tmp/vr
#pragma clang diagnostic ignored "-Wunused-value"
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>
typedef float f32x2 __attribute__((ext_vector_type(2)));

enum Kind { FMA = 0, PKFMA, EXP, LOG, PKMUL, PKADD, MIX_EXP_2FMA, MIX_EXP_4FMA, MIX_EXP_2PK, CVTPK, PKFMA16, MULLO, SPLIT_EXP_FMA, SPLIT_EXP_PK,
            FMA_DEP, MIX_SCAN, CVTPK_F16, CVTPKRTZ, CVT_F32_F16, CVT_F32_F16_SDWA, FMA_MIX, FMA_MIXLO, LSHL, NKIND };
static const char *KN[NKIND] = {"v_fma_f32", "v_pk_fma_f32", "v_exp_f32", "v_log_f32", "v_pk_mul_f32", "v_pk_add_f32",
                                "in-wave 1 exp : 2 fma", "in-wave 1 exp : 4 fma", "in-wave 1 exp : 2 pk_fma", "v_cvt_pk_bf16_f32",
                                "v_pk_fma_f16", "v_mul_lo_u32", "across waves: even waves exp, odd waves fma",
                                "across waves: even waves exp, odd waves pk_fma", "v_fma_f32 one dependent chain",
                                "scan step mix (12 exp/log + 26 pk + 6 plain per pair)",
                                "v_cvt_pk_f16_f32", "v_cvt_pkrtz_f16_f32", "v_cvt_f32_f16", "v_cvt_f32_f16 sdwa (high half)",
                                "v_fma_mix_f32 (f16 src2)", "v_fma_mixlo_f16", "v_lshlrev_b32"};

constexpr int NACC = 8, INNER = 8, ITERS = 512;      // instructions per wave = ITERS * INNER * NACC (per kind unit)

template <int K>
__global__ __launch_bounds__(256) void probe(const float *in, float *out, uint64_t *cyc, int *ninstr) {
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    const int wave = threadIdx.x >> 6;
    float a[NACC], q0[NACC], q1[NACC];
    f32x2 p[NACC], r[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) {
        a[i] = in[(tid * NACC + i) & 4095];
        p[i] = f32x2{in[(tid * 2 * NACC + 2 * i) & 4095], in[(tid * 2 * NACC + 2 * i + 1) & 4095]};
        r[i] = p[i] * 0.5f; q0[i] = p[i].x; q1[i] = p[i].y;
    }
    const float b = in[tid & 1023] * 0.001f + 0.999f, c = in[(tid + 7) & 1023] * 1e-3f;
    const f32x2 b2 = {b, b * 0.9999f}, c2 = {c, -c};
    uint32_t ua = __builtin_bit_cast(uint32_t, a[0]) | 1u, ub = __builtin_bit_cast(uint32_t, a[1]) | 1u;
    int n = 0;
    __syncthreads();
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int j = 0; j < INNER; ++j) {
#pragma unroll
            for (int i = 0; i < NACC; ++i) {
                if constexpr (K == FMA) { asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c)); }
                else if constexpr (K == FMA_DEP) { asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[0]) : "v"(b), "v"(c)); }
                else if constexpr (K == PKFMA) { asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(b2), "v"(c2)); }
                else if constexpr (K == PKMUL) { asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(b2)); }
                else if constexpr (K == PKADD) { asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(c2)); }
                else if constexpr (K == EXP) { asm volatile("v_exp_f32 %0, %0" : "+v"(a[i])); }
                else if constexpr (K == LOG) { asm volatile("v_log_f32 %0, %0" : "+v"(a[i])); }
                else if constexpr (K == CVTPK) { asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b)); }
                else if constexpr (K == CVTPK_F16) { asm volatile("v_cvt_pk_f16_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b)); }
                else if constexpr (K == CVTPKRTZ) { asm volatile("v_cvt_pkrtz_f16_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b)); }
                else if constexpr (K == CVT_F32_F16) { asm volatile("v_cvt_f32_f16 %0, %0" : "+v"(a[i])); }
                else if constexpr (K == CVT_F32_F16_SDWA) { asm volatile("v_cvt_f32_f16_sdwa %0, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "+v"(a[i])); }
                else if constexpr (K == FMA_MIX) { asm volatile("v_fma_mix_f32 %0, %0, %1, %2 op_sel_hi:[0,0,1]" : "+v"(a[i]) : "v"(b), "v"(c)); }
                else if constexpr (K == FMA_MIXLO) { asm volatile("v_fma_mixlo_f16 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c)); }
                else if constexpr (K == LSHL) { asm volatile("v_lshlrev_b32 %0, 16, %0" : "+v"(a[i])); }
                else if constexpr (K == PKFMA16) { asm volatile("v_pk_fma_f16 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c)); }
                else if constexpr (K == MULLO) { asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(ua) : "v"(ub)); }
                else if constexpr (K == MIX_EXP_2FMA) {
                    asm volatile("v_exp_f32 %0, %0\n\tv_fma_f32 %1, %1, %3, %4\n\tv_fma_f32 %2, %2, %3, %4"
                                 : "+v"(a[i]), "+v"(q0[i]), "+v"(q1[i]) : "v"(b), "v"(c));
                } else if constexpr (K == MIX_EXP_4FMA) {
                    asm volatile("v_exp_f32 %0, %0\n\tv_fma_f32 %1, %1, %3, %4\n\tv_fma_f32 %2, %2, %3, %4\n\t"
                                 "v_fma_f32 %1, %1, %3, %4\n\tv_fma_f32 %2, %2, %3, %4"
                                 : "+v"(a[i]), "+v"(q0[i]), "+v"(q1[i]) : "v"(b), "v"(c));
                } else if constexpr (K == MIX_EXP_2PK) {
                    asm volatile("v_exp_f32 %0, %0\n\tv_pk_fma_f32 %1, %1, %3, %4\n\tv_pk_fma_f32 %2, %2, %3, %4"
                                 : "+v"(a[i]), "+v"(p[i]), "+v"(r[i]) : "v"(b2), "v"(c2));
                } else if constexpr (K == SPLIT_EXP_FMA) {
                    if (wave & 1) { asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c)); }
                    else { asm volatile("v_exp_f32 %0, %0" : "+v"(a[i])); }
                } else if constexpr (K == SPLIT_EXP_PK) {
                    if (wave & 1) { asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(b2), "v"(c2)); }
                    else { asm volatile("v_exp_f32 %0, %0" : "+v"(a[i])); }
                } else if constexpr (K == MIX_SCAN) {
                    // the instruction multiset of one level-0 scan step for a channel PAIR (fd_scan.hip, CPL == 2, phase C):
                    // 12 transcendentals, 26 packed fp32, 6 plain -- independent here, so this is the issue floor of the mix
                    if (i == 0) {
                        asm volatile(
                            "v_exp_f32 %0, %0\n\tv_exp_f32 %1, %1\n\tv_log_f32 %2, %2\n\tv_log_f32 %3, %3\n\t"
                            "v_pk_fma_f32 %4, %4, %8, %9\n\tv_pk_fma_f32 %5, %5, %8, %9\n\tv_pk_fma_f32 %6, %6, %8, %9\n\tv_pk_fma_f32 %7, %7, %8, %9\n\t"
                            "v_pk_mul_f32 %4, %4, %8\n\tv_pk_mul_f32 %5, %5, %8\n\tv_pk_add_f32 %6, %6, %9\n\tv_pk_mul_f32 %7, %7, %8\n\t"
                            "v_min_f32 %0, %0, %10\n\tv_min_f32 %1, %1, %10\n\tv_add_f32 %2, %2, %10\n\tv_add_f32 %3, %3, %10\n\t"
                            "v_lshlrev_b32 %11, 16, %11\n\tv_and_b32 %12, 0xffff0000, %12"
                            : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3])
                            : "v"(b2), "v"(c2), "v"(b), "v"(ua), "v"(ub));
                    } else if (i <= 4) {
                        // per state n: pk_mul (a*dt), 2 exp, pk_mul (dtu*B), pk_fma (h), pk_fma (acc)
                        asm volatile("v_pk_mul_f32 %0, %0, %3\n\tv_exp_f32 %1, %1\n\tv_exp_f32 %2, %2\n\t"
                                     "v_pk_mul_f32 %0, %5, %3\n\tv_pk_fma_f32 %0, %0, %3, %4\n\tv_pk_fma_f32 %0, %5, %3, %4"
                                     : "+v"(p[i]), "+v"(a[i]), "+v"(a[(i + 4) % NACC]) : "v"(b2), "v"(c2), "v"(r[i]));
                    }
                }
            }
        }
    }
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += a[i] + p[i].x + p[i].y + q0[i] + q1[i] + r[i].x;
    s += __builtin_bit_cast(float, ua) + b2.x + c2.x;
    if (s == 12345.678f) out[tid] = s;
    if ((threadIdx.x & 63) == 0) cyc[tid >> 6] = t1 - t0;
    if (tid == 0) {
        if (K == MIX_EXP_2FMA || K == MIX_EXP_2PK) n = 3;
        else if (K == MIX_EXP_4FMA) n = 5;
        else if (K == MIX_SCAN) n = 18 + 4 * 6;
        else n = NACC;
        if (K == MIX_SCAN) *ninstr = n * INNER * ITERS; else if (n != NACC) *ninstr = n * NACC * INNER * ITERS; else *ninstr = NACC * INNER * ITERS;
    }
}

template <int K>
void run(const float *in, float *out, uint64_t *cyc, int *ninstr, int wps) {
    // wps waves per SIMD: 256-thread blocks (4 waves -> one per SIMD of a CU), wps blocks per CU, 256 CUs
    const int blocks = 256 * wps;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(probe<K>, dim3(blocks), dim3(256), 0, 0, in, out, cyc, ninstr);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(probe<K>, dim3(blocks), dim3(256), 0, 0, in, out, cyc, ninstr);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<uint64_t> h(blocks * 4);
    hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    int ni; hipMemcpy(&ni, ninstr, 4, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    const double med = (double)h[h.size() / 2];
    // s_memtime ticks at 100 MHz on gfx950?  report both: ticks per instruction and wall-derived cycles at 2.4 GHz
    const double total_instr = (double)ni * blocks * 4;
    const double simd_cyc_wall = ms * 1e-3 * 2.4e9;         // cycles of a 2.4 GHz clock during the kernel
    printf("%-52s wps %d: %7.3f ms  | per-wave %8.2f ticks/instr | SIMD cycles(2.4GHz)/wave64-instr %6.2f | lanes/clk/SIMD %5.1f\n",
           KN[K], wps, ms, med / ni, simd_cyc_wall / (total_instr / 1024.0), 64.0 * (total_instr / 1024.0) / simd_cyc_wall);
}

template <int K>
void sweep(const float *in, float *out, uint64_t *cyc, int *ninstr) {
    for (int wps : {1, 2, 4, 8}) run<K>(in, out, cyc, ninstr, wps);
}

int main() {
    float *in, *out; uint64_t *cyc; int *ninstr;
    hipMalloc(&in, 4096 * 4); hipMalloc(&out, 256 * 8 * 256 * 4); hipMalloc(&cyc, 256 * 8 * 4 * 8); hipMalloc(&ninstr, 4);
    std::vector<float> h(4096);
    srand(1);
    for (auto &v : h) v = 0.25f + 0.5f * (rand() / (float)RAND_MAX);
    hipMemcpy(in, h.data(), 4096 * 4, hipMemcpyHostToDevice);
    // warm the clocks
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(probe<FMA>, dim3(2048), dim3(256), 0, 0, in, out, cyc, ninstr);
    hipDeviceSynchronize();
    if (getenv("VR_CVT_ONLY")) {           // the 16-bit format conversions of the library's two builds (round 6)
        sweep<FMA>(in, out, cyc, ninstr);
        sweep<CVTPK>(in, out, cyc, ninstr);
        sweep<CVTPK_F16>(in, out, cyc, ninstr);
        sweep<CVTPKRTZ>(in, out, cyc, ninstr);
        sweep<LSHL>(in, out, cyc, ninstr);
        sweep<CVT_F32_F16>(in, out, cyc, ninstr);
        sweep<CVT_F32_F16_SDWA>(in, out, cyc, ninstr);
        sweep<FMA_MIX>(in, out, cyc, ninstr);
        sweep<FMA_MIXLO>(in, out, cyc, ninstr);
        return 0;
    }
    sweep<FMA>(in, out, cyc, ninstr);
    sweep<FMA_DEP>(in, out, cyc, ninstr);
    sweep<PKFMA>(in, out, cyc, ninstr);
    sweep<PKMUL>(in, out, cyc, ninstr);
    sweep<PKADD>(in, out, cyc, ninstr);
    sweep<EXP>(in, out, cyc, ninstr);
    sweep<LOG>(in, out, cyc, ninstr);
    sweep<CVTPK>(in, out, cyc, ninstr);
    sweep<PKFMA16>(in, out, cyc, ninstr);
    sweep<MULLO>(in, out, cyc, ninstr);
    sweep<MIX_EXP_2FMA>(in, out, cyc, ninstr);
    sweep<MIX_EXP_4FMA>(in, out, cyc, ninstr);
    sweep<MIX_EXP_2PK>(in, out, cyc, ninstr);
    sweep<SPLIT_EXP_FMA>(in, out, cyc, ninstr);
    sweep<SPLIT_EXP_PK>(in, out, cyc, ninstr);
    sweep<MIX_SCAN>(in, out, cyc, ninstr);
    return 0;
}
