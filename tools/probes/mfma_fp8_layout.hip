// Probe (development tool): operand lane map of v_mfma_scale_f32_16x16x128_f8f6f4 with e4m3 operands, checked with
// exact small-integer data against a host matmul.  Hypothesis H1: lane l holds A[row l & 15][k = 32 (l >> 4) + j] and
// B[k = 32 (l >> 4) + j][col l & 15] in byte j = 0..31 of its 8-dword fragment; C/D as every 16x16 MFMA.
//   hipcc --offload-arch=gfx950 -O2 tools/probes/mfma_fp8_layout.hip -o /tmp/mfma_fp8_layout && /tmp/mfma_fp8_layout
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

__global__ void k(const uint8_t *A, const uint8_t *B, float *D, int hyp) {   // A [16][128], B [128][16] fp8 bytes
    const int l = threadIdx.x, r = l & 15, g = l >> 4;
    union { i32x8 v; uint8_t b[32]; } a, bb;
    for (int j = 0; j < 32; ++j) {
        const int kk = hyp == 0 ? 32 * g + j : (hyp == 1 ? 16 * g + (j & 15) + 64 * (j >> 4) : 8 * g + (j & 7) + 32 * (j >> 3));
        a.b[j] = A[r * 128 + kk];
        bb.b[j] = B[kk * 16 + r];
    }
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a.v, bb.v, acc, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
    for (int i = 0; i < 4; ++i) D[(4 * g + i) * 16 + r] = acc[i];
}

static uint8_t enc(int v) {    // small integers as OCP e4m3: sign, exponent bias 7, 3 mantissa bits
    static const uint8_t tab[5] = {0x00, 0x38, 0x40, 0x44, 0x48};   // 0, 1, 2, 3, 4
    return (uint8_t)(tab[abs(v)] | (v < 0 ? 0x80 : 0));
}

int main() {
    uint8_t hA[16 * 128], hB[128 * 16];
    int iA[16 * 128], iB[128 * 16];
    srand(1);
    for (int i = 0; i < 16 * 128; ++i) { iA[i] = rand() % 9 - 4; hA[i] = enc(iA[i]); }
    for (int i = 0; i < 128 * 16; ++i) { iB[i] = rand() % 9 - 4; hB[i] = enc(iB[i]); }
    uint8_t *dA, *dB;
    float *dD, hD[256];
    hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dD, sizeof hD);
    hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice);
    hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
    for (int hyp = 0; hyp < 3; ++hyp) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD, hyp);
        hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int i = 0; i < 16; ++i)
            for (int j = 0; j < 16; ++j) {
                int s = 0;
                for (int kk = 0; kk < 128; ++kk) s += iA[i * 128 + kk] * iB[kk * 16 + j];
                if ((float)s != hD[i * 16 + j]) ++bad;
            }
        printf("hypothesis %d: %d of 256 outputs wrong%s\n", hyp, bad, bad ? "" : "  <-- lane map confirmed");
    }
    return 0;
}
