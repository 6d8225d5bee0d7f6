// Probe (round 4): how fast do the output stores of the 256x256 pointwise GEMM tile go by store SHAPE?
// A [M = 32768][N = 2048] bf16 matrix (134 MB, the in_proj of the 512-channel blocks at 64x64, batch 8) is written by
// 1024 workgroups of 512 threads, one 256 x 256 tile each, nothing else in the kernel:
//   mode 0  the kernel's epilogue: a wave instruction writes 16 pixels x 64 B (8 channels per lane, 4 lane groups),
//           the other half of each 128-B line by a later instruction
//   mode 1  full rows: a wave instruction writes 2 pixels x 512 B (32 lanes x 16 B per pixel)
//   mode 2  as mode 0 but both halves of a line by consecutive instructions (jp innermost)
// build: hipcc --offload-arch=gfx950 -O3 -o store_pattern store_pattern.hip ; run: ./store_pattern
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

template <int MODE>
__global__ __launch_bounds__(512) void k(uint16_t *out, int N) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int mt = blockIdx.x, nt = blockIdx.y;
    const u32x4 v = {(uint32_t)tid, (uint32_t)mt, (uint32_t)nt, 7u};
    if (MODE == 0 || MODE == 2) {
        const int wm = wave >> 1, wn = wave & 1, fr = lane & 15, fg = lane >> 4;
        if (MODE == 0) {
            for (int jp = 0; jp < 4; ++jp)
                for (int i = 0; i < 4; ++i) {
                    const int64_t m = mt * 256 + 64 * wm + 16 * i + fr;
                    const int n0 = nt * 256 + 128 * wn + 32 * jp + 8 * fg;
                    *(u32x4 *)(out + m * N + n0) = v;
                }
        } else {
            for (int i = 0; i < 4; ++i)
                for (int jp = 0; jp < 4; ++jp) {
                    const int64_t m = mt * 256 + 64 * wm + 16 * i + fr;
                    const int n0 = nt * 256 + 128 * wn + 32 * jp + 8 * fg;
                    *(u32x4 *)(out + m * N + n0) = v;
                }
        }
    } else {
        // 256 rows x 512 B: thread -> (row r0 + 16 k, 16-byte chunk c)
        const int c = tid & 31, r0 = tid >> 5;
        for (int kk = 0; kk < 16; ++kk) {
            const int64_t m = mt * 256 + r0 + 16 * kk;
            *(u32x4 *)(out + m * N + nt * 256 + 8 * c) = v;
        }
    }
}

int main() {
    const int M = 32768, N = 2048;
    uint16_t *out;
    hipMalloc(&out, (size_t)M * N * 2);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    dim3 grid(M / 256, N / 256), block(512);
    for (int mode = 0; mode < 3; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            for (int it = 0; it < 20; ++it) {
                if (mode == 0) hipLaunchKernelGGL(k<0>, grid, block, 0, 0, out, N);
                else if (mode == 1) hipLaunchKernelGGL(k<1>, grid, block, 0, 0, out, N);
                else hipLaunchKernelGGL(k<2>, grid, block, 0, 0, out, N);
            }
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep) printf("mode %d: %.1f us per launch, %.2f TB/s\n", mode, ms / 20 * 1e3, (double)M * N * 2 / (ms / 20 * 1e-3) / 1e12);
        }
    }
    return 0;
}
