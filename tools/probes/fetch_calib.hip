// Probe (round 4): what does rocprofv3's FETCH_SIZE report for the access shapes of this library?  MI355X_MICROARCH.md
// calibrates it for 16 B / lane full-line streaming reads only (reports exactly 1/2 of the bytes) and calls other
// widths uncalibrated.  Every kernel below reads the same 512 MiB buffer exactly once:
//   k_wide     16 B / lane, a wave instruction = 1 KiB contiguous                    (the calibrated case)
//   k_frag     the x_proj fragment pattern of the scan: a wave instruction = 16 rows x 64 B (row stride 256 B), the
//              other half of each 128-B line by the NEXT instruction of the same wave
//   k_dword    4 B / lane, a wave instruction = 256 B contiguous                      (scan recurrence, 2 channels / lane)
//   k_short    2 B / lane, a wave instruction = 128 B contiguous                      (scan recurrence, 1 channel / lane)
// build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 -o /tmp/fc tools/probes/fetch_calib.hip
//                              rocprofv3 --pmc FETCH_SIZE --kernel-trace -d out -o fc --output-format csv -- /tmp/fc
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
constexpr size_t BYTES = 512ull << 20;

__global__ void k_wide(const u32x4 *p, uint32_t *o) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const u32x4 v = p[i];
    if ((v.x ^ v.y ^ v.z ^ v.w) == 0x12345678u) o[0] = 1;
}
// tile = 16 rows x 256 B = 4 KiB per wave; 4 instructions (ks), lane = (fr = row, fg): bytes 64 ks + 16 fg of row fr
__global__ void k_frag(const unsigned char *p, uint32_t *o) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t tile = ((size_t)blockIdx.x * 4 + wave) * 4096;
    const int fr = lane & 15, fg = lane >> 4;
    uint32_t acc = 0;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        const u32x4 v = *(const u32x4 *)(p + tile + fr * 256 + 64 * ks + 16 * fg);
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) o[0] = 1;
}
__global__ void k_dword(const uint32_t *p, uint32_t *o) {
    // a wave walks 16 consecutive 256-B rows, one row per instruction (like 16 recurrence steps)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t base = ((size_t)blockIdx.x * 4 + wave) * 1024;        // dwords: 16 rows x 64
    uint32_t acc = 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc ^= p[base + r * 64 + lane];
    if (acc == 0x12345678u) o[0] = 1;
}
__global__ void k_short(const uint16_t *p, uint32_t *o) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t base = ((size_t)blockIdx.x * 4 + wave) * 1024;        // shorts: 16 rows x 64
    uint32_t acc = 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc ^= p[base + r * 64 + lane];
    if (acc == 0x1234u) o[0] = 1;
}

// second touch: fragment pattern, then the same 4 KiB tile again as 16 dword rows (what scan phase A does per block)
template <int GAP>
__global__ void k_retouch(const unsigned char *p, uint32_t *o) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t tile = ((size_t)blockIdx.x * 4 + wave) * 4096;
    const int fr = lane & 15, fg = lane >> 4;
    uint32_t acc = 0;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        const u32x4 v = *(const u32x4 *)(p + tile + fr * 256 + 64 * ks + 16 * fg);
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (GAP) __builtin_amdgcn_s_sleep(GAP);
    const uint32_t *q = (const uint32_t *)(p + tile);
#pragma unroll
    for (int r = 0; r < 16; ++r) acc ^= q[r * 64 + lane] * (r + 3);
    if (acc == 0x12345678u) o[0] = 1;
}

int main() {
    unsigned char *buf; uint32_t *o;
    hipMalloc(&buf, BYTES); hipMalloc(&o, 4);
    hipMemset(buf, 1, BYTES);
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(k_wide, dim3(BYTES / 16 / 256), dim3(256), 0, 0, (const u32x4 *)buf, o);
        hipLaunchKernelGGL(k_frag, dim3(BYTES / 4096 / 4), dim3(256), 0, 0, buf, o);
        hipLaunchKernelGGL(k_dword, dim3(BYTES / 4 / 1024 / 4), dim3(256), 0, 0, (const uint32_t *)buf, o);
        hipLaunchKernelGGL(k_short, dim3(BYTES / 2 / 1024 / 4), dim3(256), 0, 0, (const uint16_t *)buf, o);
        hipLaunchKernelGGL(k_retouch<0>, dim3(BYTES / 4096 / 4), dim3(256), 0, 0, buf, o);
        hipLaunchKernelGGL(k_retouch<100>, dim3(BYTES / 4096 / 4), dim3(256), 0, 0, buf, o);
    }
    hipDeviceSynchronize();
    printf("done: each kernel read %zu MiB\n", BYTES >> 20);
    return 0;
}
