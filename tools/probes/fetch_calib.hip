// Calibration of rocprofv3 FETCH_SIZE for the wide conv kernel's read pattern (MI355X_MICROARCH.md: FETCH_SIZE counts 128-B requests of a
// wide coalesced stream at 64 B; "other access widths uncalibrated").  Three streaming kernels over a 1 GiB buffer (> MALL):
//   full : every lane 16 B, 64 lanes = 1 KiB contiguous                         (the documented case: FETCH_SIZE = bytes / 2)
//   half : groups of 4 lanes read 64 B contiguous at a 128-B stride            (the 32-channel-chunk patch read: 64 B of each 128-B line)
//   half2: the same, then the OTHER 64 B of every line in a second pass         (both halves, separated in time like two chunks)
// Run:  rocprofv3 --pmc FETCH_SIZE --kernel-trace -d out -- ./fetch_calib   and compare with the byte counts printed here.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
__global__ void k_full(const u32x4* __restrict__ p, unsigned* sink, size_t n16) {
    u32x4 acc = {0, 0, 0, 0};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) { u32x4 v = p[i]; acc += v; }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 0x12345678u) *sink = 1;
}
__global__ void k_half(const u32x4* __restrict__ p, unsigned* sink, size_t nlines, int which) {
    u32x4 acc = {0, 0, 0, 0};
    const size_t total = nlines * 4;   // 4 lanes x 16 B per line
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t line = i >> 2, sl = i & 3;
        u32x4 v = p[line * 8 + which * 4 + sl]; acc += v;
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 0x12345678u) *sink = 1;
}
int main() {
    const size_t bytes = 1ull << 30;
    u32x4* p; unsigned* sink;
    hipMalloc(&p, bytes); hipMalloc(&sink, 4); hipMemset(p, 1, bytes);
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(k_full, dim3(4096), dim3(256), 0, 0, p, sink, bytes / 16);
        hipLaunchKernelGGL(k_half, dim3(4096), dim3(256), 0, 0, p, sink, bytes / 128, 0);
        hipLaunchKernelGGL(k_half, dim3(4096), dim3(256), 0, 0, p, sink, bytes / 128, 1);
    }
    hipDeviceSynchronize();
    printf("k_full requests %zu bytes; each k_half pass requests %zu bytes (64 B of every 128-B line)\n", bytes, bytes / 2);
    return 0;
}
