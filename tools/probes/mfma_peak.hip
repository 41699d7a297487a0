// Attainable dense bf16 MFMA rate and shader clock on this box: every wave issues independent
// v_mfma_f32_32x32x16_bf16 from registers only (no LDS, no memory).  Prints TFLOP/s and the clock derived
// from s_memtime (shader clock) against wall_clock64 (constant 100 MHz).  hipcc --offload-arch=gfx950 -O3.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int NACC>
__global__ __launch_bounds__(256) void mfma_loop(float* out, unsigned long long* clk, int iters) {
    f32x16 acc[NACC];
    for (int a = 0; a < NACC; ++a)
        for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    bf16x8 x, y;
    for (int e = 0; e < 8; ++e) { x[e] = (__bf16)(float)(threadIdx.x & 3); y[e] = (__bf16)1.0f; }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), w0 = wall_clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int a = 0; a < NACC; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, acc[a], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), w1 = wall_clock64();
    float s = 0.f;
    for (int a = 0; a < NACC; ++a) s += acc[a][0];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = w1 - w0; }
}

template <int NACC>
void run(int wgs_per_cu, int iters) {
    int cus = 0;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    const int grid = cus * wgs_per_cu;
    float* out; unsigned long long* clk;
    hipMalloc(&out, sizeof(float) * grid * 256);
    hipMalloc(&clk, 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    mfma_loop<NACC><<<grid, 256>>>(out, clk, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    mfma_loop<NACC><<<grid, 256>>>(out, clk, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    const double flops = 2.0 * 32 * 32 * 16 * (double)NACC * iters * 4.0 * grid;
    printf("NACC=%d wgs/cu=%d: %.3f ms  %.1f TFLOP/s   s_memtime ticks=%llu wall(100MHz)=%llu -> s_memtime rate %.1f MHz;  "
           "cycles per MFMA per SIMD (at that rate)=%.2f\n", NACC, wgs_per_cu, ms, flops / ms / 1e9, h[0], h[1],
           100.0 * h[0] / (double)h[1], (double)h[0] / ((double)NACC * iters * wgs_per_cu));
    hipFree(out); hipFree(clk);
}

int main() {
    run<4>(1, 20000);
    run<4>(2, 20000);
    run<8>(1, 20000);
    return 0;
}
