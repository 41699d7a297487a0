// Do the matrix pipe and the VALU of a SIMD run at the same time?  (round 5: the ablation builds of the causal attention forward show
// its MFMA time, its softmax VALU time and its DMA wait ADDING UP -- profiles/r05_attn_ablation.txt.)
// One work-group = 4 waves = one wave per SIMD (x WPS work-groups per CU).  Per loop iteration a wave issues
//     M  v_mfma_f32_32x32x16_bf16   (two independent accumulator chains)   and
//     V  VALU instructions (v_fma_f32 on registers the MFMAs do not touch; or v_exp_f32 with TRANS=1),
// either interleaved (V / M VALU after every MFMA) or in two phases (all MFMAs, then all VALU).  Accumulators in ArchVGPRs (what hipcc
// emits for the attention kernels) or in AccVGPRs (inline assembly, the "a" constraint).
// Prints cycles per iteration (s_memtime / 100 MHz wall clock is avoided: wall time over many iterations and the shader clock from
// the MFMA-only run).   usage: coissue            hipcc --offload-arch=gfx950 -O3 -o coissue coissue.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

// MODE: 0 MFMA only, 1 VALU only, 2 interleaved, 3 two phases.  ACC: 0 ArchVGPR, 1 AccVGPR.  TRANS: VALU op = v_exp_f32 instead of v_fma_f32
template <int MODE, int ACC, int TRANS, int M, int V, int WPS>
__global__ __launch_bounds__(256, WPS) void probe(float* out, int iters, float seed) {
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
    bf16x8 a, b;
#pragma unroll
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(seed + (float)(threadIdx.x & 7)); b[e] = (__bf16)(seed * 0.5f); }
    float x[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) x[e] = seed + (float)e;
    const float c1 = 1.0000001f, c2 = seed * 1e-9f;
    auto mfma = [&](int i) {
        if constexpr (ACC == 0) {
            if (i & 1) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc1) : "v"(a), "v"(b));
            else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc0) : "v"(a), "v"(b));
        } else {
            if (i & 1) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc1) : "v"(a), "v"(b));
            else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc0) : "v"(a), "v"(b));
        }
    };
    auto valu = [&](int j) {
        if constexpr (TRANS) asm volatile("v_exp_f32 %0, %0" : "+v"(x[j & 7]));
        else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[j & 7]) : "v"(c1), "v"(c2));
    };
    for (int it = 0; it < iters; ++it) {
        if constexpr (MODE == 0) {
#pragma unroll
            for (int i = 0; i < M; ++i) mfma(i);
        } else if constexpr (MODE == 1) {
#pragma unroll
            for (int j = 0; j < V; ++j) valu(j);
        } else if constexpr (MODE == 2) {
#pragma unroll
            for (int i = 0; i < M; ++i) {
                mfma(i);
#pragma unroll
                for (int j = 0; j < V / M; ++j) valu(i * (V / M) + j);
            }
        } else {
#pragma unroll
            for (int i = 0; i < M; ++i) mfma(i);
#pragma unroll
            for (int j = 0; j < V; ++j) valu(j);
        }
    }
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc0[r] + acc1[r];
#pragma unroll
    for (int e = 0; e < 8; ++e) s += x[e];
    if (s == 12345.678f) out[threadIdx.x] = s;
}

template <int MODE, int ACC, int TRANS, int M, int V, int WPS>
double run(const char* what, float* out, int cus, double ghz) {
    const int iters = 20000;
    hipLaunchKernelGGL((probe<MODE, ACC, TRANS, M, V, WPS>), dim3(cus * WPS), dim3(256), 0, 0, out, 100, 1.0f);
    hipDeviceSynchronize();
    const auto t0 = std::chrono::steady_clock::now();
    hipLaunchKernelGGL((probe<MODE, ACC, TRANS, M, V, WPS>), dim3(cus * WPS), dim3(256), 0, 0, out, iters, 1.0f);
    hipDeviceSynchronize();
    const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    const double ns_it = sec * 1e9 / iters;
    printf("%-74s %8.1f ns per iteration", what, ns_it);
    if (ghz > 0) printf("  = %7.0f cycles at %.2f GHz", ns_it * ghz, ghz);
    printf("\n");
    fflush(stdout);
    return ns_it;
}

int main() {
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    float* out;
    hipMalloc(&out, 4096);
    constexpr int M = 16, V = 128;      // per iteration: 16 MFMAs (512 pipe cycles) and 128 VALU (512 issue cycles at 4 per instruction)
    const double t_m = run<0, 0, 0, M, V, 1>("MFMA only (16 x 32x32x16, ArchVGPR accumulators), 1 wave / SIMD", out, cus, 0);
    const double ghz = 16 * 32 / t_m;   // 16 MFMAs x 32 cycles
    printf("shader clock from the MFMA-only run: %.2f GHz (16 MFMAs = 512 cycles)\n", ghz);
    run<0, 1, 0, M, V, 1>("MFMA only, AccVGPR accumulators", out, cus, ghz);
    run<1, 0, 0, M, V, 1>("VALU only (128 v_fma_f32)", out, cus, ghz);
    run<1, 0, 1, M, V, 1>("VALU only (128 v_exp_f32)", out, cus, ghz);
    run<2, 0, 0, M, V, 1>("interleaved: MFMA + 8 v_fma each, ArchVGPR acc, 1 wave / SIMD", out, cus, ghz);
    run<2, 1, 0, M, V, 1>("interleaved: MFMA + 8 v_fma each, AccVGPR acc, 1 wave / SIMD", out, cus, ghz);
    run<3, 0, 0, M, V, 1>("two phases: 16 MFMA then 128 v_fma, ArchVGPR acc, 1 wave / SIMD", out, cus, ghz);
    run<3, 1, 0, M, V, 1>("two phases, AccVGPR acc, 1 wave / SIMD", out, cus, ghz);
    run<3, 0, 0, M, V, 2>("two phases, ArchVGPR acc, 2 waves / SIMD (ideal: 2 x 512 cycles per iteration pair)", out, cus, ghz);
    run<3, 1, 0, M, V, 2>("two phases, AccVGPR acc, 2 waves / SIMD", out, cus, ghz);
    run<3, 0, 0, M, V, 4>("two phases, ArchVGPR acc, 4 waves / SIMD", out, cus, ghz);
    run<3, 1, 0, M, V, 4>("two phases, AccVGPR acc, 4 waves / SIMD", out, cus, ghz);
    run<2, 0, 0, M, V, 4>("interleaved, ArchVGPR acc, 4 waves / SIMD", out, cus, ghz);
    run<2, 0, 1, M, V, 1>("interleaved: MFMA + 8 v_exp each, ArchVGPR acc, 1 wave / SIMD", out, cus, ghz);
    run<2, 1, 1, M, V, 1>("interleaved: MFMA + 8 v_exp each, AccVGPR acc, 1 wave / SIMD", out, cus, ghz);
    run<0, 0, 0, M, V, 4>("MFMA only, 4 waves / SIMD (pipe-bound: 4 x 512 cycles)", out, cus, ghz);
    run<1, 0, 0, M, V, 4>("VALU only, 4 waves / SIMD (4 x 512 cycles)", out, cus, ghz);
    return 0;
}
