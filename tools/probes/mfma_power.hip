// Energy budget of the matrix pipe (VERDICT r4 "next" #1): how fast do the MFMA units run, and at what package power, when they are
// fed the way the convolution kernels feed them -- for SECONDS, so that the hwmon sampler (tools/probes/energy_budget.py) sees the
// steady state.  One wave tile of NI x NJ 32x32 accumulator blocks; per k-step NI + NJ operand fragments and NI * NJ
// v_mfma_f32_32x32x16_bf16.  Operand source:
//   const : one constant fragment pair (what tools/probes/mfma_peak measures: no operand toggling)
//   reg   : a pool of RANDOM bf16 fragments in registers, rotated every k-step (operand buses toggle, no LDS)
//   lds   : the fragments are ds_read_b128 from LDS filled with random bf16 (NI + NJ reads per NI * NJ MFMAs: 0.75 per MFMA for the
//           wide kernel's 4 x 2 wave tile, 0.5 for a 4 x 4 tile at one wave per SIMD)
// usage: mfma_power <const|reg|lds> <NIxNJ: 42|44> <wgs_per_cu> <seconds>     (256 threads per work-group = one wave per SIMD per WG)
// hipcc --offload-arch=gfx950 -O3 -o mfma_power mfma_power.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

__device__ __forceinline__ unsigned hash32(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
// two random bf16 in +-[0.5, 2): random sign and mantissa, exponent 126 / 127
__device__ __forceinline__ unsigned rand_bf16_pair(unsigned key) {
    const unsigned r = hash32(key);
    return (r & 0x80ff80ffu) | 0x3f003f00u;
}
__device__ __forceinline__ bf16x8 rand_frag(unsigned key) {
    u32x4 v = {rand_bf16_pair(key * 4u), rand_bf16_pair(key * 4u + 1), rand_bf16_pair(key * 4u + 2), rand_bf16_pair(key * 4u + 3)};
    return *reinterpret_cast<bf16x8*>(&v);
}

constexpr int POOL = 4;            // k-steps of register-resident fragments (reg mode)
constexpr int LDS_BYTES = 64 * 1024;

// MODE 0 const, 1 reg, 2 lds
template <int NI, int NJ, int MODE, int WPS>
__global__ __launch_bounds__(256, WPS) void mfma_loop(float* out, unsigned long long* clk, int iters) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[MODE == 2 ? LDS_BYTES : 16];
    f32x16 acc[NI][NJ];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int tid = threadIdx.x, lane = tid & 63;
    bf16x8 pa[MODE == 1 ? POOL : 1][NI], pb[MODE == 1 ? POOL : 1][NJ];
    if constexpr (MODE == 0) {
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
            for (int e = 0; e < 8; ++e) pa[0][i][e] = (__bf16)(float)(tid & 3);
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int e = 0; e < 8; ++e) pb[0][j][e] = (__bf16)1.0f;
    } else if constexpr (MODE == 1) {
#pragma unroll
        for (int p = 0; p < POOL; ++p) {
#pragma unroll
            for (int i = 0; i < NI; ++i) pa[p][i] = rand_frag((blockIdx.x * 256 + tid) * 64 + p * 8 + i);
#pragma unroll
            for (int j = 0; j < NJ; ++j) pb[p][j] = rand_frag((blockIdx.x * 256 + tid) * 64 + 32 + p * 8 + j);
        }
    } else {
        for (int k = tid; k < LDS_BYTES / 16; k += 256) *reinterpret_cast<bf16x8*>(lds + k * 16) = rand_frag(blockIdx.x * 8192 + k);
        __syncthreads();
    }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
        if constexpr (MODE == 2) {
            // 6 k-steps per iteration (one stage of the wide kernel); conflict-free: a wave reads 64 consecutive 16-byte slots
            const unsigned char* base = lds + (((unsigned)it * 1024u) & 8191u) + lane * 16;
#pragma unroll
            for (int n = 0; n < 6; ++n) {
                bf16x8 a[NI], b[NJ];
#pragma unroll
                for (int i = 0; i < NI; ++i) a[i] = *reinterpret_cast<const bf16x8*>(base + (n * (NI + NJ) + i) * 1024);
#pragma unroll
                for (int j = 0; j < NJ; ++j) b[j] = *reinterpret_cast<const bf16x8*>(base + (n * (NI + NJ) + NI + j) * 1024);
#pragma unroll
                for (int i = 0; i < NI; ++i)
#pragma unroll
                    for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[j], a[i], acc[i][j], 0, 0, 0);
            }
            asm volatile("" ::: "memory");
        } else {
#pragma unroll
            for (int n = 0; n < (MODE == 1 ? POOL : 4); ++n) {
                const int p = MODE == 1 ? n : 0;
#pragma unroll
                for (int i = 0; i < NI; ++i)
#pragma unroll
                    for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pb[p][j], pa[p][i], acc[i][j], 0, 0, 0);
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), w1 = wall_clock64();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    out[blockIdx.x * 256 + tid] = s;
    if (tid == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = w1 - w0; }
}

template <int NI, int NJ, int MODE, int WPS>
int run(int wgs_per_cu, double seconds) {
    int cus = 0;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    const int grid = cus * wgs_per_cu;
    float* out; unsigned long long* clk;
    hipMalloc(&out, sizeof(float) * grid * 256);
    hipMalloc(&clk, 16);
    const int ksteps = MODE == 2 ? 6 : (MODE == 1 ? POOL : 4);
    const int iters = 60000 / ksteps * 8 / (NI * NJ);           // ~ 15-25 ms per launch
    auto kern = mfma_loop<NI, NJ, MODE, WPS>;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    kern<<<grid, 256>>>(out, clk, iters);
    if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
    const auto T0 = std::chrono::steady_clock::now();
    double tail_ms = 0; long tail_n = 0, launches = 0;
    for (;;) {
        const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - T0).count();
        if (el >= seconds) break;
        // batches of 8 launches, each batch event-timed; the last second's batches make the reported rate
        hipEventRecord(e0);
        for (int k = 0; k < 8; ++k) kern<<<grid, 256>>>(out, clk, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        launches += 8;
        if (el >= seconds - 1.5) { tail_ms += ms; tail_n += 8; }
    }
    unsigned long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    const double flops = 2.0 * 32 * 32 * 16 * (double)NI * NJ * ksteps * iters * 4.0 * grid;
    const double ms1 = tail_ms / tail_n;
    printf("mfma_power mode=%s tile=%dx%d wgs/cu=%d (waves/SIMD=%d)%s: %ld launches, steady state %.3f ms per launch = %.1f TFLOP/s; "
           "s_memtime clock %.0f MHz; cycles per MFMA per SIMD %.2f\n", MODE == 0 ? "const" : MODE == 1 ? "reg" : "lds", NI, NJ, wgs_per_cu, wgs_per_cu,
           MODE == 2 ? (NI * NJ == 8 ? " 0.75 ds_read_b128/MFMA" : " 0.5 ds_read_b128/MFMA") : "", launches, ms1, flops / ms1 / 1e9,
           100.0 * h[0] / (double)h[1], (double)h[0] / ((double)NI * NJ * ksteps * iters * wgs_per_cu));
    return 0;
}

int main(int argc, char** argv) {
    if (argc < 5) { printf("usage: mfma_power <const|reg|lds> <42|44> <wgs_per_cu 1|2> <seconds>\n"); return 2; }
    const int mode = !strcmp(argv[1], "const") ? 0 : !strcmp(argv[1], "reg") ? 1 : 2;
    const int tile = atoi(argv[2]), wg = atoi(argv[3]);
    const double sec = atof(argv[4]);
    if (tile == 42 && wg == 2) {
        if (mode == 0) return run<4, 2, 0, 2>(2, sec);
        if (mode == 1) return run<4, 2, 1, 2>(2, sec);
        return run<4, 2, 2, 2>(2, sec);
    }
    if (tile == 42 && wg == 1) {
        if (mode == 0) return run<4, 2, 0, 1>(1, sec);
        if (mode == 1) return run<4, 2, 1, 1>(1, sec);
        return run<4, 2, 2, 1>(1, sec);
    }
    if (tile == 44 && wg == 1) {
        if (mode == 0) return run<4, 4, 0, 1>(1, sec);
        if (mode == 1) return run<4, 4, 1, 1>(1, sec);
        return run<4, 4, 2, 1>(1, sec);
    }
    printf("unsupported combination\n");
    return 2;
}
