// Is a FUSED Winograd F(2x2,3x3) 128->128 convolution feasible on one gfx950 CU?  (VERDICT r5 "next" #1; the arithmetic is in
// profiles/r06_winograd_feasibility.txt, this probe measures the inner loop of the ONLY geometry the register file admits.)
//
// F(2x2,3x3) needs the accumulators of all 16 transform points of a tile set at once (the output transform follows the Cin
// reduction): 16 x T tiles x C couts fp32.  A CU has 4 x 512 x 64 = 131072 registers; with half of them as accumulators
// T x C = 32 x 128: a work-group tile of 32 Winograd tiles = 8 x 16 output pixels x 128 couts (the direct kernel: 16 x 32 pixels x
// 128 couts in the same 65536 accumulators), and the transformed weights U (16 points x Cin x 128 couts x 2 B = 512 KiB at
// Cin = 128) are re-streamed L2 -> LDS for every one of those 128-pixel tiles: 4 KiB per output pixel against the direct kernel's
// 0.58.  The probe runs that loop with the right instruction mix and byte counts (NOT a convolution: no indexing, no epilogue):
//   per work-group tile, per 16-channel k-step: every wave (8 waves, 2 transform points each) DMAs its own 8 KiB of U
//   (2 points x 128 couts x 32 B) from an L2-resident 512 KiB image into its private double buffer, the work-group transforms
//   10 x 18 raw pixels x 16 channels into V[16 points][32 tiles][16 ch] (LDS -> fp32 adds -> bf16 -> LDS: 28 VALU per thread),
//   then 2 B fragments + 8 A fragments feed 8 MFMAs per wave.  8 k-steps per tile (Cin = 128), 64 tiles per CU = the
//   128 -> 128 @256^2 B = 32 launch (16384 tiles of 128 pixels).
// modes (bit mask): 1 weight DMA, 2 fragment reads + MFMAs, 4 input-transform VALU + LDS traffic.
// usage: wino_loop [tiles_per_wg=64]       hipcc --offload-arch=gfx950 -O3 -o wino_loop wino_loop.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;

constexpr int WB = 8 * 1024;                 // one wave's U slice of one k-step: 2 points x 128 couts x 32 B
constexpr int L_W = 0;                       // [8 waves][2 buffers][8 KiB]
constexpr int L_V = 8 * 2 * WB;              // [16 points][32 tiles][32 B] = 16 KiB
constexpr int L_RAW = L_V + 16 * 1024;       // 10 x 18 pixels x 32 B = 5760 B (rounded to 8 KiB)
constexpr int L_TOTAL = L_RAW + 8 * 1024;

template <int MODE>
__global__ __launch_bounds__(512, 2) void wino_loop(const unsigned char* __restrict__ u_img, int tiles, float* sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, g = lane >> 5;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(u_img), 0, 512 * 1024, 0x00020000);
    for (int i = tid; i < L_TOTAL / 16; i += 512)      // random-ish bf16 in LDS (operand toggling as in a real launch)
        reinterpret_cast<u32x4*>(smem)[i] = u32x4{0x3f813e7fu ^ (i * 2654435761u & 0x807f807fu), 0xbf104011u ^ (i * 40503u & 0x807f807fu),
                                                  0x3e99bf33u ^ (i * 69069u & 0x807f807fu), 0x40123daau ^ (i * 1664525u & 0x807f807fu)};
    __syncthreads();
    f32x16 acc[2][4];
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[p][c][r] = 0.f;

    auto w_issue = [&](int kstep, int buf) {           // 8 pieces of 1 KiB
        if constexpr (MODE & 1) {
            const int soff = __builtin_amdgcn_readfirstlane((kstep & 7) * 64 * 1024 + wave * WB);
#pragma unroll
            for (int q = 0; q < 8; ++q)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(smem + L_W + (wave * 2 + buf) * WB + q * 1024),
                                                         16, lane * 16, soff + q * 1024, 0, 0);
        }
    };
    w_issue(0, 0);
    int ks = 0;
    for (int t = 0; t < tiles; ++t) {
#pragma unroll 1
        for (int s = 0; s < 8; ++s, ++ks) {
            w_issue(ks + 1, (ks + 1) & 1);
            if constexpr (MODE & 4) {
                // input transform, the work-group's share per k-step: 32 tiles x 16 ch x 56 lane-operations / 512 threads = 56 per thread
                // for 2 (tile, 8-channel) ... modelled as: 16 raw bf16 of one channel pair -> 16 transformed: 16 unpacks, 32 adds, 8 converts
                const unsigned* raw = reinterpret_cast<const unsigned*>(smem + L_RAW) + (tid & 255) * 4;
                u32x4 r0 = *reinterpret_cast<const u32x4*>(raw), r1 = *reinterpret_cast<const u32x4*>(raw + 1024);
                float d[16];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    d[4 * i] = __uint_as_float(r0[i] << 16); d[4 * i + 1] = __uint_as_float(r0[i] & 0xffff0000u);
                    d[4 * i + 2] = __uint_as_float(r1[i] << 16); d[4 * i + 3] = __uint_as_float(r1[i] & 0xffff0000u);
                }
                float e[16], v[16];
#pragma unroll
                for (int c = 0; c < 4; ++c) {           // B^T d
                    e[c] = d[c] - d[8 + c]; e[4 + c] = d[4 + c] + d[8 + c]; e[8 + c] = d[8 + c] - d[4 + c]; e[12 + c] = d[4 + c] - d[12 + c];
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {           // (B^T d) B
                    v[4 * r] = e[4 * r] - e[4 * r + 2]; v[4 * r + 1] = e[4 * r + 1] + e[4 * r + 2];
                    v[4 * r + 2] = e[4 * r + 2] - e[4 * r + 1]; v[4 * r + 3] = e[4 * r + 1] - e[4 * r + 3];
                }
                u32x4 o0, o1;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const bf16x2 a = __builtin_convertvector(f32x2{v[2 * i], v[2 * i + 1]}, bf16x2);
                    const bf16x2 b = __builtin_convertvector(f32x2{v[8 + 2 * i], v[8 + 2 * i + 1]}, bf16x2);
                    o0[i] = *reinterpret_cast<const unsigned*>(&a); o1[i] = *reinterpret_cast<const unsigned*>(&b);
                }
                *reinterpret_cast<u32x4*>(smem + L_V + tid * 16) = o0;
                *reinterpret_cast<u32x4*>(smem + L_V + 8192 + tid * 16) = o1;
            }
            if constexpr (MODE & 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");   // this k-step's U has landed (in-order retirement)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();              // V of this k-step visible
            if constexpr (MODE & 2) {
                const unsigned char* wb = smem + L_W + (wave * 2 + (ks & 1)) * WB;
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    const bf16x8 b = *reinterpret_cast<const bf16x8*>(smem + L_V + ((2 * wave + p) * 32 + l31) * 32 + ((g ^ (l31 >> 4)) << 4));
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const bf16x8 a = *reinterpret_cast<const bf16x8*>(wb + (p * 128 + c * 32 + l31) * 32 + ((g ^ (l31 >> 4)) << 4));
                        acc[p][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[p][c], 0, 0, 0);
                    }
                }
            }
            __builtin_amdgcn_s_barrier();              // V may be overwritten
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float s = 0.f;
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) s += acc[p][c][r];
    if (s == 1234.5f) sink[0] = s + smem[L_V + tid];
}

template <int MODE>
float run(const unsigned char* u, int tiles, float* sink) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(wino_loop<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, L_TOTAL);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(wino_loop<MODE>, dim3(256), dim3(512), L_TOTAL, 0, u, tiles, sink);
    hipDeviceSynchronize();
    const int reps = 20;
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(wino_loop<MODE>, dim3(256), dim3(512), L_TOTAL, 0, u, tiles, sink);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms / reps;
}

int main(int argc, char** argv) {
    const int tiles = argc > 1 ? atoi(argv[1]) : 64;
    unsigned char* u; float* sink;
    hipMalloc(&u, 512 * 1024); hipMalloc(&sink, 4);
    unsigned* h = (unsigned*)malloc(512 * 1024);
    for (int i = 0; i < 128 * 1024; ++i) h[i] = 0x3f003f00u | ((i * 2654435761u) & 0x80ff80ffu);
    hipMemcpy(u, h, 512 * 1024, hipMemcpyHostToDevice);
    printf("Winograd F(2x2,3x3) inner-loop probe: 256 work-groups x %d tiles of 8x16 pixels x 128 couts, Cin = 128 (= %.0f output pixels)\n",
           tiles, 256.0 * tiles * 128);
    printf("U stream L2 -> LDS per launch: %.2f GB; MFMAs per launch: %.0f (direct 3x3: %.0f)\n", 256.0 * tiles * 512 * 1024 / 1e9,
           256.0 * tiles * 8 * 8 * 8, 256.0 * tiles * 128 * 128 * 128 * 9 / 16384.0);
    printf("%-44s %8s\n", "mode", "ms");
    printf("%-44s %8.3f\n", "U DMA only (1)", run<1>(u, tiles, sink));
    printf("%-44s %8.3f\n", "fragment reads + MFMAs only (2)", run<2>(u, tiles, sink));
    printf("%-44s %8.3f\n", "U DMA + MFMAs (3)", run<3>(u, tiles, sink));
    printf("%-44s %8.3f\n", "input transform only (4)", run<4>(u, tiles, sink));
    printf("%-44s %8.3f\n", "U DMA + MFMAs + input transform (7)", run<7>(u, tiles, sink));
    printf("gate (VERDICT r5 #1): 0.40 ms for the whole convolution; direct kernel K1: 0.50-0.52 ms\n");
    return 0;
}
