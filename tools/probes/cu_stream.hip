// How fast can ONE work-group per CU pull data, as a function of loads in flight and of sharing?  (round 4: the spatial attention
// kernels move everything at ~27 GB/s per CU whatever the prefetch depth; this probe separates "latency x in-flight" from a real limit.)
// 256 work-groups (one per CU; 160 KiB of dynamic LDS requested to force that), T threads each; every work-group streams REGION bytes
// `reps` times with DEPTH 16-byte loads per thread in flight (issued back to back, then consumed).
//   private : every work-group its own region                       (256 x 512 KiB = 128 MiB: L2 misses on first touch, MALL after)
//   shared8 : the 8 work-groups that land on one XCD (id % 8 equal, consecutive id / 8) share a region   (L2 hits after first touch)
// Prints GB/s per CU and aggregate.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

template <int DEPTH>
__global__ void k_stream(const u32x4* __restrict__ base, size_t region16, int share, int reps, unsigned* sink) {
    extern __shared__ unsigned char smem[];
    const int w = blockIdx.x;
    const int xcd = w % 8, j = w / 8;
    const size_t reg = share ? (size_t)(xcd * (gridDim.x / 8 / share) + j / share) : (size_t)w;
    const u32x4* p = base + reg * region16;
    u32x4 acc = {0, 0, 0, 0};
    const int T = blockDim.x;
    for (int rep = 0; rep < reps; ++rep) {
        for (size_t i0 = 0; i0 + (size_t)DEPTH * T <= region16; i0 += (size_t)DEPTH * T) {
            u32x4 v[DEPTH];
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) v[d] = p[i0 + (size_t)d * T + threadIdx.x];
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) acc += v[d];
        }
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 0x12345678u) { *sink = 1; smem[0] = 1; }
}

template <int DEPTH>
float run(const u32x4* p, size_t region_bytes, int share, int reps, int T, unsigned* sink) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_stream<DEPTH>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k_stream<DEPTH>, dim3(256), dim3(T), 150 * 1024, 0, p, region_bytes / 16, share, reps, sink);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(k_stream<DEPTH>, dim3(256), dim3(T), 150 * 1024, 0, p, region_bytes / 16, share, reps, sink);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms / 5;
}

int main() {
    const size_t region = 512 * 1024;
    u32x4* p; unsigned* sink;
    hipMalloc(&p, 256 * region); hipMalloc(&sink, 4); hipMemset(p, 1, 256 * region);
    const int reps = 4;
    printf("%-8s %5s %5s %10s %12s %12s\n", "mode", "T", "depth", "us", "GB/s per CU", "TB/s total");
    for (int share : {0, 8}) {
        for (int T : {256, 512, 1024}) {
            for (int depth : {2, 4, 8, 16, 32}) {
                float ms = 0;
                if (depth == 2) ms = run<2>(p, region, share, reps, T, sink);
                if (depth == 4) ms = run<4>(p, region, share, reps, T, sink);
                if (depth == 8) ms = run<8>(p, region, share, reps, T, sink);
                if (depth == 16) ms = run<16>(p, region, share, reps, T, sink);
                if (depth == 32) ms = run<32>(p, region, share, reps, T, sink);
                const double bytes = (double)region * reps;
                printf("%-8s %5d %5d %10.1f %12.1f %12.2f\n", share ? "shared8" : "private", T, depth, ms * 1e3, bytes / ms / 1e6, bytes * 256 / ms / 1e9);
            }
        }
    }
    return 0;
}
