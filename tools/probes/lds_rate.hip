// LDS read issue rate at the wgrad kernel's occupancy (one 8-wave work-group per CU, 2 waves per SIMD): cycles per wave-instruction
// for ds_read_b64_tr_b16 / ds_read_b64 / ds_read_b128, conflict-free addresses, 16 independent reads per lgkmcnt(0).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

template <int MODE>
__global__ __launch_bounds__(512) void probe(unsigned long long* out, int iters, int nwaves) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 32768 / 4; i += 512) reinterpret_cast<unsigned*>(smem)[i] = i;
    __syncthreads();
    if (wave >= nwaves) return;
    // wgrad's patch-fragment addressing: lane -> pixel 8 g + (sl >> 2), 64-byte block swizzle, 8 (sl & 3) bytes
    const int g = lane >> 5, G16 = (lane >> 4) & 1, sl = lane & 15, t4 = sl >> 2;
    unsigned addr;
    if (MODE == 2) addr = lane * 16;                       // b128: 1 KiB contiguous
    else if (MODE == 1) addr = lane * 8;                   // b64: 512 B contiguous
    else addr = (8 * g + t4) * 128 + (((t4 >> 1) & 1) << 6) + 32 * G16 + 8 * (sl & 3);
    addr += wave * 2048;
    unsigned acc = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            if (MODE == 0) {
                const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(smem + addr + k * 1024 % 16384));
                acc ^= (unsigned)v[0] ^ (unsigned)v[3];
            } else if (MODE == 1) {
                const u32x2 v = *reinterpret_cast<const u32x2*>(smem + addr + k * 1024 % 16384);
                acc ^= v[0] ^ v[1];
            } else {
                const u32x4 v = *reinterpret_cast<const u32x4*>(smem + addr + k * 1024 % 16384);
                acc ^= v[0] ^ v[3];
            }
        }
        asm volatile("" ::: "memory");
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) { out[blockIdx.x * 8 + wave] = t1 - t0; }
    if (acc == 0x12345678u) out[0] = acc;
}

int main() {
    unsigned long long* d; hipMalloc(&d, 256 * 8 * 8);
    unsigned long long h[8];
    const int iters = 2000;
    const char* names[3] = {"ds_read_b64_tr_b16", "ds_read_b64", "ds_read_b128"};
    for (int mode = 0; mode < 3; ++mode)
        for (int nw : {8, 4, 1}) {
            for (int rep = 0; rep < 2; ++rep) {
                if (mode == 0) hipLaunchKernelGGL(probe<0>, dim3(256), dim3(512), 65536, 0, d, iters, nw);
                if (mode == 1) hipLaunchKernelGGL(probe<1>, dim3(256), dim3(512), 65536, 0, d, iters, nw);
                if (mode == 2) hipLaunchKernelGGL(probe<2>, dim3(256), dim3(512), 65536, 0, d, iters, nw);
                hipDeviceSynchronize();
            }
            hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
            double per_wave = (double)h[0] / (iters * 16.0);
            printf("%-20s waves/CU=%d: %.2f cycles per instruction per wave -> %.2f cycles per instruction per CU (%.0f B/clk/CU)\n", names[mode], nw,
                   per_wave, per_wave / nw, (mode == 2 ? 1024.0 : 512.0) / (per_wave / nw));
        }
    return 0;
}
