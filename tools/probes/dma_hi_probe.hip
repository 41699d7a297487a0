// Probe: `buffer_load_dwordx4 ... lds` (MUBUF LDS-DMA) with an LDS destination ABOVE 64 KiB: where does the data land?
// (the LDS base travels in M0; if only M0[15:0] is honoured the destination wraps modulo 64 KiB)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include <vector>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
constexpr int LDSB = 160 * 1024;
__global__ void k(const unsigned char* x, unsigned* out, int nbytes, int dst_off, int use_global) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    for (int i = threadIdx.x; i < LDSB / 16; i += 64) *reinterpret_cast<u32x4*>(smem + i * 16) = u32x4{0xABABABABu, 0xABABABABu, 0xABABABABu, 0xABABABABu};
    __syncthreads();
    auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, nbytes, 0x00020000);
    if (use_global)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(x + threadIdx.x * 16),
                                         (__attribute__((address_space(3))) void*)(smem + dst_off), 16, 0, 0);
    else
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(smem + dst_off), 16, threadIdx.x * 16, 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    for (int i = threadIdx.x; i < LDSB / 4; i += 64) out[i] = *reinterpret_cast<unsigned*>(smem + i * 4);
}
int main() {
    const int n = 1024;
    std::vector<unsigned char> h(n);
    for (int i = 0; i < n; ++i) h[i] = (unsigned char)((i * 7 + i / 256) % 251 + 1);
    unsigned char* dx; unsigned* dout;
    (void)hipMalloc(&dx, n); (void)hipMalloc(&dout, LDSB);
    (void)hipMemcpy(dx, h.data(), n, hipMemcpyHostToDevice);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, LDSB);
    unsigned exp0; memcpy(&exp0, &h[0], 4);
    for (int use_global = 0; use_global < 2; ++use_global)
        for (int dst : {0x1000, 0x14800, 0x20000}) {
            hipLaunchKernelGGL(k, dim3(1), dim3(64), LDSB, 0, dx, dout, n, dst, use_global);
            std::vector<unsigned> o(LDSB / 4);
            (void)hipMemcpy(o.data(), dout, LDSB, hipMemcpyDeviceToHost);
            int found = -1, nfound = 0;
            for (int i = 0; i < LDSB / 4; ++i) if (o[i] == exp0) { if (found < 0) found = i * 4; ++nfound; }
            printf("%s dst 0x%05x: first dword of the source found at LDS byte 0x%05x (%d matches)%s\n", use_global ? "global_load_lds" : "buffer_load_lds",
                   dst, found, nfound, found == dst ? "  OK" : "  *** MISPLACED ***");
        }
    return 0;
}
