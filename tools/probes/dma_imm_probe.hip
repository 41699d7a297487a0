// Probe: does the immediate offset of `buffer_load_dwordx4 ... lds` advance the LDS address as well as the memory address?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include <vector>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
__global__ void k(const unsigned char* x, unsigned* out, int nbytes) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    for (int i = threadIdx.x; i < 4096 / 16; i += 64) *reinterpret_cast<u32x4*>(smem + i * 16) = u32x4{0xABABABABu, 0xABABABABu, 0xABABABABu, 0xABABABABu};
    __syncthreads();
    auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, nbytes, 0x00020000);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)smem, 16, threadIdx.x * 16, 0, 1024, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    for (int i = threadIdx.x; i < 4096 / 4; i += 64) out[i] = *reinterpret_cast<unsigned*>(smem + i * 4);
}
int main() {
    const int n = 4096;
    std::vector<unsigned char> h(n);
    for (int i = 0; i < n; ++i) h[i] = (unsigned char)((i * 7 + i / 256) % 251 + 1);
    unsigned char* dx; unsigned* dout;
    hipMalloc(&dx, n); hipMalloc(&dout, 4096);
    hipMemcpy(dx, h.data(), n, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 4096, 0, dx, dout, n);
    std::vector<unsigned> o(1024);
    hipMemcpy(o.data(), dout, 4096, hipMemcpyDeviceToHost);
    // where did memory bytes [1024, 2048) land?
    int at0 = 0, at1024 = 0;
    for (int i = 0; i < 256; ++i) {
        unsigned exp; memcpy(&exp, &h[1024 + i * 4], 4);
        at0 += (o[i] == exp); at1024 += (o[256 + i] == exp);
    }
    unsigned exp0; memcpy(&exp0, &h[0], 4);
    printf("memory[1024..2047] found at LDS+0: %d/256 dwords, at LDS+1024: %d/256; LDS+0 holds memory[0]? %d\n", at0, at1024, o[0] == exp0);
    printf("%s\n", at1024 == 256 ? "IMM_OFFSET_ADVANCES_LDS" : (at0 == 256 ? "IMM_OFFSET_MEMORY_ONLY" : "IMM_OFFSET_UNKNOWN"));
    return 0;
}
