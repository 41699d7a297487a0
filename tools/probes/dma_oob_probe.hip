// Probe: what does an OUT-OF-RANGE lane of `buffer_load_dwordx4 ... lds` (LDS-DMA through a buffer descriptor) leave in LDS?
// The conv3x3 stream kernel relies on "zeros" for its halo padding.  LDS is pre-filled with 0xAB; odd lanes get a voffset
// beyond num_records.  Prints what the odd lanes' 16-byte slots hold afterwards.
//   hipcc --offload-arch=gfx950 -O2 tools/probes/dma_oob_probe.hip -o tools/probes/dma_oob_probe && tools/probes/dma_oob_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include <vector>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

__global__ void k(const unsigned char* x, unsigned* out, int nbytes) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    *reinterpret_cast<u32x4*>(smem + threadIdx.x * 16) = u32x4{0xABABABABu, 0xABABABABu, 0xABABABABu, 0xABABABABu};
    __syncthreads();
    auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, nbytes, 0x00020000);
    const int voff = (lane & 1) ? (int)0x80000000 : lane * 16;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)smem, 16, voff, 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    const u32x4 v = *reinterpret_cast<u32x4*>(smem + threadIdx.x * 16);
    for (int e = 0; e < 4; ++e) out[threadIdx.x * 4 + e] = v[e];
    // OOB buffer_store must be dropped, OOB buffer_load to VGPRs must return 0
    const u32x4 r = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, 0, 0);
    for (int e = 0; e < 4; ++e) out[256 + threadIdx.x * 4 + e] = r[e];
}

int main() {
    const int n = 64 * 16;
    std::vector<unsigned char> h(n);
    for (int i = 0; i < n; ++i) h[i] = (unsigned char)(i % 251 + 1);
    unsigned char* dx; unsigned* dout;
    hipMalloc(&dx, n); hipMalloc(&dout, 512 * 4);
    hipMemcpy(dx, h.data(), n, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 1024, 0, dx, dout, n);
    std::vector<unsigned> o(512);
    hipMemcpy(o.data(), dout, 512 * 4, hipMemcpyDeviceToHost);
    int zero = 0, stale = 0, other = 0, ok_even = 0, vz = 0;
    for (int l = 0; l < 64; ++l) {
        for (int e = 0; e < 4; ++e) {
            const unsigned v = o[l * 4 + e];
            if (l & 1) { if (v == 0) ++zero; else if (v == 0xABABABABu) ++stale; else ++other; }
            else { unsigned exp; memcpy(&exp, &h[l * 16 + e * 4], 4); ok_even += (v == exp); }
            if ((l & 1) && o[256 + l * 4 + e] == 0) ++vz;
        }
    }
    printf("LDS-DMA OOB lanes: %d dwords zero, %d stale(0xAB), %d other; in-range dwords correct %d/128; VGPR OOB loads zero %d/128\n",
           zero, stale, other, ok_even, vz);
    printf("%s\n", (zero == 128 && ok_even == 128) ? "DMA_OOB_WRITES_ZERO" : "DMA_OOB_DOES_NOT_WRITE_ZERO");
    return 0;
}
