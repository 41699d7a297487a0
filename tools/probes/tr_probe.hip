// Probe of ds_read_b64_tr_b16 semantics on gfx950: LDS holds u16 value == element index; every lane issues the
// transpose read at address base + lane_addr[lane] and prints the 4 u16 it receives.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
__global__ void probe(const int* lane_addr_bytes, uint16_t* out) {
    __shared__ __attribute__((aligned(16))) uint16_t lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (uint16_t)i;
    __syncthreads();
    const unsigned addr = (unsigned)(uintptr_t)lds + lane_addr_bytes[threadIdx.x];
    unsigned long long r;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(addr) : "memory");
    out[threadIdx.x * 4 + 0] = (uint16_t)(r & 0xffff);
    out[threadIdx.x * 4 + 1] = (uint16_t)((r >> 16) & 0xffff);
    out[threadIdx.x * 4 + 2] = (uint16_t)((r >> 32) & 0xffff);
    out[threadIdx.x * 4 + 3] = (uint16_t)((r >> 48) & 0xffff);
}
int main() {
    int h_addr[64]; uint16_t h_out[256];
    int *d_addr; uint16_t* d_out;
    hipMalloc(&d_addr, sizeof(h_addr)); hipMalloc(&d_out, sizeof(h_out));
    const char* names[3] = {"A: lane l -> byte l*8 (contiguous)", "B: lane l -> row (l&15)*64B + (l>>4)*8B  (16 rows of 32 elems, 4-elem column block per lane group)",
                            "C: lane l -> row ((l&15)>>2)... rows of 32B: byte ((l&3)*8 + ((l&15)>>2)*32 + (l>>4)*128)"};
    for (int t = 0; t < 3; ++t) {
        for (int l = 0; l < 64; ++l) {
            if (t == 0) h_addr[l] = l * 8;
            else if (t == 1) h_addr[l] = (l & 15) * 64 + (l >> 4) * 8;
            else h_addr[l] = (l & 3) * 8 + ((l & 15) >> 2) * 32 + (l >> 4) * 128;
        }
        hipMemcpy(d_addr, h_addr, sizeof(h_addr), hipMemcpyHostToDevice);
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d_addr, d_out);
        hipMemcpy(h_out, d_out, sizeof(h_out), hipMemcpyDeviceToHost);
        printf("== pattern %s\n", names[t]);
        for (int l = 0; l < 64; ++l) printf("lane %2d (addr elem %4d): %4d %4d %4d %4d\n", l, h_addr[l] / 2, h_out[l*4], h_out[l*4+1], h_out[l*4+2], h_out[l*4+3]);
    }
    return 0;
}
