// Multi-tensor Adam for gfx950: ONE launch updates every parameter of a model (reference train.py:99-103: torch.optim.Adam on the
// VQ model's ~400 parameters, 95 M fp32 elements).  The step is a stream of 7 fp32 passes (read p, g, m, v; write p, m, v = 28 bytes per
// element, 2.67 GB for VQ-IMG: 0.42 ms at 6.3 TB/s); torch's fused multi-tensor path takes 10 launches and 0.75 ms for it
// (profiles/r04_kernel_trace_vq_final.txt).  A block owns 4096 consecutive elements of one tensor; the (tensor, offset) of a block comes
// from a table in device memory (binary search over first_block, as in misc.hip's tiled pack), so tiny tensors (biases, norm
// weights) cost one block each and no launch of their own.
// Arithmetic and its ORDER follow torch's fused kernel (torch/csrc ... FusedAdamMathFunctor, non-amsgrad, maximize = false):
//   g' = g + wd * p;  m = b1 m + (1 - b1) g';  v = b2 v + (1 - b2) g' g';  p -= (lr / bc1) * m / (sqrt(v) / sqrt(bc2) + eps)
// with bc1 = 1 - b1^t, bc2 = 1 - b2^t computed by the caller in double precision.
#include "mas_common.h"
#include "../../include/mas_hip.h"
#include <math.h>

namespace {

constexpr int ADAM_NT = 256, ADAM_PER_BLOCK = 4096;     // 4 x float4 per thread and tensor

__global__ __launch_bounds__(ADAM_NT) void adam_multi_kernel(const MasAdamItem* __restrict__ items, int n_items, float step_size, float inv_bc2_sqrt,
                                                             float b1, float b2, float eps, float wd) {   // (inv_bc2_sqrt: sqrt(bias_correction2) itself)
    int lo = 0, hi = n_items - 1;
    const int b = blockIdx.x;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (items[mid].first_block <= b) lo = mid; else hi = mid - 1;
    }
    const MasAdamItem it = items[lo];
    const long long base = (long long)(b - it.first_block) * ADAM_PER_BLOCK;
    const float omb1 = 1.0f - b1, omb2 = 1.0f - b2;
    auto upd = [&](float& p, float g, float& m, float& v) {
        if (wd != 0.0f) g += p * wd;
        m = fmaf(omb1, g - m, m);                    // torch: lerp(exp_avg, grad, 1 - beta1)
        v = b2 * v + omb2 * g * g;
        const float denom = sqrtf(v) / inv_bc2_sqrt + eps;
        p -= step_size * m / denom;
    };
    const bool vec = ((reinterpret_cast<uintptr_t>(it.p) | reinterpret_cast<uintptr_t>(it.g) | reinterpret_cast<uintptr_t>(it.m) |
                       reinterpret_cast<uintptr_t>(it.v)) & 15) == 0;
    if (vec && base + ADAM_PER_BLOCK <= it.n) {
        f32x4 p[4], g[4], m[4], v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const long long i = base + (long long)(k * ADAM_NT + threadIdx.x) * 4;
            p[k] = *reinterpret_cast<const f32x4*>(it.p + i); g[k] = *reinterpret_cast<const f32x4*>(it.g + i);
            m[k] = *reinterpret_cast<const f32x4*>(it.m + i); v[k] = *reinterpret_cast<const f32x4*>(it.v + i);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { float pp = p[k][e], mm = m[k][e], vv = v[k][e]; upd(pp, g[k][e], mm, vv); p[k][e] = pp; m[k][e] = mm; v[k][e] = vv; }
            const long long i = base + (long long)(k * ADAM_NT + threadIdx.x) * 4;
            *reinterpret_cast<f32x4*>(it.p + i) = p[k]; *reinterpret_cast<f32x4*>(it.m + i) = m[k]; *reinterpret_cast<f32x4*>(it.v + i) = v[k];
        }
    } else {                                         // a tensor's last block, or unaligned storage: element by element
        for (int k = threadIdx.x; k < ADAM_PER_BLOCK; k += ADAM_NT) {
            const long long i = base + k;
            if (i < it.n) { float pp = it.p[i], mm = it.m[i], vv = it.v[i]; upd(pp, it.g[i], mm, vv); it.p[i] = pp; it.m[i] = mm; it.v[i] = vv; }
        }
    }
}

}  // namespace

extern "C" int mas_adam_blocks(long long numel) { return numel <= 0 ? 0 : (int)((numel + ADAM_PER_BLOCK - 1) / ADAM_PER_BLOCK); }

extern "C" int mas_adam_multi(const MasAdamItem* items_device, int n_items, int total_blocks, float lr, float beta1, float beta2, float eps,
                              float weight_decay, double bias_correction1, double bias_correction2, void* stream) {
    MAS_ENTER();
    if (!items_device || n_items <= 0 || total_blocks <= 0) MAS_FAIL(MAS_EINVAL, "adam_multi: empty batch");
    if (!(bias_correction1 > 0.0) || !(bias_correction2 > 0.0)) MAS_FAIL(MAS_EINVAL, "adam_multi: bias corrections must be positive (step >= 1)");
    const float step_size = (float)((double)lr / bias_correction1);
    const float inv_bc2_sqrt = (float)sqrt(bias_correction2);      // (passed as the divisor, like torch's bias_correction2_sqrt)
    hipLaunchKernelGGL(adam_multi_kernel, dim3((unsigned)total_blocks), dim3(ADAM_NT), 0, reinterpret_cast<hipStream_t>(stream), items_device, n_items,
                       step_size, inv_bc2_sqrt, beta1, beta2, eps, weight_decay);
    MAS_CHECK_LAUNCH("adam_multi");
    return MAS_OK;
}
