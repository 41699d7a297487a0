// BatchNorm over [M][C] fp32 NHWC activations for gfx950: the `nn.SyncBatchNorm(embed_dim)` behind `quant_conv` (reference
// models/vqvae.py:15-16), training (batch statistics, running-statistics update) and evaluation, forward and backward.
//
// The latent it normalises is tiny (32 x 16 x 16 x 256 fp32 = 8 MB): what matters is the launch count and a fixed summation order,
// not bandwidth.  Every per-channel sum is two-stage -- per-work-group partial rows, then one fp64 fold in block order -- so the
// statistics and both parameter gradients are bitwise reproducible run to run.  The sums leave the library as fp64 so that the host side
// can all_reduce them across ranks (SyncBatchNorm's exchange: sum, sum of squares and element count in the forward; sum dy and
// sum dy * xhat in the backward) and hand the GLOBAL sums back to the finalize / apply kernels, which read the element count from
// device memory (no host synchronisation).
#include "mas_common.h"

namespace {

constexpr int NT = 256;
constexpr int BN_ROWS_PER_BLOCK = 128;
constexpr int BN_MAX_BLOCKS = 1024;

// per-work-group partial sums over a slice of rows: partial[blk][2][C], fp64 from the first addition on (the products x * x are exact
// in fp64): the variance is formed as E[x^2] - mean^2, which cancels catastrophically in fp32 as soon as |mean| >> std (mean / std = 100
// already cost 1e-3 of the variance with fp32 partial sums); in fp64 the same ratio costs 1e-12.  8 MB tensor: the extra
// fp64 issue slots do not show (the launches are latency-bound).
//   dy == nullptr: (sum x, sum x^2);  else (sum dy, sum dy * xhat) with xhat = (x - mean) * rstd
// A thread owns a column quad q = tid % quads and walks rows rl, rl + R, ... of the slice (R = NT / quads row lanes); LDS folds the row
// lanes in a fixed order.  C % 4 == 0, C <= 4 * NT.
// T = float or bf16 storage.  slope != 1 (backward sums only): the gradient first goes through the derivative of the LeakyReLU that
// follows the normalisation, g = dy * (u > 0 ? 1 : slope) with u = x * scale + shift recomputed from scale_shift.
template <typename T> struct Quad;
template <> struct Quad<float> {
    static __device__ __forceinline__ f32x4 ld(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
    static __device__ __forceinline__ void st(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
};
template <> struct Quad<bf16_t> {
    static __device__ __forceinline__ f32x4 ld(const bf16_t* p) {
        const bf16x4 v = *reinterpret_cast<const bf16x4*>(p);
        return f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
    }
    static __device__ __forceinline__ void st(bf16_t* p, f32x4 v) {
        *reinterpret_cast<bf16x4*>(p) = bf16x4{(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
    }
};

template <typename T>
__global__ __launch_bounds__(NT) void bn_partial_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                                        const float* __restrict__ mean_rstd, const float* __restrict__ scale_shift, float slope,
                                                        int M, int C, int rows_per_block, double* __restrict__ partial) {
    extern __shared__ double red[];                  // [R][2][C]
    const int quads = C / 4, R = NT / quads > 0 ? NT / quads : 1;
    const int tid = threadIdx.x, q = tid % quads, rl = tid / quads;
    const int r0 = blockIdx.x * rows_per_block, r1 = min(M, r0 + rows_per_block);
    double s1[4] = {0.0, 0.0, 0.0, 0.0}, s2[4] = {0.0, 0.0, 0.0, 0.0};
    if (rl < R && q < quads) {
        f32x4 mu = {0.0f, 0.0f, 0.0f, 0.0f}, rs = {1.0f, 1.0f, 1.0f, 1.0f}, sc = {1.0f, 1.0f, 1.0f, 1.0f}, sh = {0.0f, 0.0f, 0.0f, 0.0f};
        const bool leaky = dy && slope != 1.0f;
        if (dy) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { mu[e] = mean_rstd[(4 * q + e) * 2]; rs[e] = mean_rstd[(4 * q + e) * 2 + 1]; }
        }
        if (leaky) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { sc[e] = scale_shift[(4 * q + e) * 2]; sh[e] = scale_shift[(4 * q + e) * 2 + 1]; }
        }
        for (int r = r0 + rl; r < r1; r += R) {
            const f32x4 xv = Quad<T>::ld(x + (size_t)r * C + 4 * q);
            if (dy) {
                f32x4 dv = Quad<T>::ld(dy + (size_t)r * C + 4 * q);
                if (leaky) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) dv[e] = (xv[e] * sc[e] + sh[e] > 0.0f) ? dv[e] : dv[e] * slope;
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) { s1[e] += (double)dv[e]; s2[e] += (double)dv[e] * (double)((xv[e] - mu[e]) * rs[e]); }
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) { s1[e] += (double)xv[e]; s2[e] += (double)xv[e] * (double)xv[e]; }
            }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) { red[(rl * 2 + 0) * C + 4 * q + e] = s1[e]; red[(rl * 2 + 1) * C + 4 * q + e] = s2[e]; }
    }
    __syncthreads();
    for (int i = tid; i < 2 * C; i += NT) {
        double a = 0.0;
        for (int k = 0; k < R; ++k) a += red[k * 2 * C + i];
        partial[(size_t)blockIdx.x * 2 * C + i] = a;
    }
}

// sums[j] = sum over blocks of partial[blk][j] in fp64, block order; sums[2 C] = the element count of this rank
__global__ __launch_bounds__(64) void bn_fold_kernel(const double* __restrict__ partial, int nblk, int C, double count, double* __restrict__ sums) {
    const int j = blockIdx.x * 64 + threadIdx.x;
    if (j < 2 * C) {
        double a = 0.0;
        int b = 0;
        for (; b + 8 <= nblk; b += 8) {              // eight loads in flight, additions in block order
            double v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = partial[(size_t)(b + k) * 2 * C + j];
#pragma unroll
            for (int k = 0; k < 8; ++k) a += v[k];
        }
        for (; b < nblk; ++b) a += partial[(size_t)b * 2 * C + j];
        sums[j] = a;
    }
    if (j == 0) sums[2 * C] = count;
}

// training: mean / rstd from the (global) sums, the affine pair y = x * scale + shift, and the running statistics (unbiased variance,
// torch's convention); evaluation (sums == nullptr): the pair from the running statistics
__global__ __launch_bounds__(NT) void bn_finalize_kernel(const double* __restrict__ sums, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         float eps, float momentum, float* __restrict__ running_mean, float* __restrict__ running_var,
                                                         float* __restrict__ mean_rstd, float* __restrict__ scale_shift, int C) {
    const int c = blockIdx.x * NT + threadIdx.x;
    if (c >= C) return;
    double mean, var;
    if (sums) {
        const double n = sums[2 * C];
        mean = sums[c] / n;
        var = sums[C + c] / n - mean * mean;
        if (var < 0.0) var = 0.0;
        if (running_mean) running_mean[c] = (float)((1.0 - momentum) * running_mean[c] + momentum * mean);
        if (running_var) running_var[c] = (float)((1.0 - momentum) * running_var[c] + momentum * (n > 1.0 ? var * n / (n - 1.0) : var));
    } else {
        mean = running_mean[c];
        var = running_var[c];
    }
    const double rstd = 1.0 / sqrt(var + (double)eps);
    const double g = gamma ? (double)gamma[c] : 1.0, b = beta ? (double)beta[c] : 0.0;
    if (mean_rstd) { mean_rstd[2 * c] = (float)mean; mean_rstd[2 * c + 1] = (float)rstd; }
    scale_shift[2 * c] = (float)(g * rstd);
    scale_shift[2 * c + 1] = (float)(b - mean * g * rstd);
}

// y = act(x * scale + shift), act = LeakyReLU(slope) (slope == 1: none)
template <typename T>
__global__ __launch_bounds__(NT) void bn_apply_kernel(const T* __restrict__ x, const float* __restrict__ scale_shift, T* __restrict__ y, float slope,
                                                      long long n4, int C) {
    const int quads = C / 4;
    for (long long i = (long long)blockIdx.x * NT + threadIdx.x; i < n4; i += (long long)gridDim.x * NT) {
        const int q = (int)(i % quads);
        const f32x4 v = Quad<T>::ld(x + i * 4);
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float u = v[e] * scale_shift[(4 * q + e) * 2] + scale_shift[(4 * q + e) * 2 + 1];
            o[e] = u > 0.0f ? u : u * slope;
        }
        Quad<T>::st(y + i * 4, o);
    }
}

// dx = gamma * rstd * (dy - S1 / n - xhat * S2 / n) with the GLOBAL sums S1 = sum dy, S2 = sum dy * xhat and the global count n = sums[2 C]
template <typename T>
__global__ __launch_bounds__(NT) void bn_bwd_apply_kernel(const T* __restrict__ x, const T* __restrict__ dy, const float* __restrict__ mean_rstd,
                                                          const float* __restrict__ gamma, const float* __restrict__ scale_shift, float slope,
                                                          const double* __restrict__ sums, T* __restrict__ dx, long long n4, int C) {
    const int quads = C / 4;
    const double n = sums[2 * C];
    const bool leaky = slope != 1.0f;
    for (long long i = (long long)blockIdx.x * NT + threadIdx.x; i < n4; i += (long long)gridDim.x * NT) {
        const int q = (int)(i % quads);
        const f32x4 xv = Quad<T>::ld(x + i * 4);
        const f32x4 dv = Quad<T>::ld(dy + i * 4);
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int c = 4 * q + e;
            const float mu = mean_rstd[2 * c], rs = mean_rstd[2 * c + 1];
            const float k1 = (float)(sums[c] / n), k2 = (float)(sums[C + c] / n);
            const float g = gamma ? gamma[c] : 1.0f;
            float d = dv[e];
            if (leaky) d = (xv[e] * scale_shift[2 * c] + scale_shift[2 * c + 1] > 0.0f) ? d : d * slope;
            o[e] = g * rs * (d - k1 - (xv[e] - mu) * rs * k2);
        }
        Quad<T>::st(dx + i * 4, o);
    }
}

int bn_blocks(int M) {
    int nb = mas_cdiv(M, BN_ROWS_PER_BLOCK);
    return nb > BN_MAX_BLOCKS ? BN_MAX_BLOCKS : (nb < 1 ? 1 : nb);
}
int bn_check(const char* what, int M, int C) {
    if (M <= 0 || C <= 0) MAS_FAIL(MAS_EINVAL, "%s: bad shape M=%d C=%d", what, M, C);
    if (C % 4 || C > 4 * NT) MAS_FAIL(MAS_EUNSUPPORTED, "%s: C=%d must be a multiple of 4 and <= %d", what, C, 4 * NT);
    return MAS_OK;
}
int bn_grid(long long n4) {
    long long g = (n4 + NT - 1) / NT;
    const long long cap = 8LL * mas_num_cus();
    return (int)(g > cap ? cap : (g < 1 ? 1 : g));
}

}  // namespace

extern "C" size_t mas_bn_workspace(int M, int C) {
    if (M <= 0 || C <= 0) return 0;
    return (size_t)bn_blocks(M) * 2 * (size_t)C * sizeof(double);
}

namespace {
template <typename T>
int bn_partial_launch(const void* x, const void* dy, const float* mean_rstd, const float* scale_shift, float slope, int M, int C, double* sums,
                      void* workspace, hipStream_t s) {
    const int nblk = bn_blocks(M), rows_per_block = mas_cdiv(M, nblk);
    const int quads = C / 4, R = NT / quads > 0 ? NT / quads : 1;
    double* partial = reinterpret_cast<double*>(workspace);
    hipLaunchKernelGGL(bn_partial_kernel<T>, dim3(nblk), dim3(NT), (size_t)R * 2 * C * sizeof(double), s, (const T*)x, (const T*)dy, mean_rstd,
                       scale_shift, slope, M, C, rows_per_block, partial);
    MAS_CHECK_LAUNCH("bn_partial");
    hipLaunchKernelGGL(bn_fold_kernel, dim3(mas_cdiv(2 * C, 64)), dim3(64), 0, s, partial, nblk, C, (double)M, sums);
    MAS_CHECK_LAUNCH("bn_fold");
    return MAS_OK;
}
}  // namespace

extern "C" int mas_bn_partial_sums_act(const void* x, const void* dy, const float* mean_rstd, const float* scale_shift, float slope, int dtype,
                                       int M, int C, double* sums, void* workspace, size_t workspace_bytes, void* stream) {
    MAS_ENTER();
    if (!x || !sums || !workspace) MAS_FAIL(MAS_EINVAL, "bn_partial_sums: null argument");
    if (dy && !mean_rstd) MAS_FAIL(MAS_EINVAL, "bn_partial_sums: the backward sums need mean_rstd");
    if (dy && slope != 1.0f && !scale_shift) MAS_FAIL(MAS_EINVAL, "bn_partial_sums: the LeakyReLU mask needs scale_shift");
    if (dtype != MAS_F32 && dtype != MAS_BF16) MAS_FAIL(MAS_EINVAL, "bn_partial_sums: dtype %d", dtype);
    if (int rc = bn_check("bn_partial_sums", M, C)) return rc;
    if (workspace_bytes < mas_bn_workspace(M, C)) MAS_FAIL(MAS_EWORKSPACE, "bn_partial_sums: workspace too small");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    return dtype == MAS_F32 ? bn_partial_launch<float>(x, dy, mean_rstd, scale_shift, slope, M, C, sums, workspace, s)
                            : bn_partial_launch<bf16_t>(x, dy, mean_rstd, scale_shift, slope, M, C, sums, workspace, s);
}

extern "C" int mas_bn_partial_sums(const float* x, const float* dy, const float* mean_rstd, int M, int C, double* sums, void* workspace,
                                   size_t workspace_bytes, void* stream) {
    return mas_bn_partial_sums_act(x, dy, mean_rstd, nullptr, 1.0f, MAS_F32, M, C, sums, workspace, workspace_bytes, stream);
}

extern "C" int mas_bn_finalize(const double* sums, const float* gamma, const float* beta, float eps, float momentum, float* running_mean,
                               float* running_var, float* mean_rstd, float* scale_shift, int C, void* stream) {
    MAS_ENTER();
    if (!scale_shift || C <= 0) MAS_FAIL(MAS_EINVAL, "bn_finalize: null argument or C=%d", C);
    if (!sums && (!running_mean || !running_var)) MAS_FAIL(MAS_EINVAL, "bn_finalize: evaluation mode needs the running statistics");
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(mas_cdiv(C, NT)), dim3(NT), 0, reinterpret_cast<hipStream_t>(stream), sums, gamma, beta, eps, momentum,
                       running_mean, running_var, mean_rstd, scale_shift, C);
    MAS_CHECK_LAUNCH("bn_finalize");
    return MAS_OK;
}

extern "C" int mas_bn_apply_act(const void* x, const float* scale_shift, void* y, float slope, int dtype, int M, int C, void* stream) {
    MAS_ENTER();
    if (!x || !scale_shift || !y) MAS_FAIL(MAS_EINVAL, "bn_apply: null argument");
    if (dtype != MAS_F32 && dtype != MAS_BF16) MAS_FAIL(MAS_EINVAL, "bn_apply: dtype %d", dtype);
    if (int rc = bn_check("bn_apply", M, C)) return rc;
    const long long n4 = (long long)M * C / 4;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (dtype == MAS_F32) hipLaunchKernelGGL(bn_apply_kernel<float>, dim3(bn_grid(n4)), dim3(NT), 0, s, (const float*)x, scale_shift, (float*)y, slope, n4, C);
    else hipLaunchKernelGGL(bn_apply_kernel<bf16_t>, dim3(bn_grid(n4)), dim3(NT), 0, s, (const bf16_t*)x, scale_shift, (bf16_t*)y, slope, n4, C);
    MAS_CHECK_LAUNCH("bn_apply");
    return MAS_OK;
}

extern "C" int mas_bn_apply(const float* x, const float* scale_shift, float* y, int M, int C, void* stream) {
    return mas_bn_apply_act(x, scale_shift, y, 1.0f, MAS_F32, M, C, stream);
}

extern "C" int mas_bn_bwd_apply_act(const void* x, const void* dy, const float* mean_rstd, const float* gamma, const float* scale_shift, float slope,
                                    const double* sums, void* dx, int dtype, int M, int C, void* stream) {
    MAS_ENTER();
    if (!x || !dy || !mean_rstd || !sums || !dx) MAS_FAIL(MAS_EINVAL, "bn_bwd_apply: null argument");
    if (slope != 1.0f && !scale_shift) MAS_FAIL(MAS_EINVAL, "bn_bwd_apply: the LeakyReLU mask needs scale_shift");
    if (dtype != MAS_F32 && dtype != MAS_BF16) MAS_FAIL(MAS_EINVAL, "bn_bwd_apply: dtype %d", dtype);
    if (int rc = bn_check("bn_bwd_apply", M, C)) return rc;
    const long long n4 = (long long)M * C / 4;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (dtype == MAS_F32)
        hipLaunchKernelGGL(bn_bwd_apply_kernel<float>, dim3(bn_grid(n4)), dim3(NT), 0, s, (const float*)x, (const float*)dy, mean_rstd, gamma, scale_shift,
                           slope, sums, (float*)dx, n4, C);
    else
        hipLaunchKernelGGL(bn_bwd_apply_kernel<bf16_t>, dim3(bn_grid(n4)), dim3(NT), 0, s, (const bf16_t*)x, (const bf16_t*)dy, mean_rstd, gamma,
                           scale_shift, slope, sums, (bf16_t*)dx, n4, C);
    MAS_CHECK_LAUNCH("bn_bwd_apply");
    return MAS_OK;
}

extern "C" int mas_bn_bwd_apply(const float* x, const float* dy, const float* mean_rstd, const float* gamma, const double* sums, float* dx, int M,
                                int C, void* stream) {
    return mas_bn_bwd_apply_act(x, dy, mean_rstd, gamma, nullptr, 1.0f, sums, dx, MAS_F32, M, C, stream);
}
