// Streaming (HBM-bound) row operators of the transformer layer for gfx950:
//   * tanh-GELU forward / backward          (reference models/transformer.py:11-14, used by MLP :129)
//   * LayerNorm(eps) forward / backward with an optional fused residual add  y = res + LN(x)
//     (reference TransformerLayer.forward, models/transformer.py:197-210: pre-LN + sandwich-LN + residual)
// In eager PyTorch the GELU alone is ~9 elementwise launches forward and as many backward over the
// [B, S, 4d] tensor, and every LayerNorm is bracketed by dtype-cast copies; here each is one pass with
// 16-byte loads/stores and fp32 arithmetic, bf16 or fp32 storage on either side.
#include "mas_common.h"
#include <math.h>

namespace {

constexpr int NT = 256;

template <typename T> struct EwVec;     // 16-byte vectors
template <> struct EwVec<float> { static constexpr int N = 4; };
template <> struct EwVec<bf16_t> { static constexpr int N = 8; };

template <typename T, int N>
__device__ __forceinline__ void ld_vec(const T* p, float (&v)[N]) {
    u32x4 raw = *reinterpret_cast<const u32x4*>(p);
    const T* e = reinterpret_cast<const T*>(&raw);
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = (float)e[i];
}
template <typename T, int N>
__device__ __forceinline__ void st_vec(T* p, const float (&v)[N]) {
    u32x4 raw;
    T* e = reinterpret_cast<T*>(&raw);
#pragma unroll
    for (int i = 0; i < N; ++i) e[i] = (T)v[i];
    *reinterpret_cast<u32x4*>(p) = raw;
}

// N consecutive elements (8 or 16 bytes) as floats
template <typename T, int N>
__device__ __forceinline__ void ld_n(const T* p, float (&v)[N]) {
    typedef T VT __attribute__((ext_vector_type(N)));
    const VT raw = *reinterpret_cast<const VT*>(p);
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = (float)raw[i];
}
template <typename T, int N>
__device__ __forceinline__ void st_n(T* p, const float (&v)[N]) {
    typedef T VT __attribute__((ext_vector_type(N)));
    VT raw;
#pragma unroll
    for (int i = 0; i < N; ++i) raw[i] = (T)v[i];
    *reinterpret_cast<VT*>(p) = raw;
}

// tanh-GELU: y = 0.5 x (1 + tanh(u)),  u = k x (1 + c x^2);   dy/dx = 0.5 (1 + t) + 0.5 x (1 - t^2) k (1 + 3 c x^2)
constexpr float GK = 0.7978845608028654f, GC = 0.044715f;
template <bool EXACT>
__device__ __forceinline__ float tanh_f(float u) {
    if (EXACT) return tanhf(u);
    const float e = __expf(2.0f * u);                 // 1 - 2/(1+e^{2u}); saturates correctly at +-inf
    return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + e);
}

template <typename T>
__global__ __launch_bounds__(NT) void gelu_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, long long n) {
    constexpr int N = EwVec<T>::N;
    constexpr bool EXACT = sizeof(T) == 4;
    const long long nv = n / N, stride = (long long)gridDim.x * NT;
    for (long long i = (long long)blockIdx.x * NT + threadIdx.x; i < nv; i += stride) {
        float v[N];
        ld_vec<T, N>(x + i * N, v);
#pragma unroll
        for (int e = 0; e < N; ++e) {
            const float a = v[e], t = tanh_f<EXACT>(GK * a * (1.0f + GC * a * a));
            v[e] = 0.5f * a * (1.0f + t);
        }
        st_vec<T, N>(y + i * N, v);
    }
    if (blockIdx.x == 0 && (int)threadIdx.x < (int)(n - nv * N)) {   // ragged tail
        const long long i = nv * N + threadIdx.x;
        const float a = (float)x[i], t = tanh_f<EXACT>(GK * a * (1.0f + GC * a * a));
        y[i] = (T)(0.5f * a * (1.0f + t));
    }
}

// CS: ALSO leave the column sums of the dx written (round 6): dx is the grad_output of the Linear that produced x (`lin1`, reference
// models/transformer.py:125,129), whose bias gradient is grad_output.sum(0) -- otherwise a mas_colsum pass over the 100 MB tensor.  The
// grid-stride step (gridDim * NT * N elements) is then a multiple of the row length, so a thread meets the SAME N columns in every iteration
// and keeps their sums in registers; partial[gridDim * NT * N] viewed as [gridDim * NT * N / cols][cols] holds one value per (thread, column),
// folded by fold_rows_kernel in a fixed order.  Sums of dx AS STORED (rounded to T), like mas_colsum's.  One template for both forms.
template <typename T, bool CS>
__global__ __launch_bounds__(NT) void gelu_bwd_kernel(const T* __restrict__ x, const T* __restrict__ dy, T* __restrict__ dx, long long n,
                                                      float* __restrict__ partial) {
    constexpr int N = EwVec<T>::N;
    constexpr bool EXACT = sizeof(T) == 4;
    auto grad = [](float a, float g) {
        const float a2 = a * a, t = tanh_f<EXACT>(GK * a * (1.0f + GC * a2));
        return g * (0.5f * (1.0f + t) + 0.5f * a * (1.0f - t * t) * GK * (1.0f + 3.0f * GC * a2));
    };
    float acc[CS ? N : 1];
#pragma unroll
    for (int e = 0; e < (CS ? N : 1); ++e) acc[e] = 0.0f;
    const long long nv = n / N, stride = (long long)gridDim.x * NT;
    const long long first = (long long)blockIdx.x * NT + threadIdx.x;
    for (long long i = first; i < nv; i += stride) {
        float v[N], g[N];
        ld_vec<T, N>(x + i * N, v);
        ld_vec<T, N>(dy + i * N, g);
#pragma unroll
        for (int e = 0; e < N; ++e) v[e] = grad(v[e], g[e]);
        if constexpr (CS) {
#pragma unroll
            for (int e = 0; e < N; ++e) acc[e] += (float)(T)v[e];
        }
        st_vec<T, N>(dx + i * N, v);
    }
    if constexpr (CS) {
#pragma unroll
        for (int e = 0; e < N; ++e) partial[first * N + e] = acc[e];
    } else if (blockIdx.x == 0 && (int)threadIdx.x < (int)(n - nv * N)) {
        const long long i = nv * N + threadIdx.x;
        dx[i] = (T)grad((float)x[i], (float)dy[i]);
    }
}

// ---------------------------------------------------------------------------------------------------------
// LayerNorm.  One wave per row; a lane owns the same columns of every row (column = (k*64 + lane)*VEC + e for
// k < KMAX), the row lives in registers, statistics are exact two-pass (mean, then centred sum of squares) in
// fp32 with wave-level DPP/permute reductions -- no LDS, no second read of x.
constexpr int LN_KMAX = 4;              // D <= 64 * VEC * 4  (2048 bf16 / 1024 fp32 columns)

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

template <typename TI, typename TO>
__global__ __launch_bounds__(NT) void layernorm_fwd_kernel(const TI* __restrict__ x, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, const TO* __restrict__ res,
                                                           TO* __restrict__ y, float* __restrict__ mean_rstd, int rows, int D, float eps) {
    constexpr int VEC = 16 / (int)sizeof(TI) < 16 / (int)sizeof(TO) ? 16 / (int)sizeof(TI) : 16 / (int)sizeof(TO);   // common vector width
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nvec = D / VEC;
    for (int row = blockIdx.x * (NT / 64) + wave; row < rows; row += gridDim.x * (NT / 64)) {
        const TI* xr = x + (size_t)row * D;
        float v[LN_KMAX][VEC];
        float s = 0.0f;
#pragma unroll
        for (int k = 0; k < LN_KMAX; ++k) {
            const int c = k * 64 + lane;
            if (c < nvec) {
                ld_n<TI, VEC>(xr + c * VEC, v[k]);
#pragma unroll
                for (int e = 0; e < VEC; ++e) s += v[k][e];
            } else {
#pragma unroll
                for (int e = 0; e < VEC; ++e) v[k][e] = 0.0f;
            }
        }
        const float mean = wave_sum(s) / (float)D;
        float q = 0.0f;
#pragma unroll
        for (int k = 0; k < LN_KMAX; ++k)
            if (k * 64 + lane < nvec)
#pragma unroll
                for (int e = 0; e < VEC; ++e) { const float d = v[k][e] - mean; q += d * d; }
        const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)D + eps);
        if (lane == 0 && mean_rstd) { mean_rstd[2 * (size_t)row] = mean; mean_rstd[2 * (size_t)row + 1] = rstd; }
        TO* yr = y + (size_t)row * D;
#pragma unroll
        for (int k = 0; k < LN_KMAX; ++k) {
            const int c = k * 64 + lane;
            if (c < nvec) {
                float ga[VEC], be[VEC], o[VEC];
                ld_n<float, VEC>(gamma + c * VEC, ga);
                ld_n<float, VEC>(beta + c * VEC, be);
#pragma unroll
                for (int e = 0; e < VEC; ++e) o[e] = (v[k][e] - mean) * rstd * ga[e] + be[e];
                if (res) {
                    float rr[VEC];
                    ld_n<TO, VEC>(res + (size_t)row * D + c * VEC, rr);
#pragma unroll
                    for (int e = 0; e < VEC; ++e) o[e] += rr[e];
                }
                st_n<TO, VEC>(yr + c * VEC, o);
            }
        }
    }
}

// backward: g = dy*gamma, xhat = (x-mean)*rstd, dx = rstd*(g - mean_D(g) - xhat*mean_D(g*xhat));
// per-block partial sums of dgamma = sum_rows dy*xhat and dbeta = sum_rows dy: partial[blk][NP][D], NP = 2, or 3 with CS: the third row
// is the column sum of dx AS STORED (rounded to TI) -- the bias gradient of the Linear layer whose output this LayerNorm normalises
// (out_proj / lin2 in front of the sandwich LayerNorms, reference transformer.py:201-203,207-209), which otherwise is a mas_colsum
// pass over dx plus its fold launch.
template <typename TI, typename TO, bool CS>
__global__ __launch_bounds__(NT) void layernorm_bwd_kernel(const TI* __restrict__ x, const TO* __restrict__ dy,
                                                           const float* __restrict__ gamma, const float* __restrict__ mean_rstd,
                                                           TI* __restrict__ dx, float* __restrict__ partial, int rows, int D,
                                                           const TI* __restrict__ dx_add) {
    constexpr int NP = CS ? 3 : 2;
    constexpr int VEC = 16 / (int)sizeof(TI) < 16 / (int)sizeof(TO) ? 16 / (int)sizeof(TI) : 16 / (int)sizeof(TO);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nvec = D / VEC;
    float ag[LN_KMAX][VEC], ab[LN_KMAX][VEC], ac[CS ? LN_KMAX : 1][VEC];
#pragma unroll
    for (int k = 0; k < LN_KMAX; ++k)
#pragma unroll
        for (int e = 0; e < VEC; ++e) { ag[k][e] = 0.0f; ab[k][e] = 0.0f; if (CS) ac[k][e] = 0.0f; }
    for (int row = blockIdx.x * (NT / 64) + wave; row < rows; row += gridDim.x * (NT / 64)) {
        const float mean = mean_rstd[2 * (size_t)row], rstd = mean_rstd[2 * (size_t)row + 1];
        const TI* xr = x + (size_t)row * D;
        const TO* gr = dy + (size_t)row * D;
        float xh[LN_KMAX][VEC], g[LN_KMAX][VEC];
        float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
        for (int k = 0; k < LN_KMAX; ++k) {
            const int c = k * 64 + lane;
            if (c < nvec) {
                float xv[VEC], dv[VEC], ga[VEC];
                ld_n<TI, VEC>(xr + c * VEC, xv);
                ld_n<TO, VEC>(gr + c * VEC, dv);
                ld_n<float, VEC>(gamma + c * VEC, ga);
#pragma unroll
                for (int e = 0; e < VEC; ++e) {
                    const float d = dv[e];
                    xh[k][e] = (xv[e] - mean) * rstd;
                    g[k][e] = d * ga[e];
                    s1 += g[k][e]; s2 += g[k][e] * xh[k][e];
                    ag[k][e] += d * xh[k][e]; ab[k][e] += d;
                }
            }
        }
        const float m1 = wave_sum(s1) / (float)D, m2 = wave_sum(s2) / (float)D;
        TI* dr = dx + (size_t)row * D;
#pragma unroll
        for (int k = 0; k < LN_KMAX; ++k) {
            const int c = k * 64 + lane;
            if (c < nvec) {
                float o[VEC];
#pragma unroll
                for (int e = 0; e < VEC; ++e) o[e] = rstd * (g[k][e] - m1 - xh[k][e] * m2);
                if (dx_add) {                                  // the gradient that reached x along its skip connection
                    float sk[VEC];
                    ld_n<TI, VEC>(dx_add + (size_t)row * D + c * VEC, sk);
#pragma unroll
                    for (int e = 0; e < VEC; ++e) o[e] += sk[e];
                }
                if (CS) {
#pragma unroll
                    for (int e = 0; e < VEC; ++e) ac[k][e] += (float)(TI)o[e];
                }
                st_n<TI, VEC>(dr + c * VEC, o);
            }
        }
    }
    // fixed-order reduction over the block's waves through LDS, then one partial row per block
    __shared__ float red[(NT / 64)][NP][64 * 8 + 1];           // per k: [wave][gamma|beta|dx][lane*VEC + e]
#pragma unroll
    for (int k = 0; k < LN_KMAX; ++k) {
        __syncthreads();
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            red[wave][0][lane * VEC + e] = ag[k][e]; red[wave][1][lane * VEC + e] = ab[k][e];
            if (CS) red[wave][NP - 1][lane * VEC + e] = ac[k][e];
        }
        __syncthreads();
        for (int i = threadIdx.x; i < NP * 64 * VEC; i += NT) {
            const int which = i / (64 * VEC), j = i % (64 * VEC);
            const int col = k * 64 * VEC + j;
            if (col < D) {
                float a = 0.0f;
#pragma unroll
                for (int w = 0; w < NT / 64; ++w) a += red[w][which][j];
                partial[((size_t)blockIdx.x * NP + which) * D + col] = a;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// The sandwich LayerNorm + residual of one sub-block and the pre-LayerNorm of the NEXT one as ONE pass (round 6):
//   xnew = res + LN1(h)        (h: the attention / MLP output; reference models/transformer.py:201-203, 207-209)
//   y2   = LN2(xnew)           (:197-198 of the next sub-block / layer, or the final LayerNorm)
// The row stays in registers between the two normalisations: xnew is written once and never read back (the separate launches move
// 16 B per element, this one 12).  The BACKWARD of the pair stays two launches: fused, its five accumulator arrays (dgamma / dbeta of both
// LayerNorms + the producer's bias gradient) take the kernel to 190 VGPRs = two waves per SIMD, and a latency-bound stream at that occupancy
// ran 80 us against 34 + 25 for the two launches (profiles/r06_ln_pair.txt; docs/history/experiments/r6_ln_pair_bwd.patch).  Same arithmetic, same order and -- xnew being rounded to its storage type before the second
// statistics -- the same values as the two launches of layernorm_fwd_kernel, bit for bit.  VEC = 4 (one of the three types is fp32).
template <typename TI, typename TO, typename TY>
__global__ __launch_bounds__(NT) void layernorm_pair_fwd_kernel(const TI* __restrict__ h, const float* __restrict__ g1, const float* __restrict__ b1,
                                                                const TO* __restrict__ res, const float* __restrict__ g2, const float* __restrict__ b2,
                                                                TO* __restrict__ xnew, TY* __restrict__ y2, float* __restrict__ mr1,
                                                                float* __restrict__ mr2, int rows, int D, float eps1, float eps2) {
    constexpr int VEC = 4;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nvec = D / VEC;
    for (int row = blockIdx.x * (NT / 64) + wave; row < rows; row += gridDim.x * (NT / 64)) {
        const TI* hr = h + (size_t)row * D;
        float v[LN_KMAX][VEC];
        float s = 0.0f;
#pragma unroll
        for (int k = 0; k < LN_KMAX; ++k) {
            const int c = k * 64 + lane;
            if (c < nvec) {
                ld_n<TI, VEC>(hr + c * VEC, v[k]);
#pragma unroll
                for (int e = 0; e < VEC; ++e) s += v[k][e];
            } else {
#pragma unroll
                for (int e = 0; e < VEC; ++e) v[k][e] = 0.0f;
            }
        }
        const float mean1 = wave_sum(s) / (float)D;
        float q = 0.0f;
#pragma unroll
        for (int k = 0; k < LN_KMAX; ++k)
            if (k * 64 + lane < nvec)
#pragma unroll
                for (int e = 0; e < VEC; ++e) { const float d = v[k][e] - mean1; q += d * d; }
        const float rstd1 = 1.0f / sqrtf(wave_sum(q) / (float)D + eps1);
        if (lane == 0) { mr1[2 * (size_t)row] = mean1; mr1[2 * (size_t)row + 1] = rstd1; }
        s = 0.0f;
#pragma unroll
        for (int k = 0; k < LN_KMAX; ++k) {
            const int c = k * 64 + lane;
            if (c < nvec) {
                float ga[VEC], be[VEC], rr[VEC];
                ld_n<float, VEC>(g1 + c * VEC, ga);
                ld_n<float, VEC>(b1 + c * VEC, be);
                ld_n<TO, VEC>(res + (size_t)row * D + c * VEC, rr);
#pragma unroll
                for (int e = 0; e < VEC; ++e) {
                    float o = (v[k][e] - mean1) * rstd1 * ga[e] + be[e];
                    o += rr[e];
                    v[k][e] = (float)(TO)o;                    // what the second LayerNorm reads when the two are separate launches
                    s += v[k][e];
                }
                st_n<TO, VEC>(xnew + (size_t)row * D + c * VEC, v[k]);
            }
        }
        const float mean2 = wave_sum(s) / (float)D;
        q = 0.0f;
#pragma unroll
        for (int k = 0; k < LN_KMAX; ++k)
            if (k * 64 + lane < nvec)
#pragma unroll
                for (int e = 0; e < VEC; ++e) { const float d = v[k][e] - mean2; q += d * d; }
        const float rstd2 = 1.0f / sqrtf(wave_sum(q) / (float)D + eps2);
        if (lane == 0) { mr2[2 * (size_t)row] = mean2; mr2[2 * (size_t)row + 1] = rstd2; }
#pragma unroll
        for (int k = 0; k < LN_KMAX; ++k) {
            const int c = k * 64 + lane;
            if (c < nvec) {
                float ga[VEC], be[VEC], o[VEC];
                ld_n<float, VEC>(g2 + c * VEC, ga);
                ld_n<float, VEC>(b2 + c * VEC, be);
#pragma unroll
                for (int e = 0; e < VEC; ++e) o[e] = (v[k][e] - mean2) * rstd2 * ga[e] + be[e];
                st_n<TY, VEC>(y2 + (size_t)row * D + c * VEC, o);
            }
        }
    }
}

// dgamma / dbeta = fixed-order sum of the per-block partials [nblk][NP][D] in ONE launch: grid (D/32, NP) -- work-group (cb, which) owns
// 32 columns of dgamma (which = 0), dbeta (1) or the column sum of dx (2, when asked for); 8 threads x float4 cover the 128-byte
// column segment of a partial row, 32 row groups take rows rg, rg + 32, ... (four 16-byte loads in flight), LDS folds the row groups
// in a fixed order: bitwise run-to-run deterministic.  (Round 1 ran D/32 work-groups of scalar loads over 8 MB of partials: 40 us per
// LayerNorm backward; rounds 2-4 two launches -- 8 slices, then their sum -- 2 x 5.4 us + the dependent-launch gap, 196 launches per
// MakeAScene step.  64-96 work-groups of 16-byte loads read the same 8-12 MB in one.)
// The same kernel folds the column-sum slices (colsum below): row_stride / which_stride describe the table.
__global__ __launch_bounds__(NT) void fold_rows_kernel(const float* __restrict__ partial, int nblk, int D, long long row_stride,
                                                       long long which_stride, float* __restrict__ out_g, float* __restrict__ out_b,
                                                       float* __restrict__ out_c) {
    __shared__ float red[32][33];
    const int cq = threadIdx.x & 7, rg = threadIdx.x >> 3, which = blockIdx.y;
    const int col = blockIdx.x * 32 + cq * 4;
    float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
    if (col < D) {
        const float* src = partial + (size_t)which * which_stride + col;
        int k = rg;
        for (; k + 96 < nblk; k += 128) {
            float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const float4*>(src + (size_t)(k + 32 * u) * row_stride);
#pragma unroll
            for (int u = 0; u < 4; ++u) { a0 += v[u].x; a1 += v[u].y; a2 += v[u].z; a3 += v[u].w; }
        }
        for (; k < nblk; k += 32) {
            const float4 v = *reinterpret_cast<const float4*>(src + (size_t)k * row_stride);
            a0 += v.x; a1 += v.y; a2 += v.z; a3 += v.w;
        }
    }
    red[rg][cq * 4 + 0] = a0; red[rg][cq * 4 + 1] = a1; red[rg][cq * 4 + 2] = a2; red[rg][cq * 4 + 3] = a3;
    __syncthreads();
    if (threadIdx.x < 32) {
        const int c = blockIdx.x * 32 + threadIdx.x;
        float t = 0.0f;
#pragma unroll
        for (int r = 0; r < 32; ++r) t += red[r][threadIdx.x];
        if (c < D) (which == 0 ? out_g : which == 1 ? out_b : out_c)[c] = t;
    }
}

// Column sums of a [rows][cols] matrix, two stages.  Stage 1: work-group (cb, sl) = one wave, one 16-byte vector of columns per
// lane, rows sl, sl + nsl, ... (8 loads in flight per lane) -> tmp[sl][cols]; stage 2 (fold_rows_kernel above): 32 columns x 32 row
// groups per work-group add the nsl <= 128 partial rows (fixed order: bitwise run-to-run deterministic).
constexpr int CS_NT = 64, CS_MAX_SLICES = 128;
template <typename T>
__global__ __launch_bounds__(CS_NT) void colsum_partial(const T* __restrict__ x, int rows, int cols, int nsl, float* __restrict__ tmp) {
    constexpr int N = EwVec<T>::N;
    const int c0 = (blockIdx.x * CS_NT + threadIdx.x) * N, sl = blockIdx.y;
    if (c0 >= cols) return;
    float acc[N];
#pragma unroll
    for (int e = 0; e < N; ++e) acc[e] = 0.0f;
    int r = sl;
    for (; r + 7 * nsl < rows; r += 8 * nsl) {
        u32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const u32x4*>(x + (size_t)(r + u * nsl) * cols + c0);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const T* e_ = reinterpret_cast<const T*>(&v[u]);
#pragma unroll
            for (int e = 0; e < N; ++e) acc[e] += (float)e_[e];
        }
    }
    for (; r < rows; r += nsl) {
        const u32x4 v = *reinterpret_cast<const u32x4*>(x + (size_t)r * cols + c0);
        const T* e_ = reinterpret_cast<const T*>(&v);
#pragma unroll
        for (int e = 0; e < N; ++e) acc[e] += (float)e_[e];
    }
#pragma unroll
    for (int e = 0; e < N; ++e) tmp[(size_t)sl * cols + c0 + e] = acc[e];
}
int colsum_slices(int rows, int cols, int vec) {
    const int col_blocks = mas_cdiv(cols, CS_NT * vec);
    int nsl = mas_cdiv(4 * mas_num_cus(), col_blocks > 0 ? col_blocks : 1);      // ~4 one-wave work-groups per CU
    if (nsl > CS_MAX_SLICES) nsl = CS_MAX_SLICES;
    if (nsl > mas_cdiv(rows, 8)) nsl = mas_cdiv(rows, 8);                         // at least 8 rows per work-group
    return nsl < 1 ? 1 : nsl;
}

int ln_blocks(int rows, int per_cu = 4) {
    int nb = mas_cdiv(rows, NT / 64);
    const int cap = per_cu * mas_num_cus();
    return nb < cap ? nb : cap;
}

template <typename T>
int gelu_launch(bool bwd, const void* x, const void* dy, void* out, long long n, hipStream_t s) {
    constexpr int N = EwVec<T>::N;
    long long nb = (n / N + NT - 1) / NT;
    const long long cap = 16LL * mas_num_cus();
    if (nb > cap) nb = cap;
    if (nb < 1) nb = 1;
    if (bwd) hipLaunchKernelGGL((gelu_bwd_kernel<T, false>), dim3((unsigned)nb), dim3(NT), 0, s, (const T*)x, (const T*)dy, (T*)out, n, (float*)nullptr);
    else hipLaunchKernelGGL(gelu_fwd_kernel<T>, dim3((unsigned)nb), dim3(NT), 0, s, (const T*)x, (T*)out, n);
    MAS_CHECK_LAUNCH(bwd ? "gelu_tanh_bwd" : "gelu_tanh_fwd");
    return MAS_OK;
}

int ln_check(int in_dtype, int out_dtype, int rows, int D, const char* what) {
    if (rows <= 0 || D <= 0) MAS_FAIL(MAS_EINVAL, "%s: bad shape rows=%d D=%d", what, rows, D);
    if ((in_dtype != MAS_F32 && in_dtype != MAS_BF16) || (out_dtype != MAS_F32 && out_dtype != MAS_BF16))
        MAS_FAIL(MAS_EUNSUPPORTED, "%s: dtypes %d -> %d", what, in_dtype, out_dtype);
    const int vec = (in_dtype == MAS_BF16 && out_dtype == MAS_BF16) ? 8 : 4;
    if (D % vec != 0 || D > 64 * vec * LN_KMAX)
        MAS_FAIL(MAS_EUNSUPPORTED, "%s: D=%d must be a multiple of %d and <= %d for these dtypes", what, D, vec, 64 * vec * LN_KMAX);
    return MAS_OK;
}

}  // namespace

extern "C" int mas_gelu_tanh_fwd(const void* x, void* y, int dtype, long long n, void* stream) {
    MAS_ENTER();
    if (!x || !y || n <= 0) MAS_FAIL(MAS_EINVAL, "gelu_tanh_fwd: null argument or n=%lld", n);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (dtype == MAS_BF16) return gelu_launch<bf16_t>(false, x, nullptr, y, n, s);
    if (dtype == MAS_F32) return gelu_launch<float>(false, x, nullptr, y, n, s);
    MAS_FAIL(MAS_EUNSUPPORTED, "gelu_tanh_fwd: dtype %d", dtype);
}

extern "C" int mas_gelu_tanh_bwd(const void* x, const void* dy, void* dx, int dtype, long long n, void* stream) {
    MAS_ENTER();
    if (!x || !dy || !dx || n <= 0) MAS_FAIL(MAS_EINVAL, "gelu_tanh_bwd: null argument or n=%lld", n);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (dtype == MAS_BF16) return gelu_launch<bf16_t>(true, x, dy, dx, n, s);
    if (dtype == MAS_F32) return gelu_launch<float>(true, x, dy, dx, n, s);
    MAS_FAIL(MAS_EUNSUPPORTED, "gelu_tanh_bwd: dtype %d", dtype);
}

namespace {
// grid of gelu_bwd_colsum: gridDim * NT * N must be a multiple of cols; ~4 work-groups per CU (the per-thread sums make the partial table
// gridDim * NT * N floats: 8 MB at 1024 work-groups of bf16)
int gelu_cs_grid(int cols, int vec) {
    const long long per_block = (long long)NT * vec;
    long long a = cols, b = per_block;
    while (b) { const long long t = a % b; a = b; b = t; }
    const long long unit = cols / a;                                   // smallest block count whose elements are whole rows
    long long want = 4LL * mas_num_cus();
    long long g = want / unit * unit;
    if (g < unit) g = unit;
    return g > 65535 ? 0 : (int)g;
}
}  // namespace

extern "C" size_t mas_gelu_tanh_bwd_colsum_workspace(int dtype, int cols) {
    const int vec = dtype == MAS_BF16 ? 8 : 4;
    if (cols <= 0 || cols % vec) return 0;
    const int g = gelu_cs_grid(cols, vec);
    return g ? (size_t)g * NT * vec * sizeof(float) : 0;
}

extern "C" int mas_gelu_tanh_bwd_colsum(const void* x, const void* dy, void* dx, float* dx_colsum, int dtype, long long rows, int cols, void* workspace,
                                        size_t workspace_bytes, void* stream) {
    MAS_ENTER();
    if (!x || !dy || !dx || !dx_colsum || !workspace || rows <= 0 || cols <= 0) MAS_FAIL(MAS_EINVAL, "gelu_tanh_bwd_colsum: null argument or bad shape");
    if (dtype != MAS_BF16 && dtype != MAS_F32) MAS_FAIL(MAS_EUNSUPPORTED, "gelu_tanh_bwd_colsum: dtype %d", dtype);
    const int vec = dtype == MAS_BF16 ? 8 : 4;
    if (cols % vec) MAS_FAIL(MAS_EUNSUPPORTED, "gelu_tanh_bwd_colsum: cols=%d must be a multiple of %d", cols, vec);
    const int g = gelu_cs_grid(cols, vec);
    if (!g || workspace_bytes < mas_gelu_tanh_bwd_colsum_workspace(dtype, cols)) MAS_FAIL(MAS_EINVAL, "gelu_tanh_bwd_colsum: workspace too small");
    const long long n = rows * cols, slots = (long long)g * NT;       // (threads past the tensor's end write zero sums)
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    float* partial = reinterpret_cast<float*>(workspace);
    if (dtype == MAS_BF16) hipLaunchKernelGGL((gelu_bwd_kernel<bf16_t, true>), dim3(g), dim3(NT), 0, s, (const bf16_t*)x, (const bf16_t*)dy, (bf16_t*)dx, n, partial);
    else hipLaunchKernelGGL((gelu_bwd_kernel<float, true>), dim3(g), dim3(NT), 0, s, (const float*)x, (const float*)dy, (float*)dx, n, partial);
    const int nrow = (int)(slots * vec / cols);                        // the partial table as [nrow][cols]
    hipLaunchKernelGGL(fold_rows_kernel, dim3(mas_cdiv(cols, 32), 1), dim3(NT), 0, s, partial, nrow, cols, (long long)cols, 0LL, dx_colsum, dx_colsum, dx_colsum);
    MAS_CHECK_LAUNCH("gelu_tanh_bwd_colsum");
    return MAS_OK;
}

extern "C" int mas_layernorm_fwd(const void* x, const float* gamma, const float* beta, const void* residual, void* y,
                                 float* mean_rstd, int in_dtype, int out_dtype, int rows, int D, float eps, void* stream) {
    MAS_ENTER();
    if (!x || !gamma || !beta || !y) MAS_FAIL(MAS_EINVAL, "layernorm_fwd: null argument");
    if (int rc = ln_check(in_dtype, out_dtype, rows, D, "layernorm_fwd")) return rc;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    // (64 VGPRs: eight work-groups per CU are resident, and at rows / 2048 = 1.5 rows per wave nearly every row is in flight at once:
    //  the sandwich forward 22-23 -> 20.4 us at 12288 x 1024, the fp32 -> bf16 forward unchanged at 17.8; tools/probes/ln_probe.py)
    static const int per_cu = mas_env_int("MAS_LN_FWD_BLOCKS_PER_CU", 8);
    const dim3 grid(ln_blocks(rows, per_cu)), block(NT);
#define MAS_LN_FWD(TI, TO) hipLaunchKernelGGL((layernorm_fwd_kernel<TI, TO>), grid, block, 0, s, (const TI*)x, gamma, beta, (const TO*)residual, (TO*)y, mean_rstd, rows, D, eps)
    if (in_dtype == MAS_BF16 && out_dtype == MAS_BF16) MAS_LN_FWD(bf16_t, bf16_t);
    else if (in_dtype == MAS_BF16) MAS_LN_FWD(bf16_t, float);
    else if (out_dtype == MAS_BF16) MAS_LN_FWD(float, bf16_t);
    else MAS_LN_FWD(float, float);
#undef MAS_LN_FWD
    MAS_CHECK_LAUNCH("layernorm_fwd");
    return MAS_OK;
}

extern "C" size_t mas_layernorm_bwd_workspace(int rows, int D) {
    if (rows <= 0 || D <= 0) return 0;
    return (size_t)ln_blocks(rows) * 3 * (size_t)D * sizeof(float);                    // the per-block partials (dgamma, dbeta, column sum of dx)
}

extern "C" int mas_layernorm_bwd_colsum(const void* x, const void* dy, const float* gamma, const float* mean_rstd, const void* dx_add,
                                        void* dx, float* dgamma, float* dbeta, float* dx_colsum, int in_dtype, int out_dtype, int rows, int D,
                                        void* workspace, size_t workspace_bytes, void* stream) {
    MAS_ENTER();
    if (!x || !dy || !gamma || !mean_rstd || !dx || !dgamma || !dbeta || !workspace) MAS_FAIL(MAS_EINVAL, "layernorm_bwd: null argument");
    if (int rc = ln_check(in_dtype, out_dtype, rows, D, "layernorm_bwd")) return rc;
    if (workspace_bytes < mas_layernorm_bwd_workspace(rows, D)) MAS_FAIL(MAS_EINVAL, "layernorm_bwd: workspace too small");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    // (the column-sum variant holds 16 more accumulators: 142 VGPRs, three waves per SIMD -- so three work-groups per CU, all resident)
    const int nblk = ln_blocks(rows, dx_colsum ? 3 : 4);
    float* partial = reinterpret_cast<float*>(workspace);
#define MAS_LN_BWD(TI, TO)                                                                                                                 \
    do {                                                                                                                                   \
        if (dx_colsum) hipLaunchKernelGGL((layernorm_bwd_kernel<TI, TO, true>), dim3(nblk), dim3(NT), 0, s, (const TI*)x, (const TO*)dy, gamma, \
                                          mean_rstd, (TI*)dx, partial, rows, D, (const TI*)dx_add);                                        \
        else hipLaunchKernelGGL((layernorm_bwd_kernel<TI, TO, false>), dim3(nblk), dim3(NT), 0, s, (const TI*)x, (const TO*)dy, gamma,      \
                                mean_rstd, (TI*)dx, partial, rows, D, (const TI*)dx_add);                                                  \
    } while (0)
    if (in_dtype == MAS_BF16 && out_dtype == MAS_BF16) MAS_LN_BWD(bf16_t, bf16_t);
    else if (in_dtype == MAS_BF16) MAS_LN_BWD(bf16_t, float);
    else if (out_dtype == MAS_BF16) MAS_LN_BWD(float, bf16_t);
    else MAS_LN_BWD(float, float);
#undef MAS_LN_BWD
    const int np = dx_colsum ? 3 : 2;
    hipLaunchKernelGGL(fold_rows_kernel, dim3(mas_cdiv(D, 32), np), dim3(NT), 0, s, partial, nblk, D, (long long)np * D, (long long)D, dgamma, dbeta,
                       dx_colsum);
    MAS_CHECK_LAUNCH("layernorm_bwd");
    return MAS_OK;
}

extern "C" int mas_layernorm_bwd_add(const void* x, const void* dy, const float* gamma, const float* mean_rstd, const void* dx_add,
                                     void* dx, float* dgamma, float* dbeta, int in_dtype, int out_dtype, int rows, int D,
                                     void* workspace, size_t workspace_bytes, void* stream) {
    return mas_layernorm_bwd_colsum(x, dy, gamma, mean_rstd, dx_add, dx, dgamma, dbeta, nullptr, in_dtype, out_dtype, rows, D, workspace,
                                    workspace_bytes, stream);
}

extern "C" int mas_layernorm_bwd(const void* x, const void* dy, const float* gamma, const float* mean_rstd, void* dx,
                                 float* dgamma, float* dbeta, int in_dtype, int out_dtype, int rows, int D,
                                 void* workspace, size_t workspace_bytes, void* stream) {
    return mas_layernorm_bwd_add(x, dy, gamma, mean_rstd, nullptr, dx, dgamma, dbeta, in_dtype, out_dtype, rows, D, workspace, workspace_bytes, stream);
}

namespace {
int ln_pair_check(int h_dtype, int x_dtype, int y_dtype, int rows, int D, const char* what) {
    if (rows <= 0 || D <= 0) MAS_FAIL(MAS_EINVAL, "%s: bad shape rows=%d D=%d", what, rows, D);
    const bool mixed = h_dtype == MAS_BF16 && x_dtype == MAS_F32 && y_dtype == MAS_BF16;       // the autocast transformer: fp32 residual stream
    const bool f32 = h_dtype == MAS_F32 && x_dtype == MAS_F32 && y_dtype == MAS_F32;
    if (!mixed && !f32) MAS_FAIL(MAS_EUNSUPPORTED, "%s: dtypes (%d, %d, %d): (bf16, fp32, bf16) or all fp32", what, h_dtype, x_dtype, y_dtype);
    if (D % 4 != 0 || D > 64 * 4 * LN_KMAX) MAS_FAIL(MAS_EUNSUPPORTED, "%s: D=%d must be a multiple of 4 and <= %d", what, D, 64 * 4 * LN_KMAX);
    return MAS_OK;
}
}  // namespace

extern "C" int mas_layernorm_pair_fwd(const void* h, const float* gamma1, const float* beta1, const void* residual, const float* gamma2,
                                      const float* beta2, void* xnew, void* y2, float* mean_rstd1, float* mean_rstd2, int h_dtype, int x_dtype,
                                      int y_dtype, int rows, int D, float eps1, float eps2, void* stream) {
    MAS_ENTER();
    if (!h || !gamma1 || !beta1 || !residual || !gamma2 || !beta2 || !xnew || !y2 || !mean_rstd1 || !mean_rstd2)
        MAS_FAIL(MAS_EINVAL, "layernorm_pair_fwd: null argument");
    if (int rc = ln_pair_check(h_dtype, x_dtype, y_dtype, rows, D, "layernorm_pair_fwd")) return rc;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    static const int per_cu = mas_env_int("MAS_LN_PAIR_FWD_BLOCKS_PER_CU", 8);
    const dim3 grid(ln_blocks(rows, per_cu)), block(NT);
    if (h_dtype == MAS_BF16)
        hipLaunchKernelGGL((layernorm_pair_fwd_kernel<bf16_t, float, bf16_t>), grid, block, 0, s, (const bf16_t*)h, gamma1, beta1, (const float*)residual,
                           gamma2, beta2, (float*)xnew, (bf16_t*)y2, mean_rstd1, mean_rstd2, rows, D, eps1, eps2);
    else
        hipLaunchKernelGGL((layernorm_pair_fwd_kernel<float, float, float>), grid, block, 0, s, (const float*)h, gamma1, beta1, (const float*)residual,
                           gamma2, beta2, (float*)xnew, (float*)y2, mean_rstd1, mean_rstd2, rows, D, eps1, eps2);
    MAS_CHECK_LAUNCH("layernorm_pair_fwd");
    return MAS_OK;
}

extern "C" size_t mas_colsum_workspace(int rows, int cols) {
    if (rows <= 0 || cols <= 0) return 0;
    return (size_t)CS_MAX_SLICES * cols * sizeof(float);
}

extern "C" int mas_colsum(const void* x, int dtype, int rows, int cols, float* out, void* workspace, size_t workspace_bytes, void* stream) {
    if (!x || !out || !workspace) MAS_FAIL(MAS_EINVAL, "colsum: null argument");
    if (rows <= 0 || cols <= 0) MAS_FAIL(MAS_EINVAL, "colsum: bad shape rows=%d cols=%d", rows, cols);
    if (dtype != MAS_BF16 && dtype != MAS_F32) MAS_FAIL(MAS_EUNSUPPORTED, "colsum: dtype %d (bf16 or fp32)", dtype);
    const int vec = dtype == MAS_BF16 ? 8 : 4;
    if (cols % vec) MAS_FAIL(MAS_EUNSUPPORTED, "colsum: cols=%d must be a multiple of %d", cols, vec);
    if (workspace_bytes < mas_colsum_workspace(rows, cols)) MAS_FAIL(MAS_EINVAL, "colsum: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    int nsl = colsum_slices(rows, cols, vec);
    const int cap = (int)(workspace_bytes / ((size_t)cols * sizeof(float)));
    if (nsl > cap) nsl = cap;
    float* tmp = (float*)workspace;
    if (dtype == MAS_BF16)
        hipLaunchKernelGGL(colsum_partial<bf16_t>, dim3(mas_cdiv(cols, CS_NT * 8), nsl), dim3(CS_NT), 0, s, (const bf16_t*)x, rows, cols, nsl, tmp);
    else
        hipLaunchKernelGGL(colsum_partial<float>, dim3(mas_cdiv(cols, CS_NT * 4), nsl), dim3(CS_NT), 0, s, (const float*)x, rows, cols, nsl, tmp);
    hipLaunchKernelGGL(fold_rows_kernel, dim3(mas_cdiv(cols, 32), 1), dim3(NT), 0, s, tmp, nsl, cols, (long long)cols, 0LL, out, out, out);
    MAS_CHECK_LAUNCH("colsum");
    return MAS_OK;
}
