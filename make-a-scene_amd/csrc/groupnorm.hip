// GroupNorm statistics and GroupNorm(+SiLU) backward for NHWC tensors on gfx950.
// Replaces the reduction half of torch.nn.GroupNorm(32, C, eps=1e-6) (reference
// models/modules.py:40-41) -- the normalise/affine/SiLU half is fused into the conv loaders --
// and the autograd of GroupNorm + swish (modules.py:35-37,121-128).
//
// All kernels are HBM-bound streaming passes: 16-byte NHWC loads, a thread owns a fixed
// 16-byte channel unit and walks pixels, so per-channel partial sums live in registers.
// Reductions are two-stage with a fixed summation order (per-thread registers -> LDS slots ->
// per-block partials -> fp64 finalize kernel): bitwise reproducible run to run, no global atomics.
#include "mas_common.h"

namespace {

constexpr int NT = 256;
// Streaming access of the element-wise passes.  Stores of dx / the materialised activation are NON-TEMPORAL: the 0.5 GB tensors do not
// fit the 256 MiB Infinity Cache, and written through the default policy they evict what the next pass could still have hit (x and
// da, left there by gn_bwd_partial in the reverse block order).  Measured (profiles/r03_gn_streaming.txt): gn_bwd 128 ch @256^2
// 0.543 -> 0.528 ms, with 4096 apply blocks 0.479; @128^2 0.125 -> 0.111.  Non-temporal LOADS are a wash: gn_bwd_partial alone gains
// 19 % (207 -> 169 us = 6.35 TB/s) and gn_bwd_apply loses the same microseconds again (its operands were not left in the cache) --
// -DGN_NT_LD keeps the experiment; -DGN_NO_NT_ST restores default-policy stores.
#ifdef GN_NT_LD
#define GN_LD(p) __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p))
#else
#define GN_LD(p) (*reinterpret_cast<const u32x4*>(p))
#endif
#ifndef GN_NO_NT_ST
#define GN_ST(p, v) __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(p))
#else
#define GN_ST(p, v) (*reinterpret_cast<u32x4*>(p) = (v))
#endif
constexpr int MAX_SPLIT = 64;

// ---------------------------------------------------------------------------------------
// stage 1: per (n, split) block: sum and sum of squares per channel over a slab of pixels
// partial layout: [N][split][C][2]
template <typename T>
__global__ __launch_bounds__(NT) void gn_stats_partial(const T* __restrict__ x, int HW, int C, int nsplit,
                                                       float* __restrict__ partial, int rev) {
    constexpr int EPU = 16 / (int)sizeof(T);
    // rev: the first-dispatched blocks take the END of the tensor -- the part its producer (which walked it front to back)
    // touched last and that is most likely still resident in the 256 MiB Infinity Cache; the consumer that follows walks front
    // to back again and meets what THIS pass touched last (see mas_gn_stats)
    const int bid = rev ? (int)gridDim.x - 1 - (int)blockIdx.x : (int)blockIdx.x;
    const int n = bid / nsplit, sp = bid % nsplit;
    const int upp = C / EPU;                        // 16-byte units per pixel
    const int tid = threadIdx.x;
    extern __shared__ float red[];                  // [C][2]
    for (int i = tid; i < 2 * C; i += NT) red[i] = 0.0f;
    __syncthreads();
    const int rows_per = (HW + nsplit - 1) / nsplit;
    const int r0 = sp * rows_per, r1 = min(HW, r0 + rows_per);
    // thread t handles unit (t % upp) when NT % upp == 0; otherwise units are walked linearly
    const size_t base = (size_t)n * HW * C;
    if (NT % upp == 0) {
        const int cu = tid % upp, rstep = NT / upp;
        float s[EPU], q[EPU];
#pragma unroll
        for (int e = 0; e < EPU; ++e) { s[e] = 0.0f; q[e] = 0.0f; }
        auto add = [&](const u32x4& raw) {
            const T* rv = reinterpret_cast<const T*>(&raw);
#pragma unroll
            for (int e = 0; e < EPU; ++e) { const float v = (float)rv[e]; s[e] += v; q[e] += v * v; }
        };
        // four rows in flight per thread (one load per iteration left this HBM-bound pass at 61 % of 8 TB/s)
        int r = r0 + tid / upp;
        for (; r + 3 * rstep < r1; r += 4 * rstep) {
            const T* px = x + base + (size_t)r * C + cu * EPU;
            const u32x4 a0 = *reinterpret_cast<const u32x4*>(px), a1 = *reinterpret_cast<const u32x4*>(px + (size_t)rstep * C);
            const u32x4 a2 = *reinterpret_cast<const u32x4*>(px + (size_t)2 * rstep * C), a3 = *reinterpret_cast<const u32x4*>(px + (size_t)3 * rstep * C);
            add(a0); add(a1); add(a2); add(a3);
        }
        for (; r < r1; r += rstep) add(*reinterpret_cast<const u32x4*>(x + base + (size_t)r * C + cu * EPU));
        // fixed-order (deterministic) reduction over the NT/upp pixel-row groups
        float* slots = red + 2 * C;                 // [NT/upp][C][2]
        const int rg = tid / upp;
#pragma unroll
        for (int e = 0; e < EPU; ++e) {
            slots[((size_t)rg * C + cu * EPU + e) * 2 + 0] = s[e];
            slots[((size_t)rg * C + cu * EPU + e) * 2 + 1] = q[e];
        }
        __syncthreads();
        for (int i = tid; i < 2 * C; i += NT) {
            float a = 0.0f;
            for (int k = 0; k < rstep; ++k) a += slots[(size_t)k * 2 * C + i];
            red[i] = a;
        }
    } else {
        const long long total = (long long)(r1 - r0) * upp;
        for (long long u = tid; u < total; u += NT) {
            const int r = r0 + (int)(u / upp), cu = (int)(u % upp);
            u32x4 raw = *reinterpret_cast<const u32x4*>(x + base + (size_t)r * C + cu * EPU);
            const T* rv = reinterpret_cast<const T*>(&raw);
#pragma unroll
            for (int e = 0; e < EPU; ++e) {
                const float v = (float)rv[e];
                atomicAdd(&red[(cu * EPU + e) * 2 + 0], v);
                atomicAdd(&red[(cu * EPU + e) * 2 + 1], v * v);
            }
        }
    }
    __syncthreads();
    float* out = partial + ((size_t)(n * nsplit + sp) * C) * 2;
    for (int i = tid; i < 2 * C; i += NT) out[i] = red[i];
}

// stage 2: one block per n: combine splits (fp64), emit mean/rstd per group and scale/shift per channel
__global__ __launch_bounds__(NT) void gn_stats_finalize(const float* __restrict__ partial, int HW, int C, int G,
                                                        int nsplit, float eps, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, float* __restrict__ mean_rstd,
                                                        float* __restrict__ ss) {
    const int n = blockIdx.x, tid = threadIdx.x;
    const int cpg = C / G;
    extern __shared__ double dred[];                // [C][2] then [G][2] floats
    double* csum = dred;
    float* gstat = reinterpret_cast<float*>(dred + 2 * C);
    for (int c = tid; c < C; c += NT) {
        double s = 0.0, q = 0.0;
        // eight partials in flight per thread (the loads of a chain `s += p[sp]` went out one L2 round trip at a time: 9 us for
        // 32 splits, ~170 of these launches per step); the additions keep their order: bitwise the same sums
        int sp = 0;
        for (; sp + 8 <= nsplit; sp += 8) {
            float2 v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = *reinterpret_cast<const float2*>(partial + ((size_t)(n * nsplit + sp + k) * C + c) * 2);
#pragma unroll
            for (int k = 0; k < 8; ++k) { s += (double)v[k].x; q += (double)v[k].y; }
        }
        for (; sp < nsplit; ++sp) {
            const float* p = partial + ((size_t)(n * nsplit + sp) * C + c) * 2;
            s += (double)p[0]; q += (double)p[1];
        }
        csum[2 * c] = s; csum[2 * c + 1] = q;
    }
    __syncthreads();
    for (int g = tid; g < G; g += NT) {
        double s = 0.0, q = 0.0;
        for (int j = 0; j < cpg; ++j) { s += csum[2 * (g * cpg + j)]; q += csum[2 * (g * cpg + j) + 1]; }
        const double m = (double)cpg * (double)HW;
        const double mean = s / m;
        double var = q / m - mean * mean;
        if (var < 0.0) var = 0.0;
        const float rstd = (float)(1.0 / sqrt(var + (double)eps));
        gstat[2 * g] = (float)mean; gstat[2 * g + 1] = rstd;
        mean_rstd[((size_t)n * G + g) * 2 + 0] = (float)mean;
        mean_rstd[((size_t)n * G + g) * 2 + 1] = rstd;
    }
    __syncthreads();
    if (ss) {
        for (int c = tid; c < C; c += NT) {
            const float mean = gstat[2 * (c / cpg)], rstd = gstat[2 * (c / cpg) + 1];
            const float ga = gamma ? gamma[c] : 1.0f, be = beta ? beta[c] : 0.0f;
            const float sc = rstd * ga;
            ss[((size_t)n * C + c) * 2 + 0] = sc;
            ss[((size_t)n * C + c) * 2 + 1] = be - mean * sc;
        }
    }
}

// ---------------------------------------------------------------------------------------
// backward stage 1: per (n, split): S1[c] = sum du, S2[c] = sum du*xhat, with
//   u = x*sc+sh, du = da*silu'(u) (or da), xhat = (x-mean)*rstd
template <typename T>
__global__ __launch_bounds__(NT) void gn_bwd_partial(const T* __restrict__ x, const T* __restrict__ da, int HW, int C,
                                                     int G, int nsplit, int act, const float* __restrict__ mean_rstd,
                                                     const float* __restrict__ ss, float* __restrict__ partial, int rev) {
    constexpr int EPU = 16 / (int)sizeof(T);
    const int bid = rev ? (int)gridDim.x - 1 - (int)blockIdx.x : (int)blockIdx.x;   // (see gn_stats_partial)
    const int n = bid / nsplit, sp = bid % nsplit;
    const int upp = C / EPU, cpg = C / G;
    const int tid = threadIdx.x;
    extern __shared__ float red[];
    for (int i = tid; i < 2 * C; i += NT) red[i] = 0.0f;
    __syncthreads();
    const int rows_per = (HW + nsplit - 1) / nsplit;
    const int r0 = sp * rows_per, r1 = min(HW, r0 + rows_per);
    const size_t base = (size_t)n * HW * C;
    const long long total = (long long)(r1 - r0) * upp;
    const bool fixed = (NT % upp) == 0;
    // generic walk; when `fixed`, cu is loop invariant and the per-channel constants stay in registers
    int cu_prev = -1;
    float sc[EPU], sh[EPU], mu[EPU], rs[EPU], s1[EPU], s2[EPU];
#pragma unroll
    for (int e = 0; e < EPU; ++e) { s1[e] = 0.0f; s2[e] = 0.0f; sc[e] = sh[e] = mu[e] = rs[e] = 0.0f; }
    long long u_start = tid;
    if (fixed && total > 0) {            // the channel unit of a thread is loop invariant: constants once, two units in flight per iteration
        const int cu = tid % upp;
#pragma unroll
        for (int e = 0; e < EPU; ++e) {
            const int c = cu * EPU + e;
            sc[e] = ss[((size_t)n * C + c) * 2 + 0]; sh[e] = ss[((size_t)n * C + c) * 2 + 1];
            mu[e] = mean_rstd[((size_t)n * G + c / cpg) * 2 + 0]; rs[e] = mean_rstd[((size_t)n * G + c / cpg) * 2 + 1];
        }
        cu_prev = cu;
        auto acc2 = [&](const u32x4& rx, const u32x4& rd) {
            const T* xv = reinterpret_cast<const T*>(&rx);
            const T* dv = reinterpret_cast<const T*>(&rd);
#pragma unroll
            for (int e = 0; e < EPU; ++e) {
                const float xe = (float)xv[e];
                float du = (float)dv[e];
                if (act == MAS_ACT_AFFINE_SILU) du *= dsilu_f(xe * sc[e] + sh[e]);
                s1[e] += du; s2[e] += du * (xe - mu[e]) * rs[e];
            }
        };
        long long u = tid;
        const size_t rb = base + (size_t)r0 * C;
        for (; u + NT < total; u += 2 * NT) {
            const size_t o0 = rb + (size_t)u * EPU, o1 = rb + (size_t)(u + NT) * EPU;
            const u32x4 x0 = *reinterpret_cast<const u32x4*>(x + o0), x1 = *reinterpret_cast<const u32x4*>(x + o1);
            const u32x4 d0 = *reinterpret_cast<const u32x4*>(da + o0), d1 = *reinterpret_cast<const u32x4*>(da + o1);
            acc2(x0, d0); acc2(x1, d1);
        }
        u_start = u;
    }
    for (long long u = u_start; u < total; u += NT) {
        const int r = r0 + (int)(u / upp), cu = (int)(u % upp);
        if (cu != cu_prev) {
            if (cu_prev >= 0 && !fixed) {
#pragma unroll
                for (int e = 0; e < EPU; ++e) {
                    atomicAdd(&red[(cu_prev * EPU + e) * 2 + 0], s1[e]); atomicAdd(&red[(cu_prev * EPU + e) * 2 + 1], s2[e]);
                    s1[e] = 0.0f; s2[e] = 0.0f;
                }
            }
#pragma unroll
            for (int e = 0; e < EPU; ++e) {
                const int c = cu * EPU + e;
                sc[e] = ss[((size_t)n * C + c) * 2 + 0]; sh[e] = ss[((size_t)n * C + c) * 2 + 1];
                mu[e] = mean_rstd[((size_t)n * G + c / cpg) * 2 + 0]; rs[e] = mean_rstd[((size_t)n * G + c / cpg) * 2 + 1];
            }
            cu_prev = cu;
        }
        const size_t off = base + (size_t)r * C + cu * EPU;
        u32x4 rx = *reinterpret_cast<const u32x4*>(x + off);
        u32x4 rd = *reinterpret_cast<const u32x4*>(da + off);
        const T* xv = reinterpret_cast<const T*>(&rx);
        const T* dv = reinterpret_cast<const T*>(&rd);
#pragma unroll
        for (int e = 0; e < EPU; ++e) {
            const float xe = (float)xv[e];
            float du = (float)dv[e];
            if (act == MAS_ACT_AFFINE_SILU) du *= dsilu_f(xe * sc[e] + sh[e]);
            s1[e] += du; s2[e] += du * (xe - mu[e]) * rs[e];
        }
    }
    if (fixed) {
        float* slots = red + 2 * C;                 // [NT/upp][C][2]
        const int cu = tid % upp, rg = tid / upp, rstep = NT / upp;
#pragma unroll
        for (int e = 0; e < EPU; ++e) {
            slots[((size_t)rg * C + cu * EPU + e) * 2 + 0] = s1[e];
            slots[((size_t)rg * C + cu * EPU + e) * 2 + 1] = s2[e];
        }
        __syncthreads();
        for (int i = tid; i < 2 * C; i += NT) {
            float a = 0.0f;
            for (int k = 0; k < rstep; ++k) a += slots[(size_t)k * 2 * C + i];
            red[i] = a;
        }
    } else if (cu_prev >= 0) {
#pragma unroll
        for (int e = 0; e < EPU; ++e) {
            atomicAdd(&red[(cu_prev * EPU + e) * 2 + 0], s1[e]); atomicAdd(&red[(cu_prev * EPU + e) * 2 + 1], s2[e]);
        }
    }
    __syncthreads();
    float* out = partial + ((size_t)(n * nsplit + sp) * C) * 2;
    for (int i = tid; i < 2 * C; i += NT) out[i] = red[i];
}

// backward stage 1, bf16, channel unit of a thread loop invariant (NT % (C / 8) == 0): the same sums on PACKED fp32 math.  The generic
// kernel above issues 26 instructions per element (scalar fp32 ops, a run-time `act` branch per element, s_nop hazards around each
// transcendental) and is instruction-issue-bound at 4.3-5.3 TB/s: 537 M elements x 26 / (1024 SIMDs x 2.4 GHz / 4) = 0.36 ms for the
// 128-channel 256^2 map, HBM floor 0.17 ms.  Here two elements share each VALU instruction (v_pk_fma / v_pk_mul / v_pk_add; there are
// no MFMAs in this kernel for them to disturb) and `act` is a template parameter: ~10 issue slots per element.
template <bool SILU>
__global__ __launch_bounds__(NT) void gn_bwd_partial_pk(const bf16_t* __restrict__ x, const bf16_t* __restrict__ da, int HW, int C, int G,
                                                        int nsplit, const float* __restrict__ mean_rstd, const float* __restrict__ ss,
                                                        float* __restrict__ partial, int rev) {
    const int bid = rev ? (int)gridDim.x - 1 - (int)blockIdx.x : (int)blockIdx.x;   // (see gn_stats_partial)
    const int n = bid / nsplit, sp = bid % nsplit;
    const int upp = C / 8, cpg = C / G;
    const int tid = threadIdx.x;
    extern __shared__ float red[];
    const int rows_per = (HW + nsplit - 1) / nsplit;
    const int r0 = sp * rows_per, r1 = min(HW, r0 + rows_per);
    const long long total = (long long)(r1 - r0) * upp;
    const int cu = tid % upp;
    // the second sum is kept RAW (sum du * x): sum du * xhat = rstd * (sum du * x) - mean * rstd * (sum du) is formed by the finalize
    // kernel in fp64 -- 16 registers (108 -> 92: five waves per SIMD) and one v_pk_fma per pair less in a loop whose VALU time (~10
    // issue slots per element, a third of them the two transcendentals of silu') is within 25 % of its HBM time: 128 ch @256^2 x 96
    // 1.514 -> 1.459 ms for the three launches (profiles/r05_gn_tuning.txt)
    f32x2 sc[4], sh[4], s1[4], s2[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int c = cu * 8 + 2 * p + h;
            sc[p][h] = ss[((size_t)n * C + c) * 2 + 0]; sh[p][h] = ss[((size_t)n * C + c) * 2 + 1];
            s1[p][h] = 0.0f; s2[p][h] = 0.0f;
        }
    }
    (void)cpg; (void)mean_rstd;
    auto acc2 = [&](const u32x4& rx, const u32x4& rd) {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const f32x2 xe = bf16pair_f32(rx[p]);
            f32x2 du = bf16pair_f32(rd[p]);
            if constexpr (SILU) du = du * dsilu2_f(xe * sc[p] + sh[p]);
            s1[p] += du; s2[p] += du * xe;
        }
    };
    const bf16_t* xb_ = x + ((size_t)n * HW + r0) * C;
    const bf16_t* db_ = da + ((size_t)n * HW + r0) * C;
    long long u = tid;
    for (; u + NT < total; u += 2 * NT) {
        const size_t o0 = (size_t)u * 8, o1 = (size_t)(u + NT) * 8;
        const u32x4 x0 = GN_LD(xb_ + o0), x1 = GN_LD(xb_ + o1);
        const u32x4 d0 = GN_LD(db_ + o0), d1 = GN_LD(db_ + o1);
        acc2(x0, d0); acc2(x1, d1);
    }
    for (; u < total; u += NT) {
        const u32x4 x0 = GN_LD(xb_ + (size_t)u * 8), d0 = GN_LD(db_ + (size_t)u * 8);
        acc2(x0, d0);
    }
    float* slots = red + 2 * C;                     // [NT/upp][C][2]: fixed-order combine of the row groups, as in the generic kernel
    const int rg = tid / upp, rstep = NT / upp;
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            slots[((size_t)rg * C + cu * 8 + 2 * p + h) * 2 + 0] = s1[p][h];
            slots[((size_t)rg * C + cu * 8 + 2 * p + h) * 2 + 1] = s2[p][h];
        }
    __syncthreads();
    float* out = partial + ((size_t)(n * nsplit + sp) * C) * 2;
    for (int i = tid; i < 2 * C; i += NT) {
        float a = 0.0f;
        for (int k = 0; k < rstep; ++k) a += slots[(size_t)k * 2 * C + i];
        out[i] = a;
    }
}

// backward stage 2: one block per n: coefficient table coef[n][c][4] = {c1, k2, k3, 0} with
//   dx = c1*du + k2*x + k3 ; c1 = rstd*gamma_c ; k2 = -rstd^2*B/m ; k3 = rstd*(mean*rstd*B - A)/m
//   A = sum_{c in g} gamma_c S1_c ; B = sum_{c in g} gamma_c S2_c ; m = cpg*HW
// and per-sample contributions to dgamma/dbeta: nsum[n][c][2] = {S2, S1}
__global__ __launch_bounds__(NT) void gn_bwd_finalize(const float* __restrict__ partial, int HW, int C, int G, int nsplit,
                                                      const float* __restrict__ gamma, const float* __restrict__ mean_rstd,
                                                      float* __restrict__ coef, float* __restrict__ nsum, int raw) {
    const int n = blockIdx.x, tid = threadIdx.x;
    const int cpg = C / G;
    extern __shared__ double dred[];
    double* cs = dred;                               // [C][2]
    for (int c = tid; c < C; c += NT) {
        double a = 0.0, b = 0.0;
        int sp = 0;
        for (; sp + 8 <= nsplit; sp += 8) {          // eight loads in flight, additions in the same order (see gn_stats_finalize)
            float2 v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = *reinterpret_cast<const float2*>(partial + ((size_t)(n * nsplit + sp + k) * C + c) * 2);
#pragma unroll
            for (int k = 0; k < 8; ++k) { a += (double)v[k].x; b += (double)v[k].y; }
        }
        for (; sp < nsplit; ++sp) {
            const float* p = partial + ((size_t)(n * nsplit + sp) * C + c) * 2;
            a += (double)p[0]; b += (double)p[1];
        }
        if (raw) {                                   // b arrived as sum du * x (gn_bwd_partial_pk): -> sum du * xhat
            const double mu = mean_rstd[((size_t)n * G + c / cpg) * 2 + 0], rs = mean_rstd[((size_t)n * G + c / cpg) * 2 + 1];
            b = rs * (b - mu * a);
        }
        cs[2 * c] = a; cs[2 * c + 1] = b;
        nsum[((size_t)n * C + c) * 2 + 0] = (float)b;   // -> dgamma
        nsum[((size_t)n * C + c) * 2 + 1] = (float)a;   // -> dbeta
    }
    __syncthreads();
    for (int c = tid; c < C; c += NT) {
        const int g = c / cpg;
        double A = 0.0, B = 0.0;
        for (int j = 0; j < cpg; ++j) {
            const double ga = gamma ? (double)gamma[g * cpg + j] : 1.0;
            A += ga * cs[2 * (g * cpg + j)]; B += ga * cs[2 * (g * cpg + j) + 1];
        }
        const double mean = mean_rstd[((size_t)n * G + g) * 2 + 0], rstd = mean_rstd[((size_t)n * G + g) * 2 + 1];
        const double m = (double)cpg * (double)HW;
        float* o = coef + ((size_t)n * C + c) * 4;
        o[0] = (float)(rstd * (gamma ? (double)gamma[c] : 1.0));
        o[1] = (float)(-rstd * rstd * B / m);
        o[2] = (float)(rstd * (mean * rstd * B - A) / m);
        o[3] = 0.0f;
    }
}

// backward stage 3: elementwise dx = c1*du + k2*x + k3 (+ dres)
template <typename T>
__global__ __launch_bounds__(NT) void gn_bwd_apply(const T* __restrict__ x, const T* __restrict__ da, const T* __restrict__ dres,
                                                   T* __restrict__ dx, int HW, int C, int act, const float* __restrict__ ss,
                                                   const float* __restrict__ coef, long long units_per_n,
                                                   const float* __restrict__ nsum, int N, float* __restrict__ dgamma, float* __restrict__ dbeta) {
    constexpr int EPU = 16 / (int)sizeof(T);
    const int upp = C / EPU;
    const int n = blockIdx.y;
    if (blockIdx.y == 0 && (dgamma || dbeta)) {     // dgamma / dbeta = sum over n of nsum (the former gn_bwd_param_reduce launch: 9 us of
        for (int c = blockIdx.x * NT + threadIdx.x; c < C; c += gridDim.x * NT) {        // dependent launch latency per GroupNorm)
            double a = 0.0, b = 0.0;
            for (int m = 0; m < N; ++m) { a += (double)nsum[((size_t)m * C + c) * 2 + 0]; b += (double)nsum[((size_t)m * C + c) * 2 + 1]; }
            if (dgamma) dgamma[c] = (float)a;
            if (dbeta) dbeta[c] = (float)b;
        }
    }
    const size_t base = (size_t)n * HW * C;
    // NT % upp == 0 (checked by the host): the channel unit of a thread is loop invariant, so the five
    // per-channel coefficients live in registers and the loop is a pure 3-stream pass
    const long long u0 = (long long)blockIdx.x * NT + threadIdx.x;
    const int cu = (int)(u0 % upp);
    float sc[EPU], sh[EPU], k0[EPU], k1[EPU], k2[EPU];
#pragma unroll
    for (int e = 0; e < EPU; ++e) {
        const int c = cu * EPU + e;
        sc[e] = ss[((size_t)n * C + c) * 2]; sh[e] = ss[((size_t)n * C + c) * 2 + 1];
        const float* k = coef + ((size_t)n * C + c) * 4;
        k0[e] = k[0]; k1[e] = k[1]; k2[e] = k[2];
    }
    auto body = [&](const u32x4& rx, const u32x4& rd, const u32x4& rr, size_t off) {
        const T* xv = reinterpret_cast<const T*>(&rx);
        const T* dv = reinterpret_cast<const T*>(&rd);
        const T* rv = reinterpret_cast<const T*>(&rr);
        u32x4 ov;
        T* o = reinterpret_cast<T*>(&ov);
#pragma unroll
        for (int e = 0; e < EPU; ++e) {
            const float xe = (float)xv[e];
            float du = (float)dv[e];
            if (act == MAS_ACT_AFFINE_SILU) du *= dsilu_f(xe * sc[e] + sh[e]);
            float v = k0[e] * du + k1[e] * xe + k2[e];
            if (dres) v += (float)rv[e];
            o[e] = (T)v;
        }
        *reinterpret_cast<u32x4*>(dx + off) = ov;
    };
    // two 16-byte units per iteration: 4-6 loads in flight per thread (the pass is HBM-bound; one unit per iteration left the
    // memory pipeline at 55 % of 8 TB/s)
    const long long stride = (long long)gridDim.x * NT;
    long long u = u0;
    for (; u + stride < units_per_n; u += 2 * stride) {
        const size_t o0 = base + (size_t)u * EPU, o1 = base + (size_t)(u + stride) * EPU;
        const u32x4 rx0 = *reinterpret_cast<const u32x4*>(x + o0), rx1 = *reinterpret_cast<const u32x4*>(x + o1);
        const u32x4 rd0 = *reinterpret_cast<const u32x4*>(da + o0), rd1 = *reinterpret_cast<const u32x4*>(da + o1);
        u32x4 rr0 = {0u, 0u, 0u, 0u}, rr1 = rr0;
        if (dres) { rr0 = *reinterpret_cast<const u32x4*>(dres + o0); rr1 = *reinterpret_cast<const u32x4*>(dres + o1); }
        body(rx0, rd0, rr0, o0);
        body(rx1, rd1, rr1, o1);
    }
    for (; u < units_per_n; u += stride) {
        const size_t off = base + (size_t)u * EPU;
        const u32x4 rx = *reinterpret_cast<const u32x4*>(x + off), rd = *reinterpret_cast<const u32x4*>(da + off);
        u32x4 rr = {0u, 0u, 0u, 0u};
        if (dres) rr = *reinterpret_cast<const u32x4*>(dres + off);
        body(rx, rd, rr, off);
    }
}

// backward stage 3, bf16, on packed fp32 math with `act` / `dres` as template parameters (see gn_bwd_partial_pk: the generic kernel is
// instruction-issue-bound, 0.345 ms where three 537 MB streams take 0.26 ms)
template <bool SILU, bool RES>
__global__ __launch_bounds__(NT) void gn_bwd_apply_pk(const bf16_t* __restrict__ x, const bf16_t* __restrict__ da, const bf16_t* __restrict__ dres,
                                                      bf16_t* __restrict__ dx, int HW, int C, const float* __restrict__ ss,
                                                      const float* __restrict__ coef, long long units_per_n,
                                                      const float* __restrict__ nsum, int N, float* __restrict__ dgamma, float* __restrict__ dbeta) {
    const int upp = C / 8;
    const int n = blockIdx.y;
    if (blockIdx.y == 0 && (dgamma || dbeta)) {     // dgamma / dbeta = sum over n of nsum
        for (int c = blockIdx.x * NT + threadIdx.x; c < C; c += gridDim.x * NT) {
            double a = 0.0, b = 0.0;
            for (int m = 0; m < N; ++m) { a += (double)nsum[((size_t)m * C + c) * 2 + 0]; b += (double)nsum[((size_t)m * C + c) * 2 + 1]; }
            if (dgamma) dgamma[c] = (float)a;
            if (dbeta) dbeta[c] = (float)b;
        }
    }
    const size_t base = (size_t)n * HW * C;
    const long long u0 = (long long)blockIdx.x * NT + threadIdx.x;
    const int cu = (int)(u0 % upp);
    f32x2 sc[4], sh[4], k0[4], k1[4], k2[4];
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int c = cu * 8 + 2 * p + h;
            sc[p][h] = ss[((size_t)n * C + c) * 2]; sh[p][h] = ss[((size_t)n * C + c) * 2 + 1];
            const float* k = coef + ((size_t)n * C + c) * 4;
            k0[p][h] = k[0]; k1[p][h] = k[1]; k2[p][h] = k[2];
        }
    auto body = [&](const u32x4& rx, const u32x4& rd, const u32x4& rr, size_t off) {
        u32x4 ov;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const f32x2 xe = bf16pair_f32(rx[p]);
            f32x2 du = bf16pair_f32(rd[p]);
            if constexpr (SILU) du = du * dsilu2_f(xe * sc[p] + sh[p]);
            f32x2 v = k0[p] * du + (k1[p] * xe + k2[p]);
            if constexpr (RES) v = v + bf16pair_f32(rr[p]);
            ov[p] = f32pair_bf16(v);
        }
        GN_ST(dx + off, ov);
    };
    const long long stride = (long long)gridDim.x * NT;
    long long u = u0;
    for (; u + stride < units_per_n; u += 2 * stride) {
        const size_t o0 = base + (size_t)u * 8, o1 = base + (size_t)(u + stride) * 8;
        const u32x4 rx0 = GN_LD(x + o0), rx1 = GN_LD(x + o1);
        const u32x4 rd0 = GN_LD(da + o0), rd1 = GN_LD(da + o1);
        u32x4 rr0 = {0u, 0u, 0u, 0u}, rr1 = rr0;
        if constexpr (RES) { rr0 = GN_LD(dres + o0); rr1 = GN_LD(dres + o1); }
        body(rx0, rd0, rr0, o0);
        body(rx1, rd1, rr1, o1);
    }
    for (; u < units_per_n; u += stride) {
        const size_t off = base + (size_t)u * 8;
        const u32x4 rx = GN_LD(x + off), rd = GN_LD(da + off);
        u32x4 rr = {0u, 0u, 0u, 0u};
        if constexpr (RES) rr = GN_LD(dres + off);
        body(rx, rd, rr, off);
    }
}

// ---------------------------------------------------------------------------------------
// materialised activation: a = act(x * scale + shift) in the activation dtype (one read, one write).  The optional alternative to
// the fused loader prologue (ops.py MAS_GN_MATERIALIZE): the forward convolution AND the weight gradient of a GroupNorm(+SiLU)-fed
// layer then run prologue-free on `a`.  A thread owns a fixed 16-byte channel unit (scale / shift pairs in registers) and walks
// pixels, four 16-byte loads in flight.
// rev (MAS_GN_ACT_REV, default on): the images are walked LAST TO FIRST.  The producer of x (a convolution) and the consumer of a (the
// next convolution) both walk first to last: this pass then starts on the part of x the producer wrote last (still in the 256 MiB
// Infinity Cache) and ends on the part of a the consumer reads first.  NTS = non-temporal stores of a (MAS_GN_ACT_NT).
template <typename T, bool NTS>
__global__ __launch_bounds__(NT) void gn_act_kernel(const T* __restrict__ x, T* __restrict__ a, int C, int act, const float* __restrict__ ss,
                                                    long long units_per_n, int rev) {
    constexpr int EPU = 16 / (int)sizeof(T);
    const int upp = C / EPU;
    const int n = rev ? (int)gridDim.y - 1 - (int)blockIdx.y : (int)blockIdx.y;
    const size_t base = (size_t)n * (size_t)units_per_n * EPU;
    const long long u0 = (long long)blockIdx.x * NT + threadIdx.x;
    const int cu = (int)(u0 % upp);
    f32x2 sc[EPU / 2], sh[EPU / 2];
#pragma unroll
    for (int e = 0; e < EPU / 2; ++e) {
        const float* q = ss + ((size_t)n * C + cu * EPU + 2 * e) * 2;
        sc[e] = f32x2{q[0], q[2]}; sh[e] = f32x2{q[1], q[3]};
    }
    auto body = [&](u32x4 v, size_t off) {
        if constexpr (sizeof(T) == 2) {
            if (act == MAS_ACT_AFFINE_SILU) {
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = act_pair_bf16<true>(v[q], sc[q], sh[q]);
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = act_pair_bf16<false>(v[q], sc[q], sh[q]);
            }
        } else {
            float* f = reinterpret_cast<float*>(&v);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float u = f[e] * sc[e >> 1][e & 1] + sh[e >> 1][e & 1];
                f[e] = act == MAS_ACT_AFFINE_SILU ? silu_f(u) : u;
            }
        }
        if constexpr (NTS) GN_ST(a + off, v); else *reinterpret_cast<u32x4*>(a + off) = v;
    };
    const long long stride = (long long)gridDim.x * NT;
    long long u = u0;
    for (; u + 3 * stride < units_per_n; u += 4 * stride) {
        const size_t o0 = base + (size_t)u * EPU, o1 = o0 + (size_t)stride * EPU, o2 = o1 + (size_t)stride * EPU, o3 = o2 + (size_t)stride * EPU;
        const u32x4 v0 = *reinterpret_cast<const u32x4*>(x + o0), v1 = *reinterpret_cast<const u32x4*>(x + o1);
        const u32x4 v2 = *reinterpret_cast<const u32x4*>(x + o2), v3 = *reinterpret_cast<const u32x4*>(x + o3);
        body(v0, o0); body(v1, o1); body(v2, o2); body(v3, o3);
    }
    for (; u < units_per_n; u += stride) {
        const size_t off = base + (size_t)u * EPU;
        body(*reinterpret_cast<const u32x4*>(x + off), off);
    }
}

// ---------------------------------------------------------------------------------------
// Small maps (h*w <= 1024 pixels: the 16x16 and 32x32 levels, 47 of the 67 GroupNorms of VQ-IMG): one work-group owns the slab
// (image n, 64-channel block) -- 32 KB at 16x16, 128 KB at 32x32 -- and KEEPS IT IN REGISTERS between the reduction and the
// element-wise phase.  The groups of a block lie inside it (64 % (C / G) == 0), so nothing is exchanged between work-groups:
//   forward : statistics + finalize + activation as ONE launch instead of three (gn_stats_partial 16 us + gn_stats_finalize 6.5 us +
//             gn_act 6 us at 512 ch @16^2 x 32: three dependent launches whose latency is their cost), x read once;
//   backward: (h*w <= 512) partial sums + coefficients + apply as ONE launch instead of three, x / da read once; the sums over the batch
//             for dgamma / dbeta follow as a second, tiny launch (fixed order: bitwise reproducible, no atomics).
// A pixel's 64 channels of the block are 128 contiguous bytes = 8 units of 16 bytes: thread t owns unit t % 8 of pixels t / 8,
// t / 8 + T / 8, ... (a wave reads 8 whole 128-byte lines per load instruction).
constexpr int GS_MAXU = 16;                         // 16-byte units a thread holds per tensor: 1024 pixels x 8 units / 512 threads

struct GnSmallParams {
    const bf16_t* x; const bf16_t* da; const bf16_t* dres; bf16_t* out;      // out: a (forward) or dx (backward)
    const float* gamma; const float* beta;
    float* mean_rstd; float* ss; float* nsum;                                // forward WRITES mean_rstd / ss, backward reads them
    int N, HW, C, G, act;
    float eps;
};

// CB = channels of the block (64: a pixel's block is one 128-byte line = 8 units): thread t owns unit t % (CB / 8) of pixels
// t / (CB / 8), + TP, + 2 TP, ...
template <int UNITS, int CB>
__device__ __forceinline__ void gs_load(u32x4 (&v)[UNITS], const bf16_t* base, int C, int HW, int pl, int cu, int TP) {
#pragma unroll
    for (int k = 0; k < UNITS; ++k) {
        const int px = pl + k * TP;
        v[k] = u32x4{0u, 0u, 0u, 0u};
        if (px < HW) v[k] = *reinterpret_cast<const u32x4*>(base + (size_t)px * C + cu * 8);
    }
}

// per-channel sums over the pixel lanes, fixed order: slots [TP][CB][2] -> tot [CB][2] (fp32), visible after the trailing barrier
template <int CB>
__device__ __forceinline__ void gs_fold(float* slots, float* tot, const f32x2 (&a)[4], const f32x2 (&b)[4], int pl, int cu, int TP, int tid) {
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            slots[((size_t)pl * CB + cu * 8 + 2 * q + h) * 2 + 0] = a[q][h];
            slots[((size_t)pl * CB + cu * 8 + 2 * q + h) * 2 + 1] = b[q][h];
        }
    __syncthreads();
    if (tid < 2 * CB) {
        float acc = 0.0f;
        for (int k = 0; k < TP; ++k) acc += slots[(size_t)k * 2 * CB + tid];
        tot[tid] = acc;
    }
    __syncthreads();
}

template <int UNITS>
__global__ __launch_bounds__(512) void gn_small_fwd_kernel(GnSmallParams p) {
    constexpr int CB = 64;
    extern __shared__ __attribute__((aligned(16))) unsigned char ssm[];
    float* slots = reinterpret_cast<float*>(ssm);                    // [TP][CB][2]
    const int tid = threadIdx.x, TP = blockDim.x / (CB / 8);
    float* tot = slots + (size_t)TP * 2 * CB;                        // [CB][2] sum, sum of squares
    float* cof = tot + 2 * CB;                                       // [CB][2] scale, shift
    const int C = p.C, HW = p.HW, cpg = C / p.G, nb = C / CB;
    const int n = blockIdx.x / nb, c0 = (blockIdx.x % nb) * CB;
    const int cu = tid % (CB / 8), pl = tid / (CB / 8);
    const bf16_t* xb = p.x + (size_t)n * HW * C + c0;
    u32x4 xv[UNITS];
    gs_load<UNITS, CB>(xv, xb, C, HW, pl, cu, TP);
    f32x2 s[4], q2[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) { s[q] = f32x2{0.0f, 0.0f}; q2[q] = f32x2{0.0f, 0.0f}; }
#pragma unroll
    for (int k = 0; k < UNITS; ++k)
#pragma unroll
        for (int q = 0; q < 4; ++q) { const f32x2 v = bf16pair_f32(xv[k][q]); s[q] += v; q2[q] += v * v; }
    gs_fold<CB>(slots, tot, s, q2, pl, cu, TP, tid);
    if (tid < CB) {                                                  // thread = channel c0 + tid: its group's statistics (fp64 combination)
        const int g0 = (tid / cpg) * cpg;
        double sm = 0.0, sq = 0.0;
        for (int j = 0; j < cpg; ++j) { sm += (double)tot[2 * (g0 + j)]; sq += (double)tot[2 * (g0 + j) + 1]; }
        const double m = (double)cpg * (double)HW, mean = sm / m;
        double var = sq / m - mean * mean;
        if (var < 0.0) var = 0.0;
        const float rstd = (float)(1.0 / sqrt(var + (double)p.eps));
        const int c = c0 + tid;
        const float ga = p.gamma ? p.gamma[c] : 1.0f, be = p.beta ? p.beta[c] : 0.0f;
        const float sc = rstd * ga, sh = be - (float)mean * sc;
        cof[2 * tid] = sc; cof[2 * tid + 1] = sh;
        p.ss[((size_t)n * C + c) * 2 + 0] = sc; p.ss[((size_t)n * C + c) * 2 + 1] = sh;
        if (tid % cpg == 0) {
            p.mean_rstd[((size_t)n * p.G + c / cpg) * 2 + 0] = (float)mean;
            p.mean_rstd[((size_t)n * p.G + c / cpg) * 2 + 1] = rstd;
        }
    }
    __syncthreads();
    f32x2 sc[4], sh[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        sc[q] = f32x2{cof[2 * (cu * 8 + 2 * q)], cof[2 * (cu * 8 + 2 * q + 1)]};
        sh[q] = f32x2{cof[2 * (cu * 8 + 2 * q) + 1], cof[2 * (cu * 8 + 2 * q + 1) + 1]};
    }
    bf16_t* ab = p.out + (size_t)n * HW * C + c0;
#pragma unroll
    for (int k = 0; k < UNITS; ++k) {
        const int px = pl + k * TP;
        if (px >= HW) continue;
        u32x4 v = xv[k];
        if (p.act == MAS_ACT_AFFINE_SILU) {
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = act_pair_bf16<true>(v[q], sc[q], sh[q]);
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = act_pair_bf16<false>(v[q], sc[q], sh[q]);
        }
        *reinterpret_cast<u32x4*>(ab + (size_t)px * C + cu * 8) = v;
    }
}

template <int UNITS, int CB, bool SILU, bool RES>
__global__ __launch_bounds__(512) void gn_small_bwd_kernel(GnSmallParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char ssm[];
    float* slots = reinterpret_cast<float*>(ssm);
    const int tid = threadIdx.x, TP = blockDim.x / (CB / 8);
    float* tot = slots + (size_t)TP * 2 * CB;                        // [CB][2] S1 (sum du), S2 (sum du * xhat)
    float* cof = tot + 2 * CB;                                       // [CB][4] c1, k2, k3
    const int C = p.C, HW = p.HW, cpg = C / p.G, nb = C / CB;
    const int n = blockIdx.x / nb, c0 = (blockIdx.x % nb) * CB;
    const int cu = tid % (CB / 8), pl = tid / (CB / 8);
    const size_t ib = (size_t)n * HW * C + c0;
    u32x4 xv[UNITS], dv[UNITS];
    gs_load<UNITS, CB>(xv, p.x + ib, C, HW, pl, cu, TP);
    gs_load<UNITS, CB>(dv, p.da + ib, C, HW, pl, cu, TP);
    f32x2 sc[4], sh[4], xr[4], xb[4], s1[4], s2[4];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int c = c0 + cu * 8 + 2 * q + h;
            sc[q][h] = p.ss[((size_t)n * C + c) * 2 + 0]; sh[q][h] = p.ss[((size_t)n * C + c) * 2 + 1];
            const float mu = p.mean_rstd[((size_t)n * p.G + c / cpg) * 2 + 0], rs = p.mean_rstd[((size_t)n * p.G + c / cpg) * 2 + 1];
            xr[q][h] = rs; xb[q][h] = -mu * rs;
            s1[q][h] = 0.0f; s2[q][h] = 0.0f;
        }
#pragma unroll
    for (int k = 0; k < UNITS; ++k)
#pragma unroll
        for (int q = 0; q < 4; ++q) {                                // (pixels beyond HW were loaded as zeros: they add nothing)
            const f32x2 xe = bf16pair_f32(xv[k][q]);
            f32x2 du = bf16pair_f32(dv[k][q]);
            if constexpr (SILU) du = du * dsilu2_f(xe * sc[q] + sh[q]);
            s1[q] += du; s2[q] += du * (xe * xr[q] + xb[q]);
        }
    gs_fold<CB>(slots, tot, s1, s2, pl, cu, TP, tid);
    if (tid < CB) {
        const int c = c0 + tid, g0 = (tid / cpg) * cpg;
        double A = 0.0, B = 0.0;
        for (int j = 0; j < cpg; ++j) {
            const double ga = p.gamma ? (double)p.gamma[c0 + g0 + j] : 1.0;
            A += ga * (double)tot[2 * (g0 + j)]; B += ga * (double)tot[2 * (g0 + j) + 1];
        }
        const double mean = p.mean_rstd[((size_t)n * p.G + c / cpg) * 2 + 0], rstd = p.mean_rstd[((size_t)n * p.G + c / cpg) * 2 + 1];
        const double m = (double)cpg * (double)HW;
        cof[4 * tid + 0] = (float)(rstd * (p.gamma ? (double)p.gamma[c] : 1.0));
        cof[4 * tid + 1] = (float)(-rstd * rstd * B / m);
        cof[4 * tid + 2] = (float)(rstd * (mean * rstd * B - A) / m);
        p.nsum[((size_t)n * C + c) * 2 + 0] = tot[2 * tid + 1];      // -> dgamma
        p.nsum[((size_t)n * C + c) * 2 + 1] = tot[2 * tid];          // -> dbeta
    }
    __syncthreads();
    f32x2 k0[4], k1[4], k2[4];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int cl = cu * 8 + 2 * q + h;
            k0[q][h] = cof[4 * cl]; k1[q][h] = cof[4 * cl + 1]; k2[q][h] = cof[4 * cl + 2];
        }
#pragma unroll
    for (int k = 0; k < UNITS; ++k) {
        const int px = pl + k * TP;
        if (px >= HW) continue;
        const size_t off = ib + (size_t)px * C + cu * 8;
        u32x4 rr = {0u, 0u, 0u, 0u};
        if constexpr (RES) rr = *reinterpret_cast<const u32x4*>(p.dres + off);
        u32x4 ov;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x2 xe = bf16pair_f32(xv[k][q]);
            f32x2 du = bf16pair_f32(dv[k][q]);
            if constexpr (SILU) du = du * dsilu2_f(xe * sc[q] + sh[q]);
            f32x2 v = k0[q] * du + (k1[q] * xe + k2[q]);
            if constexpr (RES) v = v + bf16pair_f32(rr[q]);
            ov[q] = f32pair_bf16(v);
        }
        *reinterpret_cast<u32x4*>(p.out + off) = ov;
    }
}

// dgamma / dbeta = sums over the batch of nsum [N][C][2] (fp64, image order)
__global__ __launch_bounds__(NT) void gn_param_reduce_kernel(const float* __restrict__ nsum, int N, int C, float* __restrict__ dgamma,
                                                             float* __restrict__ dbeta) {
    const int c = blockIdx.x * NT + threadIdx.x;
    if (c >= C) return;
    double a = 0.0, b = 0.0;
    for (int m = 0; m < N; ++m) { a += (double)nsum[((size_t)m * C + c) * 2 + 0]; b += (double)nsum[((size_t)m * C + c) * 2 + 1]; }
    if (dgamma) dgamma[c] = (float)a;
    if (dbeta) dbeta[c] = (float)b;
}

// the small-map kernels take bf16 tensors of at most 1024 pixels whose channel count is a multiple of 64 with whole groups per block
inline bool gn_small_ok(int dtype, int HW, int C, int G) {
    static const int on = mas_env_int("MAS_GN_SMALL", 1);
    return on && dtype == MAS_BF16 && HW >= 1 && HW <= 1024 && C % 64 == 0 && C % G == 0 && 64 % (C / G) == 0;
}
// forward: 64-channel blocks, <= 16 units per thread (h*w <= 1024).  backward: x AND da in registers, <= 8 units of each per thread:
// h*w <= 512.  (The backward at 32x32 on 32-channel blocks measured 0.050 ms against the three launches' 0.047 at 512 channels, 0.029
// against 0.032 at 256: not worth a second geometry -- 32x32 keeps the three launches; profiles/r04_gn_small.txt.)
inline int gn_small_threads(int HW) { return HW <= 256 ? 256 : 512; }
inline bool gn_small_bwd_ok(int dtype, int HW, int C, int G) { return HW <= 512 && gn_small_ok(dtype, HW, C, G); }
inline int gn_small_units(int HW, int cb) { const int t = gn_small_threads(HW); return (HW * (cb / 8) + t - 1) / t; }
inline size_t gn_small_lds(int HW, int cb) { return ((size_t)gn_small_threads(HW) / (cb / 8) * 2 * cb + 2 * cb + 4 * cb) * sizeof(float); }

// Pass ordering against the Infinity Cache (256 MiB, memory side): the tensors of the 256x256 / 128x128 levels are 134-537 MB, so a
// pass that re-walks a tensor in the SAME direction as the pass before it finds everything it needs already evicted, while the
// opposite direction starts on the most recently touched ~quarter.  Convolutions and the backward apply pass walk images front
// to back; the two reduction passes (statistics, backward partial sums) walk back to front (+0.1 ms per step, DESIGN history R3).
int gn_reverse() { return 1; }

int pick_split(int N, int HW, int max_split = MAX_SPLIT) {
    // enough blocks to fill 256 CUs a few times over, but >= 64 pixels per block
    constexpr int target = 1024;                    // (512 / 2048 measured within 1 %: profiles/r03_gn_streaming.txt)
    int s = mas_cdiv(target, N);
    if (s > max_split) s = max_split;
    const int cap = HW / 64 > 0 ? HW / 64 : 1;
    if (s > cap) s = cap;
    return s < 1 ? 1 : s;
}

}  // namespace

extern "C" size_t mas_gn_stats_workspace(int N, int C) { return (size_t)N * MAX_SPLIT * C * 2 * sizeof(float); }

extern "C" int mas_gn_stats(const void* x, int dtype, int N, int HW, int C, int G, float eps, const float* gamma,
                            const float* beta, float* mean_rstd, float* scale_shift, void* workspace, size_t ws_bytes,
                            void* stream) {
    MAS_ENTER();
    if (!x || !mean_rstd || !workspace) MAS_FAIL(MAS_EINVAL, "gn_stats: null argument");
    if (N <= 0 || HW <= 0 || C <= 0 || G <= 0 || C % G) MAS_FAIL(MAS_EINVAL, "gn_stats: bad shape N=%d HW=%d C=%d G=%d", N, HW, C, G);
    const int epu = dtype == MAS_BF16 ? 8 : 4;
    if (C % epu) MAS_FAIL(MAS_EUNSUPPORTED, "gn_stats: C=%d must be a multiple of %d", C, epu);
    if (ws_bytes < mas_gn_stats_workspace(N, C)) MAS_FAIL(MAS_EWORKSPACE, "gn_stats: workspace too small");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int nsplit = pick_split(N, HW);
    float* partial = reinterpret_cast<float*>(workspace);
    const size_t lds1 = ((size_t)2 * C + (size_t)NT * epu * 2) * sizeof(float);
    if (dtype == MAS_BF16)
        hipLaunchKernelGGL(gn_stats_partial<bf16_t>, dim3(N * nsplit), dim3(NT), lds1, s, (const bf16_t*)x, HW, C, nsplit, partial, gn_reverse());
    else
        hipLaunchKernelGGL(gn_stats_partial<float>, dim3(N * nsplit), dim3(NT), lds1, s, (const float*)x, HW, C, nsplit, partial, gn_reverse());
    MAS_CHECK_LAUNCH("gn_stats_partial");
    const size_t lds2 = (size_t)2 * C * sizeof(double) + (size_t)2 * G * sizeof(float);
    hipLaunchKernelGGL(gn_stats_finalize, dim3(N), dim3(NT), lds2, s, partial, HW, C, G, nsplit, eps, gamma, beta, mean_rstd, scale_shift);
    MAS_CHECK_LAUNCH("gn_stats_finalize");
    return MAS_OK;
}

// GroupNorm statistics from a table of per-tile partial sums [N][rows][C][2] (sum, sum of squares) that the producing convolution
// filled in its epilogue (mas_conv_fwd_stats): the full-tensor read of mas_gn_stats disappears, only this finalize remains.
extern "C" int mas_gn_stats_from_partials(const float* partial, int N, int HW, int C, int G, int rows, float eps, const float* gamma,
                                          const float* beta, float* mean_rstd, float* scale_shift, void* stream) {
    MAS_ENTER();
    if (!partial || !mean_rstd) MAS_FAIL(MAS_EINVAL, "gn_stats_from_partials: null argument");
    if (N <= 0 || HW <= 0 || C <= 0 || G <= 0 || C % G || rows <= 0) MAS_FAIL(MAS_EINVAL, "gn_stats_from_partials: bad shape");
    const size_t lds2 = (size_t)2 * C * sizeof(double) + (size_t)2 * G * sizeof(float);
    hipLaunchKernelGGL(gn_stats_finalize, dim3(N), dim3(NT), lds2, reinterpret_cast<hipStream_t>(stream), partial, HW, C, G, rows, eps, gamma, beta,
                       mean_rstd, scale_shift);
    MAS_CHECK_LAUNCH("gn_stats_from_partials");
    return MAS_OK;
}

// a [N,HW,C] = act(x * scale + shift) with the per-(sample, channel) scale / shift pairs of mas_gn_stats (act: MAS_ACT_AFFINE or
// MAS_ACT_AFFINE_SILU), rounded to the activation dtype exactly as the fused conv / wgrad loaders round it
extern "C" int mas_gn_act(const void* x, void* a, int dtype, int N, int HW, int C, int act, const float* scale_shift, void* stream) {
    MAS_ENTER();
    if (!x || !a || !scale_shift) MAS_FAIL(MAS_EINVAL, "gn_act: null argument");
    if (act != MAS_ACT_AFFINE && act != MAS_ACT_AFFINE_SILU) MAS_FAIL(MAS_EINVAL, "gn_act: bad act %d", act);
    if (N <= 0 || HW <= 0 || C <= 0) MAS_FAIL(MAS_EINVAL, "gn_act: bad shape");
    const int epu = dtype == MAS_BF16 ? 8 : 4;
    if (C % epu || NT % (C / epu)) MAS_FAIL(MAS_EUNSUPPORTED, "gn_act: C=%d: C/%d must divide %d", C, epu, NT);
    const long long units_per_n = (long long)HW * C / epu;
    int gx = (int)((units_per_n + NT - 1) / NT);
    static const int act_blocks = mas_env_int("MAS_GN_ACT_BLOCKS", 8192);
    const int cap = mas_cdiv(act_blocks, N) > 0 ? mas_cdiv(act_blocks, N) : 1;
    if (gx > cap) gx = cap;
    if (gx < 1) gx = 1;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    static const int rev = mas_env_int("MAS_GN_ACT_REV", 1), nts = mas_env_int("MAS_GN_ACT_NT", 1);
    if (dtype == MAS_BF16 && nts)
        hipLaunchKernelGGL((gn_act_kernel<bf16_t, true>), dim3(gx, N), dim3(NT), 0, s, (const bf16_t*)x, (bf16_t*)a, C, act, scale_shift, units_per_n, rev);
    else if (dtype == MAS_BF16)
        hipLaunchKernelGGL((gn_act_kernel<bf16_t, false>), dim3(gx, N), dim3(NT), 0, s, (const bf16_t*)x, (bf16_t*)a, C, act, scale_shift, units_per_n, rev);
    else
        hipLaunchKernelGGL((gn_act_kernel<float, true>), dim3(gx, N), dim3(NT), 0, s, (const float*)x, (float*)a, C, act, scale_shift, units_per_n, rev);
    MAS_CHECK_LAUNCH("gn_act");
    return MAS_OK;
}

// ---- small maps: statistics + finalize + activation in one launch; the backward in one launch + the batch sums ----
namespace {
template <int UNITS>
void gn_small_fwd_launch(const GnSmallParams& p, int T, size_t lds, hipStream_t s) {
    hipLaunchKernelGGL(gn_small_fwd_kernel<UNITS>, dim3((unsigned)(p.N * (p.C / 64))), dim3(T), lds, s, p);
}
template <int UNITS, int CB>
void gn_small_bwd_launch(const GnSmallParams& p, int T, size_t lds, hipStream_t s) {
    const dim3 grid((unsigned)(p.N * (p.C / CB)));
    const bool silu = p.act == MAS_ACT_AFFINE_SILU;
    if (silu && p.dres) hipLaunchKernelGGL((gn_small_bwd_kernel<UNITS, CB, true, true>), grid, dim3(T), lds, s, p);
    else if (silu) hipLaunchKernelGGL((gn_small_bwd_kernel<UNITS, CB, true, false>), grid, dim3(T), lds, s, p);
    else if (p.dres) hipLaunchKernelGGL((gn_small_bwd_kernel<UNITS, CB, false, true>), grid, dim3(T), lds, s, p);
    else hipLaunchKernelGGL((gn_small_bwd_kernel<UNITS, CB, false, false>), grid, dim3(T), lds, s, p);
}
}  // namespace

extern "C" int mas_gn_small_supported(int dtype, int HW, int C, int G) { return (G > 0 && C > 0 && gn_small_ok(dtype, HW, C, G)) ? 1 : 0; }

// GroupNorm statistics AND the materialised activation in one launch (small maps: mas_gn_small_supported): the same mean_rstd /
// scale_shift as mas_gn_stats (up to the summation order) and a = act(x * scale + shift) exactly as mas_gn_act forms it from them.
extern "C" int mas_gn_stats_act(const void* x, void* a, int dtype, int N, int HW, int C, int G, float eps, const float* gamma,
                                const float* beta, int act, float* mean_rstd, float* scale_shift, void* stream) {
    MAS_ENTER();
    if (!x || !a || !mean_rstd || !scale_shift) MAS_FAIL(MAS_EINVAL, "gn_stats_act: null argument");
    if (act != MAS_ACT_AFFINE && act != MAS_ACT_AFFINE_SILU) MAS_FAIL(MAS_EINVAL, "gn_stats_act: bad act %d", act);
    if (N <= 0 || HW <= 0 || C <= 0 || G <= 0 || C % G) MAS_FAIL(MAS_EINVAL, "gn_stats_act: bad shape");
    if (!gn_small_ok(dtype, HW, C, G)) MAS_FAIL(MAS_EUNSUPPORTED, "gn_stats_act: needs bf16, h*w <= 1024, C %% 64 == 0, 64 %% (C / G) == 0");
    GnSmallParams p{};
    p.x = (const bf16_t*)x; p.out = (bf16_t*)a; p.gamma = gamma; p.beta = beta; p.mean_rstd = mean_rstd; p.ss = scale_shift;
    p.N = N; p.HW = HW; p.C = C; p.G = G; p.act = act; p.eps = eps;
    const int T = gn_small_threads(HW), u = gn_small_units(HW, 64);
    const size_t lds = gn_small_lds(HW, 64);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (u <= 1) gn_small_fwd_launch<1>(p, T, lds, s);
    else if (u <= 2) gn_small_fwd_launch<2>(p, T, lds, s);
    else if (u <= 4) gn_small_fwd_launch<4>(p, T, lds, s);
    else if (u <= 8) gn_small_fwd_launch<8>(p, T, lds, s);
    else gn_small_fwd_launch<GS_MAXU>(p, T, lds, s);
    MAS_CHECK_LAUNCH("gn_small_fwd");
    return MAS_OK;
}

// workspace: partial [N][MAX_SPLIT][C][2] + coef [N][C][4] + nsum [N][C][2]
extern "C" size_t mas_gn_bwd_workspace(int N, int C) {
    return ((size_t)N * MAX_SPLIT * C * 2 + (size_t)N * C * 4 + (size_t)N * C * 2) * sizeof(float);
}

static int gn_bwd_check(const void* x, const void* da, const void* dx, int dtype, int N, int HW, int C, int G, int act, const float* mean_rstd,
                        const float* scale_shift, const void* workspace, size_t ws_bytes) {
    if (!x || !da || !dx || !mean_rstd || !scale_shift || !workspace) MAS_FAIL(MAS_EINVAL, "gn_bwd: null argument");
    if (act != MAS_ACT_AFFINE && act != MAS_ACT_AFFINE_SILU) MAS_FAIL(MAS_EINVAL, "gn_bwd: bad act %d", act);
    if (N <= 0 || HW <= 0 || C <= 0 || G <= 0 || C % G) MAS_FAIL(MAS_EINVAL, "gn_bwd: bad shape");
    const int epu = dtype == MAS_BF16 ? 8 : 4;
    if (C % epu || NT % (C / epu)) MAS_FAIL(MAS_EUNSUPPORTED, "gn_bwd: C=%d: C/%d must divide %d", C, epu, NT);
    if (ws_bytes < mas_gn_bwd_workspace(N, C)) MAS_FAIL(MAS_EWORKSPACE, "gn_bwd: workspace too small");
    return MAS_OK;
}

// The entry point the autograd nodes call: the small-map kernel up to 512 pixels (bf16), else the three launches.  (Two persistent
// one-launch kernels that read x / da from HBM once were built in round 4, both correct, both slower -- 0.61 / 0.86 vs 0.48 ms at
// 128 ch @256^2 x 32: profiles/r04_gn_coop_v1.txt, r04_gn_queue_v2.txt; the second is kept as docs/history/experiments/r4_gn_queue.patch.)
extern "C" int mas_gn_bwd(const void* x, const void* da, const void* dres, int dtype, int N, int HW, int C, int G, int act,
                          const float* gamma, const float* mean_rstd, const float* scale_shift, void* dx, float* dgamma,
                          float* dbeta, void* workspace, size_t ws_bytes, void* stream) {
    if (x && da && dx && mean_rstd && scale_shift && workspace && N > 0 && G > 0 && C > 0 && C % G == 0 && (act == MAS_ACT_AFFINE || act == MAS_ACT_AFFINE_SILU) &&
        gn_small_bwd_ok(dtype, HW, C, G) && ws_bytes >= mas_gn_bwd_workspace(N, C)) {
        // small maps: the (image, 64-channel block) slab lives in registers between the sums and the element-wise phase (gn_small_bwd_kernel)
        MAS_ENTER();
        GnSmallParams p{};
        p.x = (const bf16_t*)x; p.da = (const bf16_t*)da; p.dres = (const bf16_t*)dres; p.out = (bf16_t*)dx; p.gamma = gamma;
        p.mean_rstd = const_cast<float*>(mean_rstd); p.ss = const_cast<float*>(scale_shift); p.nsum = reinterpret_cast<float*>(workspace);
        p.N = N; p.HW = HW; p.C = C; p.G = G; p.act = act;
        const int T = gn_small_threads(HW), u = gn_small_units(HW, 64);
        const size_t lds = gn_small_lds(HW, 64);
        hipStream_t s = reinterpret_cast<hipStream_t>(stream);
        if (u <= 1) gn_small_bwd_launch<1, 64>(p, T, lds, s);
        else if (u <= 2) gn_small_bwd_launch<2, 64>(p, T, lds, s);
        else if (u <= 4) gn_small_bwd_launch<4, 64>(p, T, lds, s);
        else gn_small_bwd_launch<8, 64>(p, T, lds, s);
        MAS_CHECK_LAUNCH("gn_small_bwd");
        if (dgamma || dbeta) {
            hipLaunchKernelGGL(gn_param_reduce_kernel, dim3((unsigned)((C + NT - 1) / NT)), dim3(NT), 0, s, p.nsum, N, C, dgamma, dbeta);
            MAS_CHECK_LAUNCH("gn_param_reduce");
        }
        return MAS_OK;
    }
    return mas_gn_bwd_3pass(x, da, dres, dtype, N, HW, C, G, act, gamma, mean_rstd, scale_shift, dx, dgamma, dbeta, workspace, ws_bytes, stream);
}

// three launches: reduce -> finalize -> apply (fp32, bf16 shapes the one-launch kernel does not take; x and da are read twice)
extern "C" int mas_gn_bwd_3pass(const void* x, const void* da, const void* dres, int dtype, int N, int HW, int C, int G, int act,
                                const float* gamma, const float* mean_rstd, const float* scale_shift, void* dx, float* dgamma,
                                float* dbeta, void* workspace, size_t ws_bytes, void* stream) {
    MAS_ENTER();
    const int rc0 = gn_bwd_check(x, da, dx, dtype, N, HW, C, G, act, mean_rstd, scale_shift, workspace, ws_bytes);
    if (rc0 != MAS_OK) return rc0;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const bool silu = act == MAS_ACT_AFFINE_SILU;
    const int epu = dtype == MAS_BF16 ? 8 : 4;
    float* partial = reinterpret_cast<float*>(workspace);
    float* coef = partial + (size_t)N * MAX_SPLIT * C * 2;
    float* nsum = coef + (size_t)N * C * 4;
    const size_t lds1 = ((size_t)2 * C + (size_t)NT * epu * 2) * sizeof(float);
    const bool pk = dtype == MAS_BF16;                               // (NT % (C / 8) == 0 was checked above)
    const long long units_per_n = (long long)HW * C / epu;
    static const int apply_blocks = mas_env_int("MAS_GN_APPLY_BLOCKS", 8192);
    const int nsplit = pick_split(N, HW);
    if (pk && silu)
        hipLaunchKernelGGL(gn_bwd_partial_pk<true>, dim3(N * nsplit), dim3(NT), lds1, s, (const bf16_t*)x, (const bf16_t*)da, HW, C, G, nsplit, mean_rstd, scale_shift, partial, gn_reverse());
    else if (pk)
        hipLaunchKernelGGL(gn_bwd_partial_pk<false>, dim3(N * nsplit), dim3(NT), lds1, s, (const bf16_t*)x, (const bf16_t*)da, HW, C, G, nsplit, mean_rstd, scale_shift, partial, gn_reverse());
    else
        hipLaunchKernelGGL(gn_bwd_partial<float>, dim3(N * nsplit), dim3(NT), lds1, s, (const float*)x, (const float*)da, HW, C, G, nsplit, act, mean_rstd, scale_shift, partial, gn_reverse());
    MAS_CHECK_LAUNCH("gn_bwd_partial");
    hipLaunchKernelGGL(gn_bwd_finalize, dim3(N), dim3(NT), (size_t)2 * C * sizeof(double), s, partial, HW, C, G, nsplit, gamma, mean_rstd, coef, nsum,
                       pk ? 1 : 0);
    MAS_CHECK_LAUNCH("gn_bwd_finalize");
    int gx = (int)((units_per_n + NT - 1) / NT);
    // MAS_GN_APPLY_BLOCKS (8192) blocks over the batch: the grid is (gx, N) and x-fastest, so the budget sets how many IMAGES are walked at
    // once -- 2048 resident blocks / gx.  Round 3 went 2048 -> 4096 (0.528 -> 0.479 ms at 128 ch @256^2: the write stream wants requests
    // in flight); round 5 4096 -> 8192 (eight images in flight instead of sixteen: fewer DRAM fronts; gn_act 0.197 -> 0.184, the step
    // -0.35 ms, profiles/r05_gn_tuning.txt; 16384+ is flat or worse with a residual stream).  At least four 16-byte units per thread
    // (512 ch @32^2 loses 8 % on thinner blocks)
    int cap = mas_cdiv(apply_blocks, N) > 0 ? mas_cdiv(apply_blocks, N) : 1;
    const long long thick = units_per_n / (4LL * NT);
    if (cap > thick) cap = thick > 0 ? (int)thick : 1;
    if (gx > cap) gx = cap;
    if (gx < 1) gx = 1;
#define MAS_GN_APPLY_PK(SILU, RES) hipLaunchKernelGGL((gn_bwd_apply_pk<SILU, RES>), dim3(gx, N), dim3(NT), 0, s, (const bf16_t*)x, (const bf16_t*)da, \
        (const bf16_t*)dres, (bf16_t*)dx, HW, C, scale_shift, coef, units_per_n, nsum, N, dgamma, dbeta)
    if (pk) {
        if (silu && dres) MAS_GN_APPLY_PK(true, true);
        else if (silu) MAS_GN_APPLY_PK(true, false);
        else if (dres) MAS_GN_APPLY_PK(false, true);
        else MAS_GN_APPLY_PK(false, false);
    } else
        hipLaunchKernelGGL(gn_bwd_apply<float>, dim3(gx, N), dim3(NT), 0, s, (const float*)x, (const float*)da, (const float*)dres, (float*)dx, HW, C, act, scale_shift, coef, units_per_n, nsum, N, dgamma, dbeta);
    MAS_CHECK_LAUNCH("gn_bwd_apply");
    return MAS_OK;
}
