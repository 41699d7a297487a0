// Vector-quantiser nearest-codebook lookup for gfx950.
// Replaces Codebook.forward's distance / argmin / gather / loss (reference
// models/modules.py:501-509) and its autograd (straight-through modules.py:512 + the two MSE terms
// of modules.py:509).  The reference materialises d[M,K] fp32 (268 MB at B=32); here d is never
// written: 32x32 distance tiles live in MFMA accumulators and are reduced to a running
// (min, argmin) per latent row in registers.
//
// Numerics: fp32 throughout.  dot(z,e) is an exact-fp32 FMA chain (v_mfma_f32_32x32x2_f32), and
// d = fl(fl(|z|^2 + |e|^2) - 2*dot) follows the reference's evaluation order, so indices are
// bit-exact wherever the reference's own top-2 gap exceeds fp32 summation-order noise.
// Ties resolve to the lowest index like torch.argmin.
#include "mas_common.h"
#include <math.h>

namespace {

constexpr int NT = 256, ROWS_PER_BLOCK = 128, CODES_PER_TILE = 32, MAX_KSPLIT = 64;

__global__ __launch_bounds__(NT) void row_sqnorm(const float* __restrict__ x, int rows, int D, float* __restrict__ out) {
    const int wave = (blockIdx.x * NT + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (wave >= rows) return;
    float s = 0.0f;
    for (int d = lane; d < D; d += 64) { const float v = x[(size_t)wave * D + d]; s += v * v; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) out[wave] = s;
}

struct MinIdx { float v; int i; };
__device__ __forceinline__ void take_min(float& bv, int& bi, float v, int i) {
    if (v < bv || (v == bv && i < bi)) { bv = v; bi = i; }
}

template <int D>
__global__ __launch_bounds__(NT) void vq_partial(const float* __restrict__ z, const float* __restrict__ cb,
                                                 const float* __restrict__ zz, const float* __restrict__ ee, int M, int K,
                                                 int ksplit, float* __restrict__ pval, int* __restrict__ pidx) {
    constexpr int HD = D / 2;                  // dims handled by one lane group
    constexpr int LSTR = D + 4;                // LDS row stride (floats): +16 B breaks b128 bank conflicts
    __shared__ __attribute__((aligned(16))) float tile[CODES_PER_TILE * LSTR];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 5, l31 = lane & 31;
    const int rb = blockIdx.x / ksplit, sp = blockIdx.x % ksplit;
    const int row = rb * ROWS_PER_BLOCK + wave * 32 + l31;
    const bool row_ok = row < M;

    // B operand: this lane's half of its latent row, resident in registers for the whole kernel
    float zf[HD];
    {
        const float* zp = z + (size_t)(row_ok ? row : 0) * D + g * HD;
#pragma unroll
        for (int q = 0; q < HD / 4; ++q) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(zp + 4 * q);
            zf[4 * q] = row_ok ? v[0] : 0.0f; zf[4 * q + 1] = row_ok ? v[1] : 0.0f;
            zf[4 * q + 2] = row_ok ? v[2] : 0.0f; zf[4 * q + 3] = row_ok ? v[3] : 0.0f;
        }
    }
    const float zzr = row_ok ? zz[row] : 0.0f;

    const int ntiles = (K + CODES_PER_TILE - 1) / CODES_PER_TILE;
    const int tiles_per = (ntiles + ksplit - 1) / ksplit;
    const int t0 = sp * tiles_per, t1 = min(ntiles, t0 + tiles_per);

    float best = INFINITY; int besti = 0x7fffffff;
    for (int t = t0; t < t1; ++t) {
        const int k0 = t * CODES_PER_TILE;
        __syncthreads();
        // stage 32 code vectors (coalesced float4 loads)
        for (int u = tid; u < CODES_PER_TILE * (D / 4); u += NT) {
            const int c = u / (D / 4), q = u % (D / 4);
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (k0 + c < K) v = *reinterpret_cast<const f32x4*>(cb + (size_t)(k0 + c) * D + 4 * q);
            *reinterpret_cast<f32x4*>(&tile[c * LSTR + 4 * q]) = v;
        }
        __syncthreads();
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
        const float* ap = &tile[l31 * LSTR + g * HD];
#pragma unroll
        for (int q = 0; q < HD / 4; ++q) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(ap + 4 * q);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], zf[4 * q + j], acc, 0, 0, 0);
        }
        // acc[r] = dot(code k0+acc_row(lane,r), latent row l31)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int k = k0 + acc_row(lane, r);
            if (k < K) {
                const float d = (zzr + ee[k]) - 2.0f * acc[r];
                if (d < best) { best = d; besti = k; }      // ascending k within a lane: strict < keeps the first
            }
        }
    }
    // merge the two lane groups that share a latent row
    const float ov = __shfl_xor(best, 32); const int oi = __shfl_xor(besti, 32);
    take_min(best, besti, ov, oi);
    if (g == 0 && row_ok) { pval[(size_t)row * ksplit + sp] = best; pidx[(size_t)row * ksplit + sp] = besti; }
}

// one wave per latent row: merge split partials, write idx, gather z_q, accumulate (z_q - z)^2
__global__ __launch_bounds__(NT) void vq_finalize(const float* __restrict__ z, const float* __restrict__ cb,
                                                  const float* __restrict__ pval, const int* __restrict__ pidx, int M, int D,
                                                  int ksplit, int64_t* __restrict__ idx, float* __restrict__ zq,
                                                  float* __restrict__ blocksum) {
    __shared__ float wsum[NT / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row = blockIdx.x * (NT / 64) + wave;
    float s = 0.0f;
    if (row < M) {
        float bv = INFINITY; int bi = 0x7fffffff;
        for (int sp = lane; sp < ksplit; sp += 64) take_min(bv, bi, pval[(size_t)row * ksplit + sp], pidx[(size_t)row * ksplit + sp]);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { const float ov = __shfl_xor(bv, o); const int oi = __shfl_xor(bi, o); take_min(bv, bi, ov, oi); }
        if (bi == 0x7fffffff) bi = 0;            // all-NaN row: stay in range
        if (lane == 0) idx[row] = (int64_t)bi;
        for (int d = lane; d < D; d += 64) {
            const float e = cb[(size_t)bi * D + d];
            const float df = e - z[(size_t)row * D + d];
            zq[(size_t)row * D + d] = e;
            s += df * df;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    }
    if (lane == 0) wsum[wave] = s;
    __syncthreads();
    if (threadIdx.x == 0) { float t = 0.0f; for (int w = 0; w < NT / 64; ++w) t += wsum[w]; blocksum[blockIdx.x] = t; }
}

__global__ __launch_bounds__(NT) void vq_loss_reduce(const float* __restrict__ blocksum, int n, float* __restrict__ out) {
    __shared__ double red[NT];
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += NT) s += (double)blocksum[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = NT / 2; o > 0; o >>= 1) { if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x == 0) out[0] = (float)red[0];
}

__global__ __launch_bounds__(NT) void vq_bwd_kernel(const float* __restrict__ z, const float* __restrict__ cb,
                                                    const int64_t* __restrict__ idx, const float* __restrict__ g_zq,
                                                    const float* __restrict__ g_loss, float beta, int M, int D,
                                                    float* __restrict__ dz, float* __restrict__ dcb) {
    const long long total = (long long)M * D;
    const float c = (g_loss ? g_loss[0] : 0.0f) * 2.0f / (float)total;
    for (long long i = (long long)blockIdx.x * NT + threadIdx.x; i < total; i += (long long)gridDim.x * NT) {
        const int m = (int)(i / D), d = (int)(i % D);
        const int64_t k = idx[m];
        const float diff = z[i] - cb[(size_t)k * D + d];
        if (dz) dz[i] = (g_zq ? g_zq[i] : 0.0f) + c * diff;
        if (dcb) atomicAdd(dcb + (size_t)k * D + d, -c * beta * diff);
    }
}

// The codebook gradient without atomics (round 5: the scatter-add above was the last order-dependent sum of the VQ-IMG step besides
// the fp32 1x1 convolutions).  A work-group owns VB_CODES codebook rows; its four waves scan one quarter of the M indices each (64 per
// ballot), and for every position that picked one of its codes a wave adds the 4 D-float row -c beta (z - e) into its OWN LDS
// accumulator, in position order; the four accumulators are folded in wave order and EVERY row of the range is stored (zeros
// included: the output needs no fill).  Fixed order: bitwise run-to-run deterministic.  D <= 256, D % 4 == 0.
constexpr int VB_CODES = 8;
__global__ __launch_bounds__(NT) void vq_bwd_codebook_kernel(const float* __restrict__ z, const float* __restrict__ cb,
                                                             const int64_t* __restrict__ idx, const float* __restrict__ g_loss, float beta,
                                                             int M, int K, int D, float* __restrict__ dcb) {
    __shared__ __attribute__((aligned(16))) float accs[NT / 64][VB_CODES][256];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int k0 = blockIdx.x * VB_CODES;
    for (int i = tid; i < (NT / 64) * VB_CODES * 256; i += NT) (&accs[0][0][0])[i] = 0.0f;
    __syncthreads();
    const float cbeta = -(g_loss ? g_loss[0] : 0.0f) * 2.0f / (float)((long long)M * D) * beta;
    const int per = (M + NT / 64 - 1) / (NT / 64);
    const int lo = wave * per, hi = min(M, lo + per);
    const int d4 = lane * 4;
    for (int base = lo; base < hi; base += 64) {
        const int m_l = base + lane;
        const int k_l = m_l < hi ? (int)idx[m_l] - k0 : -1;
        unsigned long long mask = __builtin_amdgcn_ballot_w64(k_l >= 0 && k_l < VB_CODES);
        while (mask) {
            const int bit = __builtin_ctzll(mask);
            mask &= mask - 1;
            const int kk = __builtin_amdgcn_readlane(k_l, bit);
            const int m = base + bit;
            if (d4 < D) {
                const f32x4 zv = *reinterpret_cast<const f32x4*>(z + (size_t)m * D + d4);
                const f32x4 ev = *reinterpret_cast<const f32x4*>(cb + (size_t)(k0 + kk) * D + d4);
                f32x4 a = *reinterpret_cast<f32x4*>(&accs[wave][kk][d4]);
#pragma unroll
                for (int e = 0; e < 4; ++e) a[e] += cbeta * (zv[e] - ev[e]);
                *reinterpret_cast<f32x4*>(&accs[wave][kk][d4]) = a;
            }
        }
    }
    __syncthreads();
    for (int i = tid; i < VB_CODES * D; i += NT) {
        const int code = i / D, d = i - code * D;
        if (k0 + code < K) {
            float t = accs[0][code][d];
#pragma unroll
            for (int w = 1; w < NT / 64; ++w) t += accs[w][code][d];
            dcb[(size_t)(k0 + code) * D + d] = t;
        }
    }
}

int pick_ksplit(int M, int K) {
    const int rb = mas_cdiv(M, ROWS_PER_BLOCK), ntiles = mas_cdiv(K, CODES_PER_TILE);
    int ks = mas_cdiv(1024, rb);
    if (ks > MAX_KSPLIT) ks = MAX_KSPLIT;
    if (ks > ntiles) ks = ntiles;
    return ks < 1 ? 1 : ks;
}

}  // namespace

// workspace: zz[M] | ee[K] | pval[M*MAX_KSPLIT] | pidx[M*MAX_KSPLIT] | blocksum[ceil(M/4)]
extern "C" size_t mas_vq_workspace(int M, int K) {
    return ((size_t)M + K + (size_t)2 * M * MAX_KSPLIT + (size_t)mas_cdiv(M, NT / 64) + 16) * sizeof(float);
}

extern "C" int mas_vq_argmin_fwd(const float* z, const float* codebook, int M, int K, int D, int64_t* idx, float* zq,
                                 float* sqerr, void* workspace, size_t ws_bytes, void* stream) {
    MAS_ENTER();
    if (!z || !codebook || !idx || !zq || !sqerr || !workspace) MAS_FAIL(MAS_EINVAL, "vq_argmin_fwd: null argument");
    if (M <= 0 || K <= 0) MAS_FAIL(MAS_EINVAL, "vq_argmin_fwd: bad shape M=%d K=%d", M, K);
    if (ws_bytes < mas_vq_workspace(M, K)) MAS_FAIL(MAS_EWORKSPACE, "vq_argmin_fwd: workspace too small");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    float* zz = reinterpret_cast<float*>(workspace);
    float* ee = zz + M;
    float* pval = ee + K;
    int* pidx = reinterpret_cast<int*>(pval + (size_t)M * MAX_KSPLIT);
    float* blocksum = reinterpret_cast<float*>(pidx + (size_t)M * MAX_KSPLIT);
    hipLaunchKernelGGL(row_sqnorm, dim3(mas_cdiv(M, NT / 64)), dim3(NT), 0, s, z, M, D, zz);
    hipLaunchKernelGGL(row_sqnorm, dim3(mas_cdiv(K, NT / 64)), dim3(NT), 0, s, codebook, K, D, ee);
    MAS_CHECK_LAUNCH("vq row_sqnorm");
    const int ksplit = pick_ksplit(M, K);
    const dim3 grid(mas_cdiv(M, ROWS_PER_BLOCK) * ksplit);
    switch (D) {
        case 32: hipLaunchKernelGGL(vq_partial<32>, grid, dim3(NT), 0, s, z, codebook, zz, ee, M, K, ksplit, pval, pidx); break;
        case 64: hipLaunchKernelGGL(vq_partial<64>, grid, dim3(NT), 0, s, z, codebook, zz, ee, M, K, ksplit, pval, pidx); break;
        case 128: hipLaunchKernelGGL(vq_partial<128>, grid, dim3(NT), 0, s, z, codebook, zz, ee, M, K, ksplit, pval, pidx); break;
        case 256: hipLaunchKernelGGL(vq_partial<256>, grid, dim3(NT), 0, s, z, codebook, zz, ee, M, K, ksplit, pval, pidx); break;
        default: MAS_FAIL(MAS_EUNSUPPORTED, "vq_argmin_fwd: codebook_dim %d not in {32,64,128,256}", D);
    }
    MAS_CHECK_LAUNCH("vq_partial");
    const int nb = mas_cdiv(M, NT / 64);
    hipLaunchKernelGGL(vq_finalize, dim3(nb), dim3(NT), 0, s, z, codebook, pval, pidx, M, D, ksplit, idx, zq, blocksum);
    hipLaunchKernelGGL(vq_loss_reduce, dim3(1), dim3(NT), 0, s, blocksum, nb, sqerr);
    MAS_CHECK_LAUNCH("vq_finalize");
    return MAS_OK;
}

extern "C" int mas_vq_bwd(const float* z, const float* codebook, const int64_t* idx, const float* g_zq, const float* g_loss,
                          float beta, int M, int K, int D, float* dz, float* dcodebook, void* stream) {
    MAS_ENTER();
    if (!z || !codebook || !idx) MAS_FAIL(MAS_EINVAL, "vq_bwd: null argument");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    long long total = (long long)M * D;
    int blocks = (int)((total + NT - 1) / NT);
    if (blocks > 4096) blocks = 4096;
    static const int det = mas_env_int("MAS_VQ_BWD_DET", 1);
    const bool fixed_order = det && dcodebook && K > 0 && D <= 256 && D % 4 == 0;       // (else: fp32 atomics into a zeroed dcodebook)
    if (dz || (dcodebook && !fixed_order)) {
        hipLaunchKernelGGL(vq_bwd_kernel, dim3(blocks), dim3(NT), 0, s, z, codebook, idx, g_zq, g_loss, beta, M, D, dz, fixed_order ? nullptr : dcodebook);
        MAS_CHECK_LAUNCH("vq_bwd");
    }
    if (fixed_order) {
        hipLaunchKernelGGL(vq_bwd_codebook_kernel, dim3((unsigned)mas_cdiv(K, VB_CODES)), dim3(NT), 0, s, z, codebook, idx, g_loss, beta, M, K, D, dcodebook);
        MAS_CHECK_LAUNCH("vq_bwd_codebook");
    }
    return MAS_OK;
}
