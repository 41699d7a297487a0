// Decode-time (KV-cached) attention for gfx950: a few new queries against a long cache of keys / values.
// Replaces the cached branch of SelfAttention.forward (reference models/transformer.py:73-115: qkv of the new positions,
// torch.cat with past_k / past_v, calculate_attention on the last rows of the mask, Softmax, matmul) for token-by-token
// sampling.  The reference re-concatenates the whole [B,H,S,hd] cache every step (O(S^2) copies per sample) and materialises
// the [B,H,nq,S] scores; here the cache is a preallocated [B, S_max, H*hd] buffer (the layout nn.Linear emits), the new
// rows are appended in place by the host wrapper, and scores never leave registers.
//
// Regime: HBM / L2 bound (every key and value row is read exactly once per query block: 2 * L * hd * sizeof(T) bytes per
// (batch, head)), no matrix cores -- one dot product per key.  One work-group per (batch, head, query); a lane owns a key:
// it reads the key's contiguous hd-element row (whole 128-byte lines at hd = 64 bf16), keeps an online-softmax partial
// (m, l, o[hd]) over its keys in registers, and the 256 partials are merged once at the end (wave shuffles, then LDS).
// Query i of the block (0 <= i < nq) sees keys 0 .. past + i (causal inside the block, transformer.py:260-263,366-370).
#include "mas_common.h"
#include <math.h>

namespace {

constexpr int DNT = 256;

struct DecodeParams {
    const void* q; const void* k; const void* v; void* o;
    long long q_bs, k_bs, v_bs, o_bs;   // batch strides (elements)
    int ld_q, ld_k, ld_v, ld_o;         // token strides (elements)
    int B, H, nq, past;
    float scale;
};

template <typename T, int HD>
__global__ __launch_bounds__(DNT) void attn_decode_kernel(DecodeParams p) {
    constexpr int EPU = 16 / (int)sizeof(T);
    constexpr int NU = HD / EPU;                 // 16-byte units per row
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int iq = blockIdx.x, bh = blockIdx.y, b = bh / p.H, h = bh % p.H;
    const int L = p.past + iq + 1;               // keys visible to this query

    const T* __restrict__ Q = reinterpret_cast<const T*>(p.q) + (size_t)b * p.q_bs + (size_t)iq * p.ld_q + (size_t)h * HD;
    const T* __restrict__ K = reinterpret_cast<const T*>(p.k) + (size_t)b * p.k_bs + (size_t)h * HD;
    const T* __restrict__ V = reinterpret_cast<const T*>(p.v) + (size_t)b * p.v_bs + (size_t)h * HD;

    float qf[HD];                                // the query, pre-scaled (transformer.py:56: q / sqrt(hd))
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        const u32x4 raw = *reinterpret_cast<const u32x4*>(Q + u * EPU);
        const T* e = reinterpret_cast<const T*>(&raw);
#pragma unroll
        for (int j = 0; j < EPU; ++j) qf[u * EPU + j] = (float)e[j] * p.scale;
    }

    float m = -1e30f, l = 0.0f, o[HD];
#pragma unroll
    for (int d = 0; d < HD; ++d) o[d] = 0.0f;

    for (int key = tid; key < L; key += DNT) {
        const T* kr = K + (size_t)key * p.ld_k;
        const T* vr = V + (size_t)key * p.ld_v;
        float s = 0.0f;
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const u32x4 raw = *reinterpret_cast<const u32x4*>(kr + u * EPU);
            const T* e = reinterpret_cast<const T*>(&raw);
#pragma unroll
            for (int j = 0; j < EPU; ++j) s += qf[u * EPU + j] * (float)e[j];
        }
        const float m_new = fmaxf(m, s);
        const float a = __expf(m - m_new), pv = __expf(s - m_new);
        l = l * a + pv;
        m = m_new;
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const u32x4 raw = *reinterpret_cast<const u32x4*>(vr + u * EPU);
            const T* e = reinterpret_cast<const T*>(&raw);
#pragma unroll
            for (int j = 0; j < EPU; ++j) o[u * EPU + j] = o[u * EPU + j] * a + pv * (float)e[j];
        }
    }

    // ---- merge the 64 lanes of a wave: common maximum, rescale, butterfly sums (fixed order: deterministic) ----
    float mw = m;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) mw = fmaxf(mw, __shfl_xor(mw, off));
    const float f = __expf(m - mw);              // lanes without a key: m = -1e30 -> f = 0 (or 1 when the whole wave is empty: l = o = 0)
    l *= f;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) l += __shfl_xor(l, off);
#pragma unroll
    for (int d = 0; d < HD; ++d) {
        float x = o[d] * f;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) x += __shfl_xor(x, off);
        o[d] = x;
    }
    // ---- merge the 4 waves through LDS ----
    __shared__ float red[4][HD + 2];
    if (lane == 0) {
        red[wave][HD] = mw; red[wave][HD + 1] = l;
    }
    if (lane < HD / 1 && lane < 64) {
        // lane d (and d + 64 for hd = 128) publishes o[d]: every lane holds the full sums, pick by a static unrolled select
#pragma unroll
        for (int d = 0; d < HD; ++d) if ((d & 63) == lane) red[wave][d] = o[d];
    }
    __syncthreads();
    if (wave == 0) {
        const float m0 = red[0][HD], m1 = red[1][HD], m2 = red[2][HD], m3 = red[3][HD];
        const float mt = fmaxf(fmaxf(m0, m1), fmaxf(m2, m3));
        const float f0 = __expf(m0 - mt), f1 = __expf(m1 - mt), f2 = __expf(m2 - mt), f3 = __expf(m3 - mt);
        const float lt = red[0][HD + 1] * f0 + red[1][HD + 1] * f1 + red[2][HD + 1] * f2 + red[3][HD + 1] * f3;
        const float inv = 1.0f / lt;
        T* dst = reinterpret_cast<T*>(p.o) + (size_t)b * p.o_bs + (size_t)iq * p.ld_o + (size_t)h * HD;
        for (int d = lane; d < HD; d += 64)
            dst[d] = (T)((red[0][d] * f0 + red[1][d] * f1 + red[2][d] * f2 + red[3][d] * f3) * inv);
    }
}

template <typename T>
int launch_decode(const DecodeParams& p, int hd, hipStream_t s) {
    const dim3 grid((unsigned)p.nq, (unsigned)(p.B * p.H));
    switch (hd) {
        case 16: hipLaunchKernelGGL((attn_decode_kernel<T, 16>), grid, dim3(DNT), 0, s, p); break;
        case 32: hipLaunchKernelGGL((attn_decode_kernel<T, 32>), grid, dim3(DNT), 0, s, p); break;
        case 64: hipLaunchKernelGGL((attn_decode_kernel<T, 64>), grid, dim3(DNT), 0, s, p); break;
        case 128: hipLaunchKernelGGL((attn_decode_kernel<T, 128>), grid, dim3(DNT), 0, s, p); break;
        default: MAS_FAIL(MAS_EUNSUPPORTED, "attn_decode: head_dim %d not in {16,32,64,128}", hd);
    }
    MAS_CHECK_LAUNCH("attn_decode");
    return MAS_OK;
}

}  // namespace

extern "C" int mas_attn_decode(const void* q, const void* k_cache, const void* v_cache, void* o, int dtype, int B, int H, int nq,
                               int past, int hd, int ld_q, int ld_k, int ld_v, int ld_o, long long q_bs, long long k_bs,
                               long long v_bs, long long o_bs, float scale, void* stream) {
    MAS_ENTER();
    if (!q || !k_cache || !v_cache || !o) MAS_FAIL(MAS_EINVAL, "attn_decode: null argument");
    if (B <= 0 || H <= 0 || nq <= 0 || past < 0) MAS_FAIL(MAS_EINVAL, "attn_decode: bad shape B=%d H=%d nq=%d past=%d", B, H, nq, past);
    const size_t esz = mas_esize(dtype);
    const int epu = 16 / (int)esz;
    if ((ld_q % epu) || (ld_k % epu) || (ld_v % epu) || (q_bs % epu) || (k_bs % epu) || (v_bs % epu) ||
        ((reinterpret_cast<uintptr_t>(q) | reinterpret_cast<uintptr_t>(k_cache) | reinterpret_cast<uintptr_t>(v_cache)) & 15))
        MAS_FAIL(MAS_EUNSUPPORTED, "attn_decode: q / k / v rows must be 16-byte aligned");
    DecodeParams p;
    p.q = q; p.k = k_cache; p.v = v_cache; p.o = o;
    p.q_bs = q_bs; p.k_bs = k_bs; p.v_bs = v_bs; p.o_bs = o_bs;
    p.ld_q = ld_q; p.ld_k = ld_k; p.ld_v = ld_v; p.ld_o = ld_o;
    p.B = B; p.H = H; p.nq = nq; p.past = past; p.scale = scale;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (dtype == MAS_BF16) return launch_decode<bf16_t>(p, hd, s);
    if (dtype == MAS_F32) return launch_decode<float>(p, hd, s);
    MAS_FAIL(MAS_EUNSUPPORTED, "attn_decode: dtype %d", dtype);
}
