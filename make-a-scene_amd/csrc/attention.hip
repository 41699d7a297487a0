// Causal multi-head self-attention forward for gfx950 (flash style: the [S,S] score matrix is never written).
// Replaces SelfAttention.calculate_attention + Softmax + matmul(probs, v) of the reference
// (models/transformer.py:44-71, 90-97) in its training configuration: scores (q/sqrt(hd)) k^T, the
// multiplicative causal mask with -1e4 fill (transformer.py:63; == exp(.)=0 in fp32), the CogView PB-relax shift
// (transformer.py:64-70; a per-(batch,head) constant, i.e. softmax invariant), softmax over keys, . v.
// The reference materialises >= 4 copies of the [B,H,1536,1536] fp32 score tensor (151 MB per sample-layer).
//
// Work-group = 4 waves = 128 queries of one (batch, head); each wave owns 32 queries and walks 32-key tiles.
// "Swapped" products keep every softmax reduction inside a lane (+ one lane^32 exchange):
//   S^T[key][query] = K_tile . Q^T      (MFMA A = K rows from LDS, B = Q^T kept in registers)
//   O^T[d][query]  += V^T . P^T         (MFMA A = V^T from a transposed LDS image, B = P^T straight from the
//                                        score accumulators: lane = query, registers = keys)
// fp32 online softmax (running max / sum per query), bf16 or exact-fp32 MFMA operands (template T).
#include "mas_common.h"
#include <math.h>

namespace {

constexpr int NT = 256, QT = 128, KT = 32;

struct AttnParams {
    const void* q; const void* k; const void* v; void* o; float* lse;
    long long q_bs, k_bs, v_bs;     // batch strides (elements)
    int ld_q, ld_k, ld_v;           // token strides (elements)
    int B, H, S;
    float scale;
};

template <typename T, int HD>
__global__ __launch_bounds__(NT) void attn_causal_fwd_kernel(AttnParams p) {
    using V8 = typename Vec8<T>::type;
    constexpr int EPU = 16 / (int)sizeof(T);
    constexpr int DT = HD < 32 ? 32 : HD;       // O^T rows padded to a 32-row MFMA tile
    constexpr int NKK = HD / 16;                // k-steps of the QK^T product
    constexpr int NMI = DT / 32;                // 32-row tiles of O^T
    constexpr int KS_ = HD + EPU;               // K tile row stride (elements)
    constexpr int VS_ = KT + EPU;               // V^T row stride (elements)
    __shared__ __attribute__((aligned(16))) T ktile[KT * KS_];
    __shared__ __attribute__((aligned(16))) T vtile[DT * VS_];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 5, l31 = lane & 31;
    const int bh = blockIdx.y, b = bh / p.H, h = bh % p.H;
    const int q0 = blockIdx.x * QT;
    const int qw = q0 + wave * 32;              // first query of this wave
    const int query = qw + l31;

    const T* __restrict__ Q = reinterpret_cast<const T*>(p.q) + (size_t)b * p.q_bs + (size_t)h * HD;
    const T* __restrict__ K = reinterpret_cast<const T*>(p.k) + (size_t)b * p.k_bs + (size_t)h * HD;
    const T* __restrict__ V = reinterpret_cast<const T*>(p.v) + (size_t)b * p.v_bs + (size_t)h * HD;

    // Q^T fragments (B operand): lane = query column, 8 consecutive head dims per k-step
    V8 qf[NKK];
#pragma unroll
    for (int kk = 0; kk < NKK; ++kk) {
        V8 v = zero8<T>();
        if (query < p.S) {
            const T* src = Q + (size_t)query * p.ld_q + kk * 16 + g * 8;
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = src[j];
        }
        qf[kk] = v;
    }

    f32x16 oacc[NMI];
#pragma unroll
    for (int i = 0; i < NMI; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[i][r] = 0.0f;
    float m = -1e30f, l = 0.0f;

    // vtile rows >= HD (padding of a 16-dim head to a 32-row tile) stay zero
    for (int i = tid; i < DT * VS_; i += NT) vtile[i] = (T)0.0f;

    const int q_last = min(q0 + QT, p.S) - 1;   // last query of the work-group: keys beyond it are never needed
    for (int k0 = 0; k0 <= q_last; k0 += KT) {
        __syncthreads();                        // previous tile fully consumed
        // ---- stage K rows and V^T (zero rows for keys >= S) --------------------------------------
        for (int u = tid; u < KT * (HD / EPU); u += NT) {
            const int key = u / (HD / EPU), cu = u % (HD / EPU);
            T kv[EPU], vv[EPU];
#pragma unroll
            for (int e = 0; e < EPU; ++e) { kv[e] = (T)0.0f; vv[e] = (T)0.0f; }
            if (k0 + key < p.S) {
                const T* ks = K + (size_t)(k0 + key) * p.ld_k + cu * EPU;
                const T* vs = V + (size_t)(k0 + key) * p.ld_v + cu * EPU;
#pragma unroll
                for (int e = 0; e < EPU; ++e) { kv[e] = ks[e]; vv[e] = vs[e]; }
            }
#pragma unroll
            for (int e = 0; e < EPU; ++e) {
                ktile[key * KS_ + cu * EPU + e] = kv[e];
                vtile[(cu * EPU + e) * VS_ + key] = vv[e];
            }
        }
        __syncthreads();
        if (k0 > qw + 31) continue;             // tile entirely above this wave's diagonal (wave-uniform)

        // ---- S^T = K . Q^T ---------------------------------------------------------------------------
        f32x16 s;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.0f;
#pragma unroll
        for (int kk = 0; kk < NKK; ++kk) {
            const V8 kf = ld8<T>(&ktile[l31 * KS_ + kk * 16 + g * 8]);
            mma16(s, kf, qf[kk]);
        }
        // ---- online softmax over keys (rows of S^T): lane = query, registers = keys -----------------
        float tmax = -1e30f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = k0 + acc_row(lane, r);
            const float sv = (key <= query && key < p.S) ? s[r] * p.scale : -1e30f;
            s[r] = sv;
            tmax = fmaxf(tmax, sv);
        }
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
        const float m_new = fmaxf(m, tmax);
        const float alpha = __expf(m - m_new);
        float rsum = 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) { const float pv = __expf(s[r] - m_new); s[r] = pv; rsum += pv; }
        rsum += __shfl_xor(rsum, 32);
        l = l * alpha + rsum;
        m = m_new;
#pragma unroll
        for (int i = 0; i < NMI; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) oacc[i][r] *= alpha;
        // ---- O^T += V^T . P^T : the B operand is the probabilities themselves.  MFMA step t pairs element j of
        // lane group g with key(t,g,j) = 16t + 8(j>>2) + 4g + (j&3) -- exactly the key owned by accumulator
        // register 8t+j -- so P needs no shuffle; V^T is read with the same key permutation (two 4-key runs).
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            V8 pf;
#pragma unroll
            for (int j = 0; j < 8; ++j) pf[j] = (T)s[8 * t + j];
#pragma unroll
            for (int i = 0; i < NMI; ++i) {
                const T* vr = &vtile[(i * 32 + l31) * VS_ + 16 * t + 4 * g];
                V8 vf;
#pragma unroll
                for (int j = 0; j < 4; ++j) { vf[j] = vr[j]; vf[4 + j] = vr[8 + j]; }
                mma16(oacc[i], vf, pf);
            }
        }
    }

    // ---- normalise and store: lane = query, accumulator rows = head dims -------------------------------
    if (query < p.S) {
        const float inv = 1.0f / l;
        T* dst = reinterpret_cast<T*>(p.o) + ((size_t)b * p.S + query) * ((size_t)p.H * HD) + (size_t)h * HD;
#pragma unroll
        for (int i = 0; i < NMI; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int d = i * 32 + acc_row(lane, r);
                if (d < HD) dst[d] = (T)(oacc[i][r] * inv);
            }
        if (p.lse && g == 0) p.lse[((size_t)b * p.H + h) * p.S + query] = m + __logf(l);
    }
}

template <typename T>
int launch_hd(const AttnParams& p, int hd, hipStream_t s) {
    const dim3 grid(mas_cdiv(p.S, QT), p.B * p.H);
    switch (hd) {
        case 16: hipLaunchKernelGGL((attn_causal_fwd_kernel<T, 16>), grid, dim3(NT), 0, s, p); break;
        case 32: hipLaunchKernelGGL((attn_causal_fwd_kernel<T, 32>), grid, dim3(NT), 0, s, p); break;
        case 64: hipLaunchKernelGGL((attn_causal_fwd_kernel<T, 64>), grid, dim3(NT), 0, s, p); break;
        case 128: hipLaunchKernelGGL((attn_causal_fwd_kernel<T, 128>), grid, dim3(NT), 0, s, p); break;
        default: MAS_FAIL(MAS_EUNSUPPORTED, "attn_causal_fwd: head_dim %d not in {16,32,64,128}", hd);
    }
    MAS_CHECK_LAUNCH("attn_causal_fwd");
    return MAS_OK;
}

}  // namespace

extern "C" int mas_attn_causal_fwd(const void* q, const void* k, const void* v, void* o, float* lse, int dtype, int B, int H,
                                   int S, int hd, int ld_q, int ld_k, int ld_v, long long q_bs, long long k_bs,
                                   long long v_bs, float scale, void* stream) {
    MAS_ENTER();
    if (!q || !k || !v || !o) MAS_FAIL(MAS_EINVAL, "attn_causal_fwd: null argument");
    if (B <= 0 || H <= 0 || S <= 0) MAS_FAIL(MAS_EINVAL, "attn_causal_fwd: bad shape B=%d H=%d S=%d", B, H, S);
    AttnParams p;
    p.q = q; p.k = k; p.v = v; p.o = o; p.lse = lse;
    p.q_bs = q_bs; p.k_bs = k_bs; p.v_bs = v_bs; p.ld_q = ld_q; p.ld_k = ld_k; p.ld_v = ld_v;
    p.B = B; p.H = H; p.S = S; p.scale = scale;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (dtype == MAS_BF16) return launch_hd<bf16_t>(p, hd, s);
    if (dtype == MAS_F32) return launch_hd<float>(p, hd, s);
    MAS_FAIL(MAS_EUNSUPPORTED, "attn_causal_fwd: dtype %d", dtype);
}
