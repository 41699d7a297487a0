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


// ---------------------------------------------------------------------------------------------------------
// Backward.  dS = P o (dP - D) * scale with P = exp(S*scale - lse), dP = dO V^T, D[q] = sum_d dO[q,d] O[q,d].
//   delta kernel : D
//   dkv kernel   : a wave owns 32 keys (K, V fragments in registers), walks 32-query tiles at / below its diagonal:
//                  S = Q K^T and dP = dO V^T with lane = key, registers = queries, so P and dS feed
//                  dV^T += dO^T P and dK^T += Q^T dS directly as MFMA B operands (same trick as the forward);
//                  every dK / dV element is written exactly once -- no atomics.
//   dq kernel    : a wave owns 32 queries, walks 32-key tiles: S^T = K Q^T, dP^T = V dO^T (lane = query),
//                  dQ^T += K^T dS^T.
struct AttnBwdParams {
    const void* q; const void* k; const void* v; const void* o; const void* dout; const float* lse; float* delta;
    void* dq; void* dk; void* dv;
    long long bs; int ld;           // q/k/v and dq/dk/dv share the fused-qkv addressing: ptr[b*bs + s*ld + h*hd + d]
    int B, H, S;
    float scale;
};

template <typename T, int HD>
__global__ __launch_bounds__(NT) void attn_bwd_delta_kernel(AttnBwdParams p) {
    const long long idx = (long long)blockIdx.x * NT + threadIdx.x;     // (b, h, query)
    const long long total = (long long)p.B * p.H * p.S;
    if (idx >= total) return;
    const int query = (int)(idx % p.S); const int h = (int)((idx / p.S) % p.H); const int b = (int)(idx / ((long long)p.S * p.H));
    const T* o = reinterpret_cast<const T*>(p.o) + ((size_t)b * p.S + query) * ((size_t)p.H * HD) + (size_t)h * HD;
    const T* g = reinterpret_cast<const T*>(p.dout) + ((size_t)b * p.S + query) * ((size_t)p.H * HD) + (size_t)h * HD;
    float acc = 0.0f;
#pragma unroll
    for (int d = 0; d < HD; ++d) acc += (float)o[d] * (float)g[d];
    p.delta[idx] = acc;
}

// stages a 32-row x HD tile both row-major (rows x HD, stride RS) and transposed (DT x 32, stride TS); rows >= S are zero
template <typename T, int HD>
__device__ __forceinline__ void stage_tile(const T* __restrict__ src, long long row_stride, int row0, int S, T* rowmaj, int RS,
                                           T* transp, int TS, int tid) {
    constexpr int EPU = 16 / (int)sizeof(T);
    for (int u = tid; u < KT * (HD / EPU); u += NT) {
        const int row = u / (HD / EPU), cu = u % (HD / EPU);
        T v[EPU];
#pragma unroll
        for (int e = 0; e < EPU; ++e) v[e] = (T)0.0f;
        if (row0 + row < S) {
            const T* sp = src + (size_t)(row0 + row) * row_stride + cu * EPU;
#pragma unroll
            for (int e = 0; e < EPU; ++e) v[e] = sp[e];
        }
#pragma unroll
        for (int e = 0; e < EPU; ++e) {
            if (rowmaj) rowmaj[row * RS + cu * EPU + e] = v[e];
            if (transp) transp[(cu * EPU + e) * TS + row] = v[e];
        }
    }
}

// A operand (rows = head dims, k = the 32 tile rows in the accumulator-register order) from a transposed image
template <typename T>
__device__ __forceinline__ typename Vec8<T>::type ld_perm(const T* trow, int t, int g) {
    const T* r = trow + 16 * t + 4 * g;
    typename Vec8<T>::type v;
#pragma unroll
    for (int j = 0; j < 4; ++j) { v[j] = r[j]; v[4 + j] = r[8 + j]; }
    return v;
}

template <typename T, int HD>
__global__ __launch_bounds__(NT) void attn_bwd_dkv_kernel(AttnBwdParams p) {
    using V8 = typename Vec8<T>::type;
    constexpr int EPU = 16 / (int)sizeof(T);
    constexpr int DT = HD < 32 ? 32 : HD, NKK = HD / 16, NMI = DT / 32;
    constexpr int RS = HD + EPU, TS = KT + EPU;
    __shared__ __attribute__((aligned(16))) T qrow[KT * RS];
    __shared__ __attribute__((aligned(16))) T grow[KT * RS];
    __shared__ __attribute__((aligned(16))) T qtr[DT * TS];
    __shared__ __attribute__((aligned(16))) T gtr[DT * TS];
    __shared__ float s_lse[KT], s_delta[KT];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 5, l31 = lane & 31;
    const int bh = blockIdx.y, b = bh / p.H, h = bh % p.H;
    const int k0 = blockIdx.x * QT;                 // 128 keys per work-group
    const int kw = k0 + wave * 32, key = kw + l31;  // this lane's key (MFMA column)
    const size_t head = (size_t)b * p.bs + (size_t)h * HD;
    const T* __restrict__ Q = reinterpret_cast<const T*>(p.q) + head;
    const T* __restrict__ K = reinterpret_cast<const T*>(p.k) + head;
    const T* __restrict__ V = reinterpret_cast<const T*>(p.v) + head;
    const T* __restrict__ G = reinterpret_cast<const T*>(p.dout) + ((size_t)b * p.S) * ((size_t)p.H * HD) + (size_t)h * HD;
    const long long g_stride = (long long)p.H * HD;

    V8 kf[NKK], vf[NKK];                            // B operands: lane = key column, 8 consecutive head dims
#pragma unroll
    for (int kk = 0; kk < NKK; ++kk) {
        kf[kk] = zero8<T>(); vf[kk] = zero8<T>();
        if (key < p.S) {
            const T* ks = K + (size_t)key * p.ld + kk * 16 + g * 8;
            const T* vs = V + (size_t)key * p.ld + kk * 16 + g * 8;
#pragma unroll
            for (int j = 0; j < 8; ++j) { kf[kk][j] = ks[j]; vf[kk][j] = vs[j]; }
        }
    }
    f32x16 dk[NMI], dv[NMI];
#pragma unroll
    for (int i = 0; i < NMI; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) { dk[i][r] = 0.0f; dv[i][r] = 0.0f; }
    for (int i = tid; i < DT * TS; i += NT) { qtr[i] = (T)0.0f; gtr[i] = (T)0.0f; }

    for (int q0 = (k0 / KT) * KT; q0 < p.S; q0 += KT) {     // causal: only queries >= the first key of the work-group
        __syncthreads();
        stage_tile<T, HD>(Q, p.ld, q0, p.S, qrow, RS, qtr, TS, tid);
        stage_tile<T, HD>(G, g_stride, q0, p.S, grow, RS, gtr, TS, tid);
        if (tid < KT) {
            const bool ok = q0 + tid < p.S;
            s_lse[tid] = ok ? p.lse[((size_t)b * p.H + h) * p.S + q0 + tid] : 0.0f;
            s_delta[tid] = ok ? p.delta[((size_t)b * p.H + h) * p.S + q0 + tid] : 0.0f;
        }
        __syncthreads();
        if (q0 + KT - 1 < kw) continue;             // every query of the tile precedes this wave's keys (wave-uniform)
        f32x16 s, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] = 0.0f; dp[r] = 0.0f; }
#pragma unroll
        for (int kk = 0; kk < NKK; ++kk) {          // rows = queries (A from LDS), columns = keys (B in registers)
            mma16(s, ld8<T>(&qrow[l31 * RS + kk * 16 + g * 8]), kf[kk]);
            mma16(dp, ld8<T>(&grow[l31 * RS + kk * 16 + g * 8]), vf[kk]);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int qi = acc_row(lane, r), query = q0 + qi;
            const bool ok = (key <= query) && (query < p.S) && (key < p.S);
            const float pv = ok ? __expf(s[r] * p.scale - s_lse[qi]) : 0.0f;
            s[r] = pv;                               // P[query][key]
            dp[r] = pv * (dp[r] - s_delta[qi]) * p.scale;   // dS[query][key]
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            V8 pf, sf;
#pragma unroll
            for (int j = 0; j < 8; ++j) { pf[j] = (T)s[8 * t + j]; sf[j] = (T)dp[8 * t + j]; }
#pragma unroll
            for (int i = 0; i < NMI; ++i) {
                mma16(dv[i], ld_perm<T>(&gtr[(i * 32 + l31) * TS], t, g), pf);    // dV^T += dO^T P
                mma16(dk[i], ld_perm<T>(&qtr[(i * 32 + l31) * TS], t, g), sf);    // dK^T += Q^T dS
            }
        }
    }
    if (key < p.S) {
        T* dkp = reinterpret_cast<T*>(p.dk) + head + (size_t)key * p.ld;
        T* dvp = reinterpret_cast<T*>(p.dv) + head + (size_t)key * p.ld;
#pragma unroll
        for (int i = 0; i < NMI; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int d = i * 32 + acc_row(lane, r);
                if (d < HD) { dkp[d] = (T)dk[i][r]; dvp[d] = (T)dv[i][r]; }
            }
    }
}

template <typename T, int HD>
__global__ __launch_bounds__(NT) void attn_bwd_dq_kernel(AttnBwdParams p) {
    using V8 = typename Vec8<T>::type;
    constexpr int EPU = 16 / (int)sizeof(T);
    constexpr int DT = HD < 32 ? 32 : HD, NKK = HD / 16, NMI = DT / 32;
    constexpr int RS = HD + EPU, TS = KT + EPU;
    __shared__ __attribute__((aligned(16))) T krow[KT * RS];
    __shared__ __attribute__((aligned(16))) T vrow[KT * RS];
    __shared__ __attribute__((aligned(16))) T ktr[DT * TS];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 5, l31 = lane & 31;
    const int bh = blockIdx.y, b = bh / p.H, h = bh % p.H;
    const int q0 = blockIdx.x * QT, qw = q0 + wave * 32, query = qw + l31;
    const size_t head = (size_t)b * p.bs + (size_t)h * HD;
    const T* __restrict__ Q = reinterpret_cast<const T*>(p.q) + head;
    const T* __restrict__ K = reinterpret_cast<const T*>(p.k) + head;
    const T* __restrict__ V = reinterpret_cast<const T*>(p.v) + head;
    const T* __restrict__ G = reinterpret_cast<const T*>(p.dout) + ((size_t)b * p.S) * ((size_t)p.H * HD) + (size_t)h * HD;

    V8 qf[NKK], gf[NKK];                            // B operands: lane = query column
#pragma unroll
    for (int kk = 0; kk < NKK; ++kk) {
        qf[kk] = zero8<T>(); gf[kk] = zero8<T>();
        if (query < p.S) {
            const T* qs = Q + (size_t)query * p.ld + kk * 16 + g * 8;
            const T* gs = G + (size_t)query * ((size_t)p.H * HD) + kk * 16 + g * 8;
#pragma unroll
            for (int j = 0; j < 8; ++j) { qf[kk][j] = qs[j]; gf[kk][j] = gs[j]; }
        }
    }
    const float my_lse = query < p.S ? p.lse[((size_t)b * p.H + h) * p.S + query] : 0.0f;
    const float my_delta = query < p.S ? p.delta[((size_t)b * p.H + h) * p.S + query] : 0.0f;
    f32x16 dq[NMI];
#pragma unroll
    for (int i = 0; i < NMI; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) dq[i][r] = 0.0f;
    for (int i = tid; i < DT * TS; i += NT) ktr[i] = (T)0.0f;

    const int q_last = min(q0 + QT, p.S) - 1;
    for (int k0 = 0; k0 <= q_last; k0 += KT) {
        __syncthreads();
        stage_tile<T, HD>(K, p.ld, k0, p.S, krow, RS, ktr, TS, tid);
        stage_tile<T, HD>(V, p.ld, k0, p.S, vrow, RS, (T*)nullptr, TS, tid);
        __syncthreads();
        if (k0 > qw + 31) continue;
        f32x16 s, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] = 0.0f; dp[r] = 0.0f; }
#pragma unroll
        for (int kk = 0; kk < NKK; ++kk) {          // rows = keys (A from LDS), columns = queries (B in registers)
            mma16(s, ld8<T>(&krow[l31 * RS + kk * 16 + g * 8]), qf[kk]);
            mma16(dp, ld8<T>(&vrow[l31 * RS + kk * 16 + g * 8]), gf[kk]);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = k0 + acc_row(lane, r);
            const bool ok = (key <= query) && (key < p.S);
            const float pv = ok ? __expf(s[r] * p.scale - my_lse) : 0.0f;
            dp[r] = pv * (dp[r] - my_delta) * p.scale;      // dS^T[key][query]
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            V8 sf;
#pragma unroll
            for (int j = 0; j < 8; ++j) sf[j] = (T)dp[8 * t + j];
#pragma unroll
            for (int i = 0; i < NMI; ++i) mma16(dq[i], ld_perm<T>(&ktr[(i * 32 + l31) * TS], t, g), sf);   // dQ^T += K^T dS^T
        }
    }
    if (query < p.S) {
        T* dst = reinterpret_cast<T*>(p.dq) + head + (size_t)query * p.ld;
#pragma unroll
        for (int i = 0; i < NMI; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int d = i * 32 + acc_row(lane, r);
                if (d < HD) dst[d] = (T)dq[i][r];
            }
    }
}

template <typename T, int HD>
int launch_bwd(const AttnBwdParams& p, hipStream_t s) {
    const long long rows = (long long)p.B * p.H * p.S;
    hipLaunchKernelGGL((attn_bwd_delta_kernel<T, HD>), dim3((unsigned)((rows + NT - 1) / NT)), dim3(NT), 0, s, p);
    const dim3 grid(mas_cdiv(p.S, QT), p.B * p.H);
    hipLaunchKernelGGL((attn_bwd_dkv_kernel<T, HD>), grid, dim3(NT), 0, s, p);
    hipLaunchKernelGGL((attn_bwd_dq_kernel<T, HD>), grid, dim3(NT), 0, s, p);
    MAS_CHECK_LAUNCH("attn_causal_bwd");
    return MAS_OK;
}

template <typename T>
int launch_bwd_hd(const AttnBwdParams& p, int hd, hipStream_t s) {
    switch (hd) {
        case 16: return launch_bwd<T, 16>(p, s);
        case 32: return launch_bwd<T, 32>(p, s);
        case 64: return launch_bwd<T, 64>(p, s);
        case 128: return launch_bwd<T, 128>(p, s);
        default: MAS_FAIL(MAS_EUNSUPPORTED, "attn_causal_bwd: head_dim %d not in {16,32,64,128}", hd);
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

extern "C" int mas_attn_causal_bwd(const void* qkv, const void* o, const void* dout, const float* lse, float* delta, void* dqkv,
                                   int dtype, int B, int H, int S, int hd, float scale, void* stream) {
    MAS_ENTER();
    if (!qkv || !o || !dout || !lse || !delta || !dqkv) MAS_FAIL(MAS_EINVAL, "attn_causal_bwd: null argument");
    if (B <= 0 || H <= 0 || S <= 0) MAS_FAIL(MAS_EINVAL, "attn_causal_bwd: bad shape");
    const size_t esz = mas_esize(dtype);
    const int d = H * hd;
    AttnBwdParams p;
    const unsigned char* x = reinterpret_cast<const unsigned char*>(qkv);
    unsigned char* gx = reinterpret_cast<unsigned char*>(dqkv);
    p.q = x; p.k = x + (size_t)d * esz; p.v = x + (size_t)2 * d * esz;
    p.dq = gx; p.dk = gx + (size_t)d * esz; p.dv = gx + (size_t)2 * d * esz;
    p.o = o; p.dout = dout; p.lse = lse; p.delta = delta;
    p.ld = 3 * d; p.bs = (long long)S * 3 * d; p.B = B; p.H = H; p.S = S; p.scale = scale;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (dtype == MAS_BF16) return launch_bwd_hd<bf16_t>(p, hd, s);
    if (dtype == MAS_F32) return launch_bwd_hd<float>(p, hd, s);
    MAS_FAIL(MAS_EUNSUPPORTED, "attn_causal_bwd: dtype %d", dtype);
}
