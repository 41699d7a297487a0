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
#include <stdint.h>
#include <stdlib.h>
#include <type_traits>

namespace {

constexpr int NT = 256, QT = 128, KT = 32;

struct AttnParams {
    const void* q; const void* k; const void* v; void* o; float* lse;
    long long q_bs, k_bs, v_bs;     // batch strides (elements)
    int ld_q, ld_k, ld_v;           // token strides (elements)
    int B, H, S;
    float scale;
    int lpt;                        // grid = (B * H, tiles) instead of (tiles, B * H): see fa_lpt()
};

// Causal attention work per work-group grows linearly with its tile index, and the grid (12 tiles x 128 heads at S = 1536, B = 8) is
// 1.5 rounds of the 1024 resident work-groups.  With the tile index in blockIdx.x the dispatcher hands out heavy and light work-groups
// interleaved, head by head, and the last heads' heaviest tiles START late: makespan ~ 7 + 24 tile-times where the balanced load is
// 19.5.  With (B * H) in blockIdx.x and the heaviest tile in blockIdx.y = 0 the dispatch order is longest-processing-time-first, and
// a head's work-groups all land on XCD (head % 8): its K / V stay in one L2 (49.0 -> 46.2 ms per MakeAScene step, profiles/r03_attn_lpt.txt).
static int fa_lpt() { return 1; }

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
// bf16 fast path of the forward (head_dim 64 / 128, 16-byte aligned rows): same "swapped" products, but
//   * 64-key tiles, two LDS stages: the next tile's K / V rows are in flight in registers (16-byte coalesced
//     global loads) across the whole compute phase and stored to the other stage afterwards -- one barrier per tile;
//   * K and V are both staged in their natural [key][head_dim] layout; the V^T operand of O^T += V^T P^T is fetched
//     with the LDS transpose read (ds_read_b64_tr_b16; semantics in conv_wgrad.hip), whose per-lane addresses
//     also absorb the accumulator-order key permutation key(t,g,j) = 16t + 8(j>>2) + 4g + (j&3);
//   * exp2 softmax (scale*log2(e) folded into one FMA), masking only on tiles that touch the diagonal or S;
//   * heaviest (last) query blocks are launched first.
// Row strides: K 2*HD+16 B (16 consecutive keys x 16 B tile the 64 banks for ds_read_b128), V 2*HD+64 B (4
// consecutive keys x 64 B tile the 256-byte bank row for the transpose read).
typedef __attribute__((ext_vector_type(4))) short fa_s16x4;
__device__ __forceinline__ bf16x8 fa_tr_frag(const unsigned char* a0, const unsigned char* a1) {
    const fa_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) fa_s16x4*)a0);
    const fa_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) fa_s16x4*)a1);
    const __attribute__((ext_vector_type(8))) short v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return *reinterpret_cast<const bf16x8*>(&v);
}

// value of the lane 32 away (the other half of an MFMA column), through v_permlane32_swap: no LDS round trip
__device__ __forceinline__ float fa_other_half(float x) {
    const unsigned u = __builtin_bit_cast(unsigned, x);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);    // r[0] = {lo, lo}, r[1] = {hi, hi}
    const unsigned lo = r[0], hi = r[1];
    return __builtin_bit_cast(float, (threadIdx.x & 32) ? lo : hi);
}
typedef __attribute__((ext_vector_type(2))) float fa_f32x2;

template <int HD>
struct FaGeo {
    static constexpr int KT2 = 64, RSK = HD * 2 + 16, RSV = HD * 2 + 64;
    static constexpr int K_BYTES = KT2 * RSK, STAGE = KT2 * (RSK + RSV);
    static constexpr size_t LDS_BYTES = 2 * (size_t)STAGE;
};

template <int HD>
__global__ __launch_bounds__(NT, 2) void attn_causal_fwd_bf16_kernel(AttnParams p) {
    using T = bf16_t;
    using G = FaGeo<HD>;
    constexpr int KT2 = G::KT2, RSK = G::RSK, RSV = G::RSV, NKK = HD / 16, NMI = HD / 32;
    constexpr int UPR = HD / 8;                      // 16-byte slots per row
    constexpr int SPT = KT2 * UPR / NT;              // slots per thread and tensor: 2 (HD 64) / 4 (HD 128)
    extern __shared__ __attribute__((aligned(16))) unsigned char fa_smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar: wave-level branches stay branches
    const int g = lane >> 5, l31 = lane & 31, G16 = (lane >> 4) & 1, sl = lane & 15;
    const int bh = blockIdx.y, b = bh / p.H, h = bh % p.H;
    const int q0 = ((int)gridDim.x - 1 - (int)blockIdx.x) * QT;     // heavy blocks first
    const int qw = q0 + wave * 32, query = qw + l31;

    const T* __restrict__ Q = reinterpret_cast<const T*>(p.q) + (size_t)b * p.q_bs + (size_t)h * HD;
    const T* __restrict__ K = reinterpret_cast<const T*>(p.k) + (size_t)b * p.k_bs + (size_t)h * HD;
    const T* __restrict__ V = reinterpret_cast<const T*>(p.v) + (size_t)b * p.v_bs + (size_t)h * HD;

    bf16x8 qf[NKK];                                  // Q^T fragments (B operand): lane = query, 8 consecutive head dims
#pragma unroll
    for (int kk = 0; kk < NKK; ++kk) {
        u32x4 raw = u32x4{0u, 0u, 0u, 0u};
        if (query < p.S) raw = *reinterpret_cast<const u32x4*>(Q + (size_t)query * p.ld_q + kk * 16 + g * 8);
        qf[kk] = *reinterpret_cast<const bf16x8*>(&raw);
    }
    f32x16 oacc[NMI];
#pragma unroll
    for (int i = 0; i < NMI; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[i][r] = 0.0f;
    float m = -1e30f, l = 0.0f;                      // running max (log2 domain) and sum per query
    const float c2 = p.scale * 1.4426950408889634f;

    u32x4 rk[SPT], rv[SPT];
    auto issue = [&](int k0) {
#pragma unroll
        for (int i = 0; i < SPT; ++i) {
            const int u = tid + i * NT, key = k0 + u / UPR, cu = u % UPR;
            const bool ok = key < p.S;
            const size_t row = ok ? (size_t)key : 0;
            rk[i] = *reinterpret_cast<const u32x4*>(K + row * p.ld_k + cu * 8);
            rv[i] = *reinterpret_cast<const u32x4*>(V + row * p.ld_v + cu * 8);
            if (!ok) { rk[i] = u32x4{0u, 0u, 0u, 0u}; rv[i] = u32x4{0u, 0u, 0u, 0u}; }
        }
    };
    auto commit = [&](unsigned char* stage) {
#pragma unroll
        for (int i = 0; i < SPT; ++i) {
            const int u = tid + i * NT, key = u / UPR, cu = u % UPR;
            *reinterpret_cast<u32x4*>(stage + key * RSK + cu * 16) = rk[i];
            *reinterpret_cast<u32x4*>(stage + G::K_BYTES + key * RSV + cu * 16) = rv[i];
        }
    };

    const int q_last = min(q0 + QT, p.S) - 1;        // keys beyond the work-group's last query are never needed
    const int n_tiles = q_last / KT2 + 1;
    issue(0);
    commit(fa_smem);
    __syncthreads();
    int cur = 0;
    for (int it = 0; it < n_tiles; ++it) {
        const int k0 = it * KT2;
        const bool more = it + 1 < n_tiles;
        if (more) issue(k0 + KT2);
        if (k0 <= qw + 31) {                         // wave-uniform: otherwise the tile is entirely above this wave's diagonal
            const unsigned char* kt = fa_smem + cur * G::STAGE;
            const unsigned char* vt = kt + G::K_BYTES;
            f32x16 s[2];
#pragma unroll
            for (int sub = 0; sub < 2; ++sub) {
#pragma unroll
                for (int r = 0; r < 16; ++r) s[sub][r] = 0.0f;
#pragma unroll
                for (int kk = 0; kk < NKK; ++kk) {
                    const bf16x8 kf = *reinterpret_cast<const bf16x8*>(kt + (sub * 32 + l31) * RSK + kk * 32 + g * 16);
                    mma16(s[sub], kf, qf[kk]);
                }
            }
            // softmax in the log2 domain: p = exp2(s*c2 - m).  c2 > 0, so the row maximum is taken on the raw scores.
            if ((k0 + KT2 - 1 > qw) || (k0 + KT2 > p.S)) {          // tile touches the diagonal or the sequence end: mask
#pragma unroll
                for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int key = k0 + sub * 32 + acc_row(lane, r);
                        if (!(key <= query && key < p.S)) s[sub][r] = -1e30f;
                    }
            }
            float tmax = -1e30f;
#pragma unroll
            for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                for (int r = 0; r < 16; ++r) tmax = fmaxf(tmax, s[sub][r]);
            tmax = fmaxf(tmax, fa_other_half(tmax)) * c2;
            const float m_new = fmaxf(m, tmax);
            const float alpha = __builtin_amdgcn_exp2f(m - m_new);
            fa_f32x2 rs2 = fa_f32x2{0.0f, 0.0f};
            const fa_f32x2 c22 = fa_f32x2{c2, c2}, nm2 = fa_f32x2{-m_new, -m_new};
#pragma unroll
            for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                for (int r = 0; r < 16; r += 2) {                    // packed fp32 FMA / add: two scores per instruction
                    fa_f32x2 t2 = fa_f32x2{s[sub][r], s[sub][r + 1]} * c22 + nm2;
                    t2[0] = __builtin_amdgcn_exp2f(t2[0]);
                    t2[1] = __builtin_amdgcn_exp2f(t2[1]);
                    s[sub][r] = t2[0]; s[sub][r + 1] = t2[1];
                    rs2 += t2;
                }
            float rsum = rs2[0] + rs2[1];
            rsum += fa_other_half(rsum);
            l = l * alpha + rsum;
            m = m_new;
            if (__builtin_amdgcn_ballot_w64(alpha != 1.0f) != 0) {   // the running maximum rarely moves after the first tiles
#pragma unroll
                for (int i = 0; i < NMI; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) oacc[i][r] *= alpha;
            }
#pragma unroll
            for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    bf16x8 pf;
#pragma unroll
                    for (int j = 0; j < 8; ++j) pf[j] = (T)s[sub][8 * t + j];
#pragma unroll
                    for (int i = 0; i < NMI; ++i) {
                        // lane s of a 16-lane group addresses key row base + (s>>2), head dims 4(s&3)..+3 of this 16-dim block
                        const unsigned char* a0 = vt + (sub * 32 + 16 * t + 4 * g + (sl >> 2)) * RSV + (i * 32 + 16 * G16 + 4 * (sl & 3)) * 2;
                        mma16(oacc[i], fa_tr_frag(a0, a0 + 8 * RSV), pf);
                    }
                }
        }
        if (more) commit(fa_smem + (cur ^ 1) * G::STAGE);
        __syncthreads();
        cur ^= 1;
    }

    if (query < p.S) {                               // lane = query, accumulator rows = head dims (4 consecutive per register quad)
        const float inv = 1.0f / l;
        T* dst = reinterpret_cast<T*>(p.o) + ((size_t)b * p.S + query) * ((size_t)p.H * HD) + (size_t)h * HD;
#pragma unroll
        for (int i = 0; i < NMI; ++i)
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                bf16x4 o4;
#pragma unroll
                for (int e = 0; e < 4; ++e) o4[e] = (T)(oacc[i][4 * rq + e] * inv);
                *reinterpret_cast<bf16x4*>(dst + i * 32 + 8 * rq + 4 * g) = o4;
            }
        if (p.lse && g == 0) p.lse[((size_t)b * p.H + h) * p.S + query] = m * 0.6931471805599453f + __logf(l);
    }
}

// ---------------------------------------------------------------------------------------------------------
// v2 of the head_dim-64 bf16 forward (round 3).  Same arithmetic, same products, same fragment layouts as the kernel above; what
// changes is how many waves a SIMD holds.  v1: 132 VGPRs (16 of them the next tile's K / V rows in flight) and 43 KB of LDS per
// work-group (row padding) = 3 waves per SIMD, and the SIMDs idle 40 % of the cycles (MFMA busy 24 %, other VALU ~36 %: nothing to
// switch to while a wave waits for its score MFMAs, its LDS reads or the barrier).  v2:
//   * K / V tiles go global -> LDS by DMA (`buffer_load_dwordx4 ... lds`, inline assembly as in conv_wgrad_dma.hip so that the
//     compiler does not put a vmcnt(0) in front of the next LDS read): no staging registers, no ds_write;
//   * unpadded 128-byte rows, XOR swizzles instead of padding (K: 16-byte slot ^ ((key >> 1) & 7), the ds_read_b128 pattern of the
//     conv kernels; V: 64-byte block ^ ((key >> 1) & 1), the transpose-read pattern of conv_wgrad_dma.hip): 32 KB per work-group;
//   * __launch_bounds__(256, 4): <= 128 VGPRs, 4 work-groups per CU.
// The key loop is unrolled by two so that the stage offset is an immediate of every LDS read.
typedef __attribute__((ext_vector_type(4))) int fa_i32x4;
__device__ __forceinline__ void fa_dma16(fa_i32x4 rs, unsigned lds, int vo) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" :: "s"(lds), "v"(vo), "s"(rs) : "memory", "m0");
}
__device__ __forceinline__ fa_i32x4 fa_rsrc(const void* base, unsigned bytes) {
    const unsigned long long a = (unsigned long long)base;
    fa_i32x4 r = {(int)(unsigned)a, (int)(unsigned)(a >> 32), (int)bytes, 0x00020000};
    r[0] = __builtin_amdgcn_readfirstlane(r[0]); r[1] = __builtin_amdgcn_readfirstlane(r[1]);
    r[2] = __builtin_amdgcn_readfirstlane(r[2]); r[3] = __builtin_amdgcn_readfirstlane(r[3]);
    return r;
}
constexpr int F2_TILE = 64 * 128;                 // one 64-key x 64-dim bf16 tile
constexpr int F2_STAGE = 2 * F2_TILE;             // K then V
constexpr int F2_LDS = 2 * F2_STAGE;              // 32 KB
constexpr int F2_OOB = (int)0x80000000;

// -DFA_TRACE (tools/build_file_variant.sh, never the shipped build): lane 0 of wave 3 of the heaviest work-group of (batch 0, head 0) sums
// the 100 MHz wall clock over the phases of its key tiles (each phase closed by a read of its last result, so MFMA / LDS / VALU
// latencies are inside the phase that caused them; every stamp is an s_memrealtime round trip of ~80 ns that lands in the phase it
// closes); tools/kbench.py attn prints them.
#ifdef FA_TRACE
__device__ long long g_fa_trace[32];
#define FA_T(i) do { if (fa_tr) { const long long now_ = wall_clock64(); fa_acc[i] += now_ - fa_last; fa_last = now_; } } while (0)
#define FA_FORCE(x) do { if (fa_tr) asm volatile("v_readfirstlane_b32 %0, %1" : "=s"(fa_sink) : "v"(x)); } while (0)
#else
#define FA_T(i) do { } while (0)
#define FA_FORCE(x) do { } while (0)
#endif
__global__ __launch_bounds__(NT, 4) void attn_causal_fwd_bf16_v2_kernel(AttnParams p) {
    using T = bf16_t;
#ifdef FA_TRACE
    const bool fa_tr = blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 192;
    long long fa_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const long long fa_t0 = wall_clock64();
    long long fa_last = fa_t0;
    int fa_sink = 0;
#endif
    constexpr int HD = 64, KT2 = 64, NKK = 4, NMI = 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char fa_smem[];
    const unsigned lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) unsigned char*)fa_smem;

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 5, l31 = lane & 31, G16 = (lane >> 4) & 1, sl = lane & 15;
    const int bh = p.lpt ? blockIdx.x : blockIdx.y, b = bh / p.H, h = bh % p.H;
    const int q0 = p.lpt ? ((int)gridDim.y - 1 - (int)blockIdx.y) * QT : ((int)gridDim.x - 1 - (int)blockIdx.x) * QT;     // heavy blocks first
    const int qw = q0 + wave * 32, query = qw + l31;

    const T* __restrict__ Q = reinterpret_cast<const T*>(p.q) + (size_t)b * p.q_bs + (size_t)h * HD;
    const T* __restrict__ K = reinterpret_cast<const T*>(p.k) + (size_t)b * p.k_bs + (size_t)h * HD;
    const T* __restrict__ V = reinterpret_cast<const T*>(p.v) + (size_t)b * p.v_bs + (size_t)h * HD;
    // one descriptor per tensor over this (batch, head)'s rows: a key >= S lands past the end and reads as zeros
    const fa_i32x4 rs_k = fa_rsrc(K, (unsigned)((size_t)(p.S - 1) * p.ld_k * 2 + HD * 2));
    const fa_i32x4 rs_v = fa_rsrc(V, (unsigned)((size_t)(p.S - 1) * p.ld_v * 2 + HD * 2));

    bf16x8 qf[NKK];                                  // Q^T fragments (B operand): lane = query, 8 consecutive head dims
#pragma unroll
    for (int kk = 0; kk < NKK; ++kk) {
        u32x4 raw = u32x4{0u, 0u, 0u, 0u};
        if (query < p.S) raw = *reinterpret_cast<const u32x4*>(Q + (size_t)query * p.ld_q + kk * 16 + g * 8);
        qf[kk] = *reinterpret_cast<const bf16x8*>(&raw);
    }
    f32x16 oacc[NMI];
#pragma unroll
    for (int i = 0; i < NMI; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[i][r] = 0.0f;
    float m = -1e30f, l = 0.0f;                      // running max (log2 domain) and sum per query
    const float c2 = p.scale * 1.4426950408889634f;

    // ---- DMA plan: a tile is 8 pieces of 8 keys x 128 B per tensor; wave w moves pieces w and w + 4 of K and of V.  The LDS image
    //      of a piece is lane-linear (key 8 piece + (lane >> 3), physical slot lane & 7): the swizzle goes on the SOURCE slot
    int vk[2], vv[2];                                // lane parts of the source offsets (relative to the tile's first key)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int key_l = (wave + 4 * j) * 8 + (lane >> 3), ps = lane & 7;
        vk[j] = key_l * p.ld_k * 2 + ((ps ^ ((key_l >> 1) & 7)) << 4);
        vv[j] = key_l * p.ld_v * 2 + (((((ps >> 2) ^ ((key_l >> 1) & 1)) << 2) | (ps & 3)) << 4);
    }
    auto issue = [&](int k0, int stage) {
        const int uk = k0 * p.ld_k * 2, uv = k0 * p.ld_v * 2;           // uniform
        const bool full = k0 + KT2 <= p.S;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int key = k0 + (wave + 4 * j) * 8 + (lane >> 3);
            const bool ok = full || key < p.S;
            const unsigned dst = lds0 + stage * F2_STAGE + (wave + 4 * j) * 1024;
            fa_dma16(rs_k, __builtin_amdgcn_readfirstlane(dst), ok ? vk[j] + uk : F2_OOB);
            fa_dma16(rs_v, __builtin_amdgcn_readfirstlane(dst + F2_TILE), ok ? vv[j] + uv : F2_OOB);
        }
    };
    // ---- fragment addresses (lane parts; the stage, sub-tile and k-step offsets are immediates)
    const int key7 = (l31 >> 1) & 7;
    int ka[NKK];                                     // K: row l31, logical slot 2 kk + g -> physical slot ^ key7
#pragma unroll
    for (int kk = 0; kk < NKK; ++kk) ka[kk] = l31 * 128 + (((2 * kk + g) ^ key7) << 4);
    int va[NMI];                                     // V^T: key row 4 g + (sl >> 2), dims 32 i + 16 G16 + 4 (sl & 3): block i ^ ((sl >> 3) & 1)
#pragma unroll
    for (int i = 0; i < NMI; ++i) va[i] = (4 * g + (sl >> 2)) * 128 + ((i ^ ((sl >> 3) & 1)) << 6) + 32 * G16 + 8 * (sl & 3);

    const int q_last = min(q0 + QT, p.S) - 1;        // keys beyond the work-group's last query are never needed
    const int n_tiles = q_last / KT2 + 1;
    issue(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    FA_T(0);                                         // prologue: Q loads + the first tile's DMA
    // the Q fragments are USED here, before the loop: the compiler then places its own wait for their global loads here and not in
    // front of the first MFMA of every tile -- where a vmcnt(0) also waits for the DMA of the NEXT tile that was issued a few
    // instructions earlier (the counter cannot tell the two apart), i.e. exposes the whole prefetch latency once per tile
    // (round 5: the forward without the per-tile DMA wait, profiles/r05_attn_ablation.txt)
#pragma unroll
    for (int kk = 0; kk < NKK; ++kk) asm volatile("" ::"v"(qf[kk]));
    __syncthreads();

    auto tile = [&](int it, auto stage_c) {
        constexpr int ST = decltype(stage_c)::value;
        const int k0 = it * KT2;
        const bool more = it + 1 < n_tiles;
#ifndef FA_ABL_NODMA
        if (more) issue(k0 + KT2, ST ^ 1);
#else
        (void)more;
#endif
        FA_T(1);                                     // DMA issue
        if (k0 <= qw + 31) {                         // wave-uniform: otherwise the tile is entirely above this wave's diagonal
            const unsigned char* kt = fa_smem + ST * F2_STAGE;
            const unsigned char* vt = kt + F2_TILE;
            f32x16 s[2];
#ifdef FA_ABL_NOQK
#pragma unroll
            for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                for (int r = 0; r < 16; ++r) s[sub][r] = (float)it;
            (void)kt;
#else
#pragma unroll
            for (int sub = 0; sub < 2; ++sub) {
#pragma unroll
                for (int r = 0; r < 16; ++r) s[sub][r] = 0.0f;
#pragma unroll
                for (int kk = 0; kk < NKK; ++kk) {
                    const bf16x8 kf = *reinterpret_cast<const bf16x8*>(kt + ka[kk] + sub * 4096);
                    mma16(s[sub], kf, qf[kk]);
                }
            }
#endif
            FA_FORCE(s[1][15]);
            FA_T(2);                                                // K fragment reads + score MFMAs
            if ((k0 + KT2 - 1 > qw) || (k0 + KT2 > p.S)) {          // tile touches the diagonal or the sequence end: mask
#pragma unroll
                for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int key = k0 + sub * 32 + acc_row(lane, r);
                        if (!(key <= query && key < p.S)) s[sub][r] = -1e30f;
                    }
            }
#ifdef FA_ABL_NOSOFTMAX                    // (ablation builds, tools/experiments/gpu_r5_12.sh: where the forward's time goes)
            const float alpha = 1.0f;
            l += s[0][0];
#else
            float tmax = -1e30f;
#pragma unroll
            for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                for (int r = 0; r < 16; ++r) tmax = fmaxf(tmax, s[sub][r]);
            tmax = fmaxf(tmax, fa_other_half(tmax)) * c2;
            const float m_new = fmaxf(m, tmax);
            const float alpha = __builtin_amdgcn_exp2f(m - m_new);
#ifdef FA_ABL_NOEXP
#define FA_EXP2(x) (x)
#else
#define FA_EXP2(x) __builtin_amdgcn_exp2f(x)
#endif
#ifdef FA_SCALAR_SOFTMAX   // one v_fma / v_exp / v_add per score (build with -fno-slp-vectorize): packed fp32 VALU beside MFMAs is priced as an
            float rs0 = 0.0f, rs1 = 0.0f;                           // anti-lever in MI355X_MICROARCH.md (2 v_pk_add +26 cycles vs 2 v_fma per gap)
            const float nm = -m_new;
#pragma unroll
            for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    const float t0 = FA_EXP2(__builtin_fmaf(s[sub][r], c2, nm));
                    const float t1 = FA_EXP2(__builtin_fmaf(s[sub][r + 1], c2, nm));
                    s[sub][r] = t0; s[sub][r + 1] = t1;
                    rs0 += t0; rs1 += t1;
                }
            float rsum = rs0 + rs1;
#else
            fa_f32x2 rs2 = fa_f32x2{0.0f, 0.0f};
            const fa_f32x2 c22 = fa_f32x2{c2, c2}, nm2 = fa_f32x2{-m_new, -m_new};
#pragma unroll
            for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    fa_f32x2 t2 = fa_f32x2{s[sub][r], s[sub][r + 1]} * c22 + nm2;
                    t2[0] = __builtin_amdgcn_exp2f(t2[0]);
                    t2[1] = __builtin_amdgcn_exp2f(t2[1]);
                    s[sub][r] = t2[0]; s[sub][r + 1] = t2[1];
                    rs2 += t2;
                }
            float rsum = rs2[0] + rs2[1];
#endif
            rsum += fa_other_half(rsum);
            l = l * alpha + rsum;
            m = m_new;
#endif
            if (__builtin_amdgcn_ballot_w64(alpha != 1.0f) != 0) {
#pragma unroll
                for (int i = 0; i < NMI; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) oacc[i][r] *= alpha;
            }
            FA_FORCE(l);
            FA_FORCE(s[1][15]);
            FA_T(3);                                                // mask + softmax (+ rescale)
#pragma unroll
            for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    bf16x8 pf;
#pragma unroll
                    for (int j = 0; j < 8; ++j) pf[j] = (T)s[sub][8 * t + j];
#ifdef FA_ABL_NOPV
                    asm volatile("" ::"v"(pf));
                    (void)vt;
#else
#pragma unroll
                    for (int i = 0; i < NMI; ++i) {
                        const unsigned char* a0 = vt + va[i] + (sub * 32 + 16 * t) * 128;
                        mma16(oacc[i], fa_tr_frag(a0, a0 + 8 * 128), pf);
                    }
#endif
                }
        }
        FA_FORCE(oacc[1][15]);
        FA_T(4);                                                     // converts + V^T fragment reads + P V MFMAs
#ifndef FA_ABL_NOBAR
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // this wave's pieces of the next tile have landed
        FA_T(5);                                                     // wait for the next tile's DMA
        __syncthreads();
        FA_T(6);                                                     // work-group barrier
#endif
    };
    for (int it = 0; it < n_tiles; it += 2) {
        tile(it, std::integral_constant<int, 0>{});
        if (it + 1 < n_tiles) tile(it + 1, std::integral_constant<int, 1>{});
    }

    if (query < p.S) {                               // lane = query, accumulator rows = head dims (4 consecutive per register quad)
        const float inv = 1.0f / l;
        T* dst = reinterpret_cast<T*>(p.o) + ((size_t)b * p.S + query) * ((size_t)p.H * HD) + (size_t)h * HD;
#pragma unroll
        for (int i = 0; i < NMI; ++i)
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                bf16x4 o4;
#pragma unroll
                for (int e = 0; e < 4; ++e) o4[e] = (T)(oacc[i][4 * rq + e] * inv);
                *reinterpret_cast<bf16x4*>(dst + i * 32 + 8 * rq + 4 * g) = o4;
            }
        if (p.lse && g == 0) p.lse[((size_t)b * p.H + h) * p.S + query] = m * 0.6931471805599453f + __logf(l);
    }
#ifdef FA_TRACE
    FA_T(7);                                         // epilogue
    if (fa_tr) {
#pragma unroll
        for (int i = 0; i < 8; ++i) g_fa_trace[i] = fa_acc[i];
        g_fa_trace[8] = n_tiles; g_fa_trace[9] = fa_t0; g_fa_trace[10] = fa_last;
    }
    if (blockIdx.x == 0 && blockIdx.y == gridDim.y - 1 && threadIdx.x == 0) { g_fa_trace[11] = fa_t0; g_fa_trace[12] = wall_clock64(); }   // the lightest block of the same head
    (void)fa_sink;
#endif
}

#ifdef FA_TRACE
}  // namespace
extern "C" int mas_fa_trace(long long* out32) { return (int)hipMemcpyFromSymbol(out32, HIP_SYMBOL(g_fa_trace), sizeof(long long) * 32); }
namespace {
#endif
int launch_fwd_fast_v2(const AttnParams& p, hipStream_t s) {
    AttnParams q = p;
    q.lpt = fa_lpt();
    const dim3 grid = q.lpt ? dim3(p.B * p.H, mas_cdiv(p.S, QT)) : dim3(mas_cdiv(p.S, QT), p.B * p.H);
    hipLaunchKernelGGL(attn_causal_fwd_bf16_v2_kernel, grid, dim3(NT), F2_LDS, s, q);
    MAS_CHECK_LAUNCH("attn_causal_fwd_v2");
    return MAS_OK;
}

template <int HD>
int launch_fwd_fast(const AttnParams& p, hipStream_t s) {
    using G = FaGeo<HD>;
    auto kern = attn_causal_fwd_bf16_kernel<HD>;
    static mas_devmask_t attr_mask{0};
    unsigned long long attr_bit;
    if (mas_attr_needed(attr_mask, &attr_bit)) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::LDS_BYTES) != hipSuccess)
            MAS_FAIL(MAS_ELAUNCH, "attn_causal_fwd: cannot set dynamic LDS size %zu", (size_t)G::LDS_BYTES);
        mas_attr_done(attr_mask, attr_bit);
    }
    hipLaunchKernelGGL(kern, dim3(mas_cdiv(p.S, QT), p.B * p.H), dim3(NT), G::LDS_BYTES, s, p);
    MAS_CHECK_LAUNCH("attn_causal_fwd");
    return MAS_OK;
}

inline bool attn_generic() { return false; }
inline bool fa_aligned(const void* ptr, long long bs, int ld) {
    return (reinterpret_cast<uintptr_t>(ptr) & 15) == 0 && (bs % 8) == 0 && (ld % 8) == 0;
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
    int lpt;                        // (see AttnParams)
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

// ---------------------------------------------------------------------------------------------------------
// bf16 / head_dim 64 fast path of the backward: the same two kernels (dK,dV per key block; dQ per query block, every
// output element written exactly once, no atomics) restructured like the fast forward:
//   * 64-row tiles of the streamed operands (Q,dO for dK/dV; K,V for dQ), two LDS stages, 16-byte coalesced global
//     loads in flight in registers across the compute phase, one barrier per tile;
//   * ONE natural-layout [row][64] image per operand serves both uses: row fragments (ds_read_b128, the A operand of
//     S = Q K^T / dP = dO V^T) and transposed fragments (ds_read_b64_tr_b16, the A operand of dV^T += dO^T P,
//     dK^T += Q^T dS, dQ^T += K^T dS^T).  Rows are 128 B with the 16-byte slot XOR-swizzled by the BIT-REVERSED
//     (row>>1)&7: 16 rows of a ds_read_b128 group still hit 16 distinct slots, and the 4 consecutive rows of a
//     transpose read fall in 4 distinct 64-byte bank segments -- both conflict-free without padding;
//   * exp2 with lse*log2(e) staged per row, masking only on tiles that touch the diagonal / the sequence end.
__device__ __forceinline__ int fa_swz(int row) { return (((row >> 1) & 1) << 2) | (((row >> 2) & 1) << 1) | ((row >> 3) & 1); }
__device__ __forceinline__ int fa_phys(int row, int slot) { return row * 128 + ((slot ^ fa_swz(row)) << 4); }

struct FaTile {                       // one operand tile: 64 rows x 64 bf16, swizzled
    static constexpr int BYTES = 64 * 128;
};

// 16-byte slots of a 64 x 64 tile: thread tid owns (row = u >> 3, slot = u & 7) for u = tid, tid + 256
__device__ __forceinline__ void fa_issue_tile(const bf16_t* __restrict__ src, long long ld, int row0, int S, int tid, u32x4 (&r)[2]) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int u = tid + i * NT, row = row0 + (u >> 3), slot = u & 7;
        const bool ok = row < S;
        r[i] = *reinterpret_cast<const u32x4*>(src + (size_t)(ok ? row : 0) * ld + slot * 8);
        if (!ok) r[i] = u32x4{0u, 0u, 0u, 0u};
    }
}
__device__ __forceinline__ void fa_commit_tile(unsigned char* tile, int tid, const u32x4 (&r)[2]) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int u = tid + i * NT;
        *reinterpret_cast<u32x4*>(tile + fa_phys(u >> 3, u & 7)) = r[i];
    }
}
// A operand, rows = tile rows: lane (l31 = row, g) holds 8 consecutive head dims of k-step kk
__device__ __forceinline__ bf16x8 fa_row_frag(const unsigned char* tile, int row, int kk, int g) {
    return *reinterpret_cast<const bf16x8*>(tile + fa_phys(row, kk * 2 + g));
}
// A operand, rows = head dims i*32 + l31, k = tile rows rb + {4g + 0..3, 8 + 4g + 0..3} (the accumulator-register order)
__device__ __forceinline__ bf16x8 fa_tr_tile_frag(const unsigned char* tile, int rb, int i, int g, int G16, int sl) {
    const int slot = i * 4 + G16 * 2 + ((sl & 3) >> 1), within = (sl & 1) * 8;
    const int r0 = rb + 4 * g + (sl >> 2);
    return fa_tr_frag(tile + fa_phys(r0, slot) + within, tile + fa_phys(r0 + 8, slot) + within);
}

// LDS-DMA staging of a swizzled 64 x 64 tile (round 3, as in the v2 forward): 8 pieces of 8 rows; wave w moves pieces w and w + 4.
// The LDS image of a piece is lane-linear (row 8 piece + (lane >> 3), physical slot lane & 7), so the swizzle goes on the SOURCE slot.
__device__ __forceinline__ void fa_dma_plan(int wave, int lane, int ld, int (&vo)[2]) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int row_l = (wave + 4 * j) * 8 + (lane >> 3);
        vo[j] = row_l * ld * 2 + (((lane & 7) ^ fa_swz(row_l)) << 4);
    }
}
__device__ __forceinline__ void fa_dma_tile(fa_i32x4 rs, unsigned lds_tile, int row0, int S, int ld, int wave, int lane, const int (&vo)[2]) {
    const int u0 = row0 * ld * 2;
    const bool full = row0 + 64 <= S;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const bool ok = full || (row0 + (wave + 4 * j) * 8 + (lane >> 3) < S);
        fa_dma16(rs, __builtin_amdgcn_readfirstlane(lds_tile + (wave + 4 * j) * 1024), ok ? vo[j] + u0 : F2_OOB);
    }
}

#ifndef FA_DKV_WGS
#define FA_DKV_WGS 2
#endif
__global__ __launch_bounds__(NT, FA_DKV_WGS) void attn_bwd_dkv_bf16_kernel(AttnBwdParams p) {
    using T = bf16_t;
    constexpr int HD = 64, NKK = 4, NMI = 2, QT2 = 64;
    constexpr int STAGE = 2 * FaTile::BYTES + 2 * QT2 * (int)sizeof(float);
    extern __shared__ __attribute__((aligned(16))) unsigned char fa_smem[];
    const unsigned lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) unsigned char*)fa_smem;

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 5, l31 = lane & 31, G16 = (lane >> 4) & 1, sl = lane & 15;
    const int bh = p.lpt ? blockIdx.x : blockIdx.y, b = bh / p.H, h = bh % p.H;
    const int k0 = (p.lpt ? blockIdx.y : blockIdx.x) * QT;          // 128 keys per work-group; block 0 is the heaviest (all queries)
    const int kw = k0 + wave * 32, key = kw + l31;
    const size_t head = (size_t)b * p.bs + (size_t)h * HD;
    const T* __restrict__ Q = reinterpret_cast<const T*>(p.q) + head;
    const T* __restrict__ K = reinterpret_cast<const T*>(p.k) + head;
    const T* __restrict__ V = reinterpret_cast<const T*>(p.v) + head;
    const long long g_stride = (long long)p.H * HD;
    const T* __restrict__ G = reinterpret_cast<const T*>(p.dout) + ((size_t)b * p.S) * (size_t)g_stride + (size_t)h * HD;
    const float* __restrict__ LSE = p.lse + ((size_t)b * p.H + h) * p.S;
    const float* __restrict__ DEL = p.delta + ((size_t)b * p.H + h) * p.S;

    bf16x8 kf[NKK], vf[NKK];                        // B operands: lane = key column, 8 consecutive head dims
#pragma unroll
    for (int kk = 0; kk < NKK; ++kk) {
        u32x4 a = u32x4{0u, 0u, 0u, 0u}, c = a;
        if (key < p.S) {
            a = *reinterpret_cast<const u32x4*>(K + (size_t)key * p.ld + kk * 16 + g * 8);
            c = *reinterpret_cast<const u32x4*>(V + (size_t)key * p.ld + kk * 16 + g * 8);
        }
        kf[kk] = *reinterpret_cast<const bf16x8*>(&a);
        vf[kk] = *reinterpret_cast<const bf16x8*>(&c);
    }
    f32x16 dk[NMI], dv[NMI];
#pragma unroll
    for (int i = 0; i < NMI; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) { dk[i][r] = 0.0f; dv[i][r] = 0.0f; }
    const float c2 = p.scale * 1.4426950408889634f;

    // Q / dO tiles: global -> LDS by DMA (round 3; no staging registers); the lse / delta rows of the tile ride through two registers
    const fa_i32x4 rs_q = fa_rsrc(Q, (unsigned)((size_t)(p.S - 1) * p.ld * 2 + HD * 2));
    const fa_i32x4 rs_g = fa_rsrc(G, (unsigned)((size_t)(p.S - 1) * g_stride * 2 + HD * 2));
    int voq[2], vog[2];
    fa_dma_plan(wave, lane, p.ld, voq);
    fa_dma_plan(wave, lane, (int)g_stride, vog);
    float rl = 0.0f, rd = 0.0f;
    auto issue = [&](int q0, int stage) {
        fa_dma_tile(rs_q, lds0 + stage * STAGE, q0, p.S, p.ld, wave, lane, voq);
        fa_dma_tile(rs_g, lds0 + stage * STAGE + FaTile::BYTES, q0, p.S, (int)g_stride, wave, lane, vog);
        if (tid < QT2) {
            const bool ok = q0 + tid < p.S;
            rl = ok ? LSE[q0 + tid] * 1.4426950408889634f : 0.0f;
            rd = ok ? DEL[q0 + tid] : 0.0f;
        }
    };
    auto commit = [&](unsigned char* st) {
        if (tid < QT2) {
            reinterpret_cast<float*>(st + 2 * FaTile::BYTES)[tid] = rl;
            reinterpret_cast<float*>(st + 2 * FaTile::BYTES)[QT2 + tid] = rd;
        }
    };

    const int q_start = (k0 / QT2) * QT2;           // causal: queries before the work-group's first key never see it
    const int n_tiles = (p.S - q_start + QT2 - 1) / QT2;
    issue(q_start, 0);
    commit(fa_smem);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int cur = 0;
    for (int it = 0; it < n_tiles; ++it) {
        const int q0 = q_start + it * QT2;
        const bool more = it + 1 < n_tiles;
#ifndef FB_ABL_NODMA                                 // (ablation builds of the BACKWARD, round 6: profiles/r06_attn_bwd_ablation.txt; -DFB_ABL_* change results)
        if (more) issue(q0 + QT2, cur ^ 1);         // the other stage was last read before the barrier that ended the previous tile
#endif
        const unsigned char* qt = fa_smem + cur * STAGE;
        const unsigned char* gt = qt + FaTile::BYTES;
        const float* lse_s = reinterpret_cast<const float*>(qt + 2 * FaTile::BYTES);
        const float* del_s = lse_s + QT2;
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            const int qs = q0 + sub * 32;           // first query of this 32-row block
            if (qs + 31 < kw) continue;             // wave-uniform: every query of the block precedes this wave's keys
            f32x16 s, dp;
#pragma unroll
            for (int kk = 0; kk < NKK; ++kk) {      // rows = queries (A from LDS), columns = keys (B in registers)
                const bf16x8 qa = fa_row_frag(qt, sub * 32 + l31, kk, g);
                const bf16x8 ga = fa_row_frag(gt, sub * 32 + l31, kk, g);
                if (kk == 0) {
                    f32x16 z;
#pragma unroll
                    for (int r = 0; r < 16; ++r) z[r] = 0.0f;
                    s = z; dp = z;
                }
#ifndef FB_ABL_NOSDP
                mma16(s, qa, kf[kk]);
                mma16(dp, ga, vf[kk]);
#else
                s[kk] += (float)qa[0]; dp[kk] += (float)ga[0];
#endif
            }
            f32x4 l4[4], d4[4];                     // rows r = 4j..4j+3 are queries 8j + 4g + 0..3 of the block
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                l4[j] = *reinterpret_cast<const f32x4*>(lse_s + sub * 32 + 8 * j + 4 * g);
                d4[j] = *reinterpret_cast<const f32x4*>(del_s + sub * 32 + 8 * j + 4 * g);
            }
            const bool needs_mask = (qs < kw + 31) || (qs + 32 > p.S) || (kw + 32 > p.S);
            // (the mask as ONE wave-uniform branch around 16 selects, not a branch per element: with the test inside the element loop the
            //  compiler cut this block into 33 basic blocks per 32 MFMAs and nothing could be scheduled across them)
#ifndef FB_ABL_NOSOFTMAX
#pragma unroll
#ifndef FB_ABL_NOEXP
            for (int r = 0; r < 16; ++r) s[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(s[r], c2, -l4[r >> 2][r & 3]));      // P[query][key]
#else
            for (int r = 0; r < 16; ++r) s[r] = __builtin_fmaf(s[r], c2, -l4[r >> 2][r & 3]);
#endif
            if (needs_mask) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int query = qs + acc_row(lane, r);
                    s[r] = ((key <= query) && (query < p.S) && (key < p.S)) ? s[r] : 0.0f;
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) dp[r] = s[r] * (dp[r] - d4[r >> 2][r & 3]) * p.scale;                            // dS[query][key]
#else
            if (needs_mask && c2 == 12345.0f) s[0] = l4[0][0] + d4[0][0];
#endif
#ifndef FB_ABL_NOOUT
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                bf16x8 pf, sf;
#pragma unroll
                for (int j = 0; j < 8; ++j) { pf[j] = (T)s[8 * t + j]; sf[j] = (T)dp[8 * t + j]; }
#pragma unroll
                for (int i = 0; i < NMI; ++i) {
                    mma16(dv[i], fa_tr_tile_frag(gt, sub * 32 + 16 * t, i, g, G16, sl), pf);    // dV^T += dO^T P
                    mma16(dk[i], fa_tr_tile_frag(qt, sub * 32 + 16 * t, i, g, G16, sl), sf);    // dK^T += Q^T dS
                }
            }
#else
#pragma unroll
            for (int r = 0; r < 16; ++r) { dv[0][r] += s[r]; dk[0][r] += dp[r]; }
#endif
        }
        if (more) commit(fa_smem + (cur ^ 1) * STAGE);
#ifndef FB_ABL_NOWAIT                                // (timing only: is the DMA's LATENCY exposed, or only its issue?)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // this wave's pieces of the next tile have landed
#endif
        __syncthreads();
        cur ^= 1;
    }
    if (key < p.S) {                                 // lane = key, accumulator rows = head dims (4 consecutive per register quad)
        T* dkp = reinterpret_cast<T*>(p.dk) + head + (size_t)key * p.ld;
        T* dvp = reinterpret_cast<T*>(p.dv) + head + (size_t)key * p.ld;
#pragma unroll
        for (int i = 0; i < NMI; ++i)
#pragma unroll
            for (int rq4 = 0; rq4 < 4; ++rq4) {
                bf16x4 a, c;
#pragma unroll
                for (int e = 0; e < 4; ++e) { a[e] = (T)dk[i][4 * rq4 + e]; c[e] = (T)dv[i][4 * rq4 + e]; }
                *reinterpret_cast<bf16x4*>(dkp + i * 32 + 8 * rq4 + 4 * g) = a;
                *reinterpret_cast<bf16x4*>(dvp + i * 32 + 8 * rq4 + 4 * g) = c;
            }
    }
}

#ifndef FA_DQ_WGS
#define FA_DQ_WGS 3
#endif
__global__ __launch_bounds__(NT, FA_DQ_WGS) void attn_bwd_dq_bf16_kernel(AttnBwdParams p) {       // (round 3: DMA staging -> <= 168 VGPRs, 3 waves per SIMD)
    using T = bf16_t;
    constexpr int HD = 64, NKK = 4, NMI = 2, KT2 = 64;
    constexpr int STAGE = 2 * FaTile::BYTES;
    extern __shared__ __attribute__((aligned(16))) unsigned char fa_smem[];
    const unsigned lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) unsigned char*)fa_smem;

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 5, l31 = lane & 31, G16 = (lane >> 4) & 1, sl = lane & 15;
    const int bh = p.lpt ? blockIdx.x : blockIdx.y, b = bh / p.H, h = bh % p.H;
    const int q0 = p.lpt ? ((int)gridDim.y - 1 - (int)blockIdx.y) * QT : ((int)gridDim.x - 1 - (int)blockIdx.x) * QT;     // heavy (late) query blocks first
    const int qw = q0 + wave * 32, query = qw + l31;
    const size_t head = (size_t)b * p.bs + (size_t)h * HD;
    const T* __restrict__ Q = reinterpret_cast<const T*>(p.q) + head;
    const T* __restrict__ K = reinterpret_cast<const T*>(p.k) + head;
    const T* __restrict__ V = reinterpret_cast<const T*>(p.v) + head;
    const long long g_stride = (long long)p.H * HD;
    const T* __restrict__ G = reinterpret_cast<const T*>(p.dout) + ((size_t)b * p.S) * (size_t)g_stride + (size_t)h * HD;

    bf16x8 qf[NKK], gf[NKK];                        // B operands: lane = query column
#pragma unroll
    for (int kk = 0; kk < NKK; ++kk) {
        u32x4 a = u32x4{0u, 0u, 0u, 0u}, c = a;
        if (query < p.S) {
            a = *reinterpret_cast<const u32x4*>(Q + (size_t)query * p.ld + kk * 16 + g * 8);
            c = *reinterpret_cast<const u32x4*>(G + (size_t)query * g_stride + kk * 16 + g * 8);
        }
        qf[kk] = *reinterpret_cast<const bf16x8*>(&a);
        gf[kk] = *reinterpret_cast<const bf16x8*>(&c);
    }
    const float my_lse2 = query < p.S ? p.lse[((size_t)b * p.H + h) * p.S + query] * 1.4426950408889634f : 0.0f;
    const float my_delta = query < p.S ? p.delta[((size_t)b * p.H + h) * p.S + query] : 0.0f;
    const float c2 = p.scale * 1.4426950408889634f;
    f32x16 dq[NMI];
#pragma unroll
    for (int i = 0; i < NMI; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) dq[i][r] = 0.0f;

    // K / V tiles: global -> LDS by DMA (no staging registers); a row >= S reads as zeros through the descriptor's bounds check
    const fa_i32x4 rs_k = fa_rsrc(K, (unsigned)((size_t)(p.S - 1) * p.ld * 2 + HD * 2));
    const fa_i32x4 rs_v = fa_rsrc(V, (unsigned)((size_t)(p.S - 1) * p.ld * 2 + HD * 2));
    int vo[2];
    fa_dma_plan(wave, lane, p.ld, vo);
    const int q_last = min(q0 + QT, p.S) - 1;
    const int n_tiles = q_last / KT2 + 1;
    fa_dma_tile(rs_k, lds0, 0, p.S, p.ld, wave, lane, vo);
    fa_dma_tile(rs_v, lds0 + FaTile::BYTES, 0, p.S, p.ld, wave, lane, vo);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int cur = 0;
    for (int it = 0; it < n_tiles; ++it) {
        const int k0 = it * KT2;
        const bool more = it + 1 < n_tiles;
#ifndef FB_ABL_NODMA
        if (more) {                                 // the other stage was last read before the barrier that ended the previous tile
            fa_dma_tile(rs_k, lds0 + (cur ^ 1) * STAGE, k0 + KT2, p.S, p.ld, wave, lane, vo);
            fa_dma_tile(rs_v, lds0 + (cur ^ 1) * STAGE + FaTile::BYTES, k0 + KT2, p.S, p.ld, wave, lane, vo);
        }
#endif
        const unsigned char* kt = fa_smem + cur * STAGE;
        const unsigned char* vt = kt + FaTile::BYTES;
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            const int ks = k0 + sub * 32;
            if (ks > qw + 31) continue;             // wave-uniform: block entirely above this wave's diagonal
            f32x16 s, dp;
#pragma unroll
            for (int kk = 0; kk < NKK; ++kk) {      // rows = keys (A from LDS), columns = queries (B in registers)
                const bf16x8 ka = fa_row_frag(kt, sub * 32 + l31, kk, g);
                const bf16x8 va = fa_row_frag(vt, sub * 32 + l31, kk, g);
                if (kk == 0) {
                    f32x16 z;
#pragma unroll
                    for (int r = 0; r < 16; ++r) z[r] = 0.0f;
                    s = z; dp = z;
                }
#ifndef FB_ABL_NOSDP
                mma16(s, ka, qf[kk]);
                mma16(dp, va, gf[kk]);
#else
                s[kk] += (float)ka[0]; dp[kk] += (float)va[0];
#endif
            }
            const bool needs_mask = (ks + 31 > qw) || (ks + 32 > p.S);
#ifndef FB_ABL_NOSOFTMAX
#pragma unroll
#ifndef FB_ABL_NOEXP
            for (int r = 0; r < 16; ++r) s[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(s[r], c2, -my_lse2));
#else
            for (int r = 0; r < 16; ++r) s[r] = __builtin_fmaf(s[r], c2, -my_lse2);
#endif
            if (needs_mask) {                                        // (one wave-uniform branch around 16 selects: see attn_bwd_dkv)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = ks + acc_row(lane, r);
                    s[r] = ((key <= query) && (key < p.S)) ? s[r] : 0.0f;
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) dp[r] = s[r] * (dp[r] - my_delta) * p.scale;                                     // dS^T[key][query]
#else
            if (needs_mask && c2 == 12345.0f) s[0] = my_lse2 + my_delta;
#endif
#ifndef FB_ABL_NOOUT
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                bf16x8 sf;
#pragma unroll
                for (int j = 0; j < 8; ++j) sf[j] = (T)dp[8 * t + j];
#pragma unroll
                for (int i = 0; i < NMI; ++i) mma16(dq[i], fa_tr_tile_frag(kt, sub * 32 + 16 * t, i, g, G16, sl), sf);   // dQ^T += K^T dS^T
            }
#else
#pragma unroll
            for (int r = 0; r < 16; ++r) { dq[0][r] += s[r]; dq[1][r] += dp[r]; }
#endif
        }
#ifndef FB_ABL_NOWAIT
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // this wave's pieces of the next tile have landed
#endif
        __syncthreads();
        cur ^= 1;
    }
    if (query < p.S) {
        T* dst = reinterpret_cast<T*>(p.dq) + head + (size_t)query * p.ld;
#pragma unroll
        for (int i = 0; i < NMI; ++i)
#pragma unroll
            for (int rq4 = 0; rq4 < 4; ++rq4) {
                bf16x4 a;
#pragma unroll
                for (int e = 0; e < 4; ++e) a[e] = (T)dq[i][4 * rq4 + e];
                *reinterpret_cast<bf16x4*>(dst + i * 32 + 8 * rq4 + 4 * g) = a;
            }
    }
}

int launch_bwd_fast64(const AttnBwdParams& p, hipStream_t s) {
    const long long rows = (long long)p.B * p.H * p.S;
    hipLaunchKernelGGL((attn_bwd_delta_kernel<bf16_t, 64>), dim3((unsigned)((rows + NT - 1) / NT)), dim3(NT), 0, s, p);
    AttnBwdParams q = p;
    q.lpt = fa_lpt();
    const dim3 grid = q.lpt ? dim3(p.B * p.H, mas_cdiv(p.S, QT)) : dim3(mas_cdiv(p.S, QT), p.B * p.H);
    constexpr size_t lds_dkv = 2 * (2 * FaTile::BYTES + 2 * 64 * sizeof(float)), lds_dq = 2 * (2 * FaTile::BYTES);
    hipLaunchKernelGGL(attn_bwd_dkv_bf16_kernel, grid, dim3(NT), lds_dkv, s, q);
    hipLaunchKernelGGL(attn_bwd_dq_bf16_kernel, grid, dim3(NT), lds_dq, s, q);
    MAS_CHECK_LAUNCH("attn_causal_bwd");
    return MAS_OK;
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
    p.B = B; p.H = H; p.S = S; p.scale = scale; p.lpt = 0;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (dtype == MAS_BF16) {
        const bool fast = (hd == 64 || hd == 128) && fa_aligned(q, q_bs, ld_q) && fa_aligned(k, k_bs, ld_k) && fa_aligned(v, v_bs, ld_v) &&
                          (reinterpret_cast<uintptr_t>(o) & 7) == 0 && !attn_generic();
        if (fast) {
            // v2 (DMA staging, swizzled unpadded LDS, 4 work-groups per CU) needs 31-bit byte offsets inside one (batch, head) slab
            const bool small = (long long)S * ld_k * 2 < 0x7fffffffLL && (long long)S * ld_v * 2 < 0x7fffffffLL;
            if (hd == 64 && small) return launch_fwd_fast_v2(p, s);
            return hd == 64 ? launch_fwd_fast<64>(p, s) : launch_fwd_fast<128>(p, s);
        }
        return launch_hd<bf16_t>(p, hd, s);
    }
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
    p.ld = 3 * d; p.bs = (long long)S * 3 * d; p.B = B; p.H = H; p.S = S; p.scale = scale; p.lpt = 0;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (dtype == MAS_BF16) {
        const bool al = ((reinterpret_cast<uintptr_t>(qkv) | reinterpret_cast<uintptr_t>(dqkv) | reinterpret_cast<uintptr_t>(dout)) & 15) == 0 && (d % 8) == 0;
        if (hd == 64 && al && !attn_generic()) return launch_bwd_fast64(p, s);
        return launch_bwd_hd<bf16_t>(p, hd, s);
    }
    if (dtype == MAS_F32) return launch_bwd_hd<float>(p, hd, s);
    MAS_FAIL(MAS_EUNSUPPORTED, "attn_causal_bwd: dtype %d", dtype);
}
