// Single-head spatial self-attention core of AttnBlock for gfx950 (reference models/modules.py:174-187: w = softmax_keys(q^T k * C^-1/2),
// h = v w^T over the h*w tokens of one image), forward and backward, bf16 storage / fp32 accumulate.
// Replaces the two torch.bmm + softmax (and, backward, four bmm + the softmax gradient) of round 1 -- ~60 small library launches
// per VQ-IMG step -- by 1 forward + 2 backward launches per AttnBlock.  S = h*w <= 256 tokens, C <= 512 channels (the reference's
// blocks are 16x16x512 for VQ-IMG, 8x8x512 for VQ-SEG), so one 32-token block's whole score row fits LDS and no online softmax is
// needed.  q, k, v are read in place from the fused [N, S, 3C] projection (q | k | v on the channel axis).
//
// Every kernel is built from two tile products (4 waves; MFMA 32x32x16 bf16):
//   prod : T[32 rows][S]   = X_blk[32][C] . Y_all[S][C]^T     (contraction over channels: both operands channel-contiguous, so
//                             Y rows stream straight from global memory as MFMA A fragments, X_blk rows come from LDS)
//   apply: O[32 rows][C]   = P[32][S] . M_all[S][C]           (contraction over tokens: M tiles are staged in their natural
//                             [token][channel] layout and read with the LDS transpose read ds_read_b64_tr_b16 -- per-lane
//                             addressing as in conv_wgrad.hip -- P rows come from LDS)
//   forward        : T = prod(Q_blk, K); P = softmax(scale T), lse;  O_blk  = apply(P, V)
//   backward, dQ   : P = exp(scale prod(Q_blk, K) - lse); dP = prod(dO_blk, V); delta = rowsum(P dP); dS = scale P (dP - delta);
//                    dQ_blk = apply(dS, K)
//   backward, dK/dV: the same with the roles of queries and keys swapped (block = 32 KEYS, columns = queries, lse / delta indexed
//                    by column): dV_blk = apply(P^T, dO), dK_blk = apply(dS^T, Q).
// No atomics: every output element is written exactly once (deterministic).
#include "mas_common.h"
#include <math.h>

namespace {

constexpr int SNT = 256;
constexpr int S_MAX = 256, C_MAX = 512;

struct SpParams {
    const bf16_t* qkv; const bf16_t* o; const bf16_t* dout; bf16_t* out; bf16_t* dqkv; float* lse; float* delta;
    int N, S, C;
    float scale;
};

typedef __attribute__((ext_vector_type(4))) short sp_s16x4;
__device__ __forceinline__ bf16x8 sp_tr_frag(const unsigned char* a0, const unsigned char* a1) {
    const sp_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) sp_s16x4*)a0);
    const sp_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) sp_s16x4*)a1);
    const __attribute__((ext_vector_type(8))) short v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return *reinterpret_cast<const bf16x8*>(&v);
}

// LDS map (bytes); row strides padded so that 32 consecutive rows do not share banks
struct SpLds {
    int RX;      // X_blk row stride: C*2 + 16
    int ST;      // T row stride: roundup(S,32)*4 + 16
    int SP;      // P / dS row stride: roundup(S,32)*2 + 16
    int RM;      // M tile row stride: C*2 + 64 (the transpose read wants 4 consecutive rows x 64 B to tile a 256-byte bank row)
    int o_x, o_t, o_p, o_p2, o_m, total;
    __host__ __device__ SpLds(int S, int C) {
        const int Sp = (S + 31) & ~31;           // the products write whole 32-column tiles
        RX = C * 2 + 16; ST = Sp * 4 + 16; SP = Sp * 2 + 16; RM = C * 2 + 64;
        o_x = 0; o_t = o_x + 32 * RX; o_p = o_t + 32 * ST; o_p2 = o_p + 32 * SP; o_m = o_p2 + 32 * SP; total = o_m + 32 * RM;
    }
};

// stage 32 rows [r0, r0+32) of a [rows][ld] bf16 matrix (channel window [0, C)) into LDS with row stride RS; rows >= S are zero
__device__ __forceinline__ void sp_stage_rows(unsigned char* dst, int RS, const bf16_t* src, int ld, int r0, int S, int C, int tid) {
    const int upr = C / 8;
    for (int u = tid; u < 32 * upr; u += SNT) {
        const int r = u / upr, cu = u - r * upr;
        u32x4 v = {0u, 0u, 0u, 0u};
        if (r0 + r < S) v = *reinterpret_cast<const u32x4*>(src + (size_t)(r0 + r) * ld + cu * 8);
        *reinterpret_cast<u32x4*>(dst + r * RS + cu * 16) = v;
    }
}

// T[n][m] = sum_c X[n][c] * Y[m][c]  for n in the staged block (32 rows, LDS), m in [0, S) (global rows, stride ld).
// MFMA: A operand (lane = m) straight from global, B operand (lane = n) from LDS; D[m][n] -> T[n][m] (fp32, LDS).
__device__ __forceinline__ void sp_prod(float* T, int STf, const unsigned char* xs, int RX, const bf16_t* Y, int ld, int S, int C,
                                        int wave, int lane) {
    const int g = lane >> 5, l31 = lane & 31;
    const int n_mt = (S + 31) / 32;
    for (int mt = wave; mt < n_mt; mt += 4) {
        const int m = mt * 32 + l31;
        const bf16_t* yr = Y + (size_t)(m < S ? m : 0) * ld + 8 * g;
        const unsigned char* xr = xs + l31 * RX + 16 * g;
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
        // C % 32 == 0: two k-steps per iteration at least; 8 global fragments are put in flight before the first MFMA of a
        // group (one L2 round trip per 8 MFMAs instead of one per MFMA)
        for (int c0 = 0; c0 < C; c0 += 128) {
            bf16x8 a[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                a[u] = zero8<bf16_t>();
                if (c0 + 16 * u < C && m < S) a[u] = *reinterpret_cast<const bf16x8*>(yr + c0 + 16 * u);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (c0 + 16 * u < C) {
                    const bf16x8 b = *reinterpret_cast<const bf16x8*>(xr + (c0 + 16 * u) * 2);
                    mma16(acc, a[u], b);
                }
            }
        }
        // acc[r] = D[m_local = (r&3) + 8(r>>2) + 4g][n = l31]
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 v = {acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]};
            *reinterpret_cast<f32x4*>(T + l31 * STf + mt * 32 + 8 * q + 4 * g) = v;
        }
    }
}

// O[n][c] = sum_s P[n][s] * M[s][c]: P (bf16 [32][S], LDS), M global [S][ld]; result acc tiles: wave w owns channel tiles
// ct = w, w+4, ... (<= 4 of them, C <= 512): acc[i][r] = D[c = ct*32 + (r&3)+8(r>>2)+4g][n = l31].  Needs __syncthreads inside.
__device__ __forceinline__ void sp_apply(f32x16 (&acc)[4], const unsigned char* ps, int SP, unsigned char* ms, int RM, const bf16_t* M, int ld,
                                         int S, int C, int tid) {
    const int lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 5, l31 = lane & 31, G16 = (lane >> 4) & 1, sl = lane & 15;
    const int n_ct = C / 32;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;
    for (int s0 = 0; s0 < S; s0 += 32) {
        __syncthreads();                          // previous tile's transpose reads are done
        sp_stage_rows(ms, RM, M, ld, s0, S, C, tid);
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const bf16x8 b = *reinterpret_cast<const bf16x8*>(ps + l31 * SP + (s0 + ks * 16 + 8 * g) * 2);     // P[n][s0 + 16 ks + 8 g ..+7]
            // transpose-read lane addressing: token (16 ks + 8 g + (sl >> 2)) (+4 for the second read), channels ct*32 + 16*G16 + 4*(sl&3) ..+3
            const unsigned char* a_lane = ms + (ks * 16 + 8 * g + (sl >> 2)) * RM + (16 * G16 + 4 * (sl & 3)) * 2;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int ct = wave + 4 * i;
                if (ct < n_ct) {
                    const bf16x8 a = sp_tr_frag(a_lane + ct * 64, a_lane + ct * 64 + 4 * RM);
                    mma16(acc[i], a, b);
                }
            }
        }
    }
}

// writes acc tiles to out[n][c] (rows r0 + l31 < S), 4 consecutive channels (8 bytes) per accumulator quad
__device__ __forceinline__ void sp_store(const f32x16 (&acc)[4], bf16_t* out, int ld, int r0, int S, int C, int tid) {
    const int lane = tid & 63, wave = tid >> 6, g = lane >> 5, l31 = lane & 31;
    if (r0 + l31 >= S) return;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int ct = wave + 4 * i;
        if (ct >= C / 32) continue;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            bf16x4 o = {(bf16_t)acc[i][4 * q], (bf16_t)acc[i][4 * q + 1], (bf16_t)acc[i][4 * q + 2], (bf16_t)acc[i][4 * q + 3]};
            *reinterpret_cast<bf16x4*>(out + (size_t)(r0 + l31) * ld + ct * 32 + 8 * q + 4 * g) = o;
        }
    }
}

// ---- forward: one work-group per (image, 32-query block) -----------------------------------------------------------------------
__global__ __launch_bounds__(SNT) void spatial_attn_fwd_kernel(SpParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const SpLds L(p.S, p.C);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nqb = (p.S + 31) / 32;
    const int n = blockIdx.x / nqb, q0 = (blockIdx.x % nqb) * 32;
    const int ld = 3 * p.C;
    const bf16_t* Q = p.qkv + (size_t)n * p.S * ld;
    const bf16_t* K = Q + p.C;
    const bf16_t* V = Q + 2 * p.C;
    float* T = reinterpret_cast<float*>(smem + L.o_t);
    const int STf = L.ST / 4;

    sp_stage_rows(smem + L.o_x, L.RX, Q, ld, q0, p.S, p.C, tid);
    __syncthreads();
    sp_prod(T, STf, smem + L.o_x, L.RX, K, ld, p.S, p.C, wave, lane);
    __syncthreads();
    // softmax over keys: wave w owns query rows 8w .. 8w+7
    for (int r = wave * 8; r < wave * 8 + 8; ++r) {
        float mx = -1e30f;
        for (int s = lane; s < p.S; s += 64) mx = fmaxf(mx, T[r * STf + s] * p.scale);
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
        float sum = 0.0f;
        for (int s = lane; s < p.S; s += 64) { const float e = __expf(T[r * STf + s] * p.scale - mx); T[r * STf + s] = e; sum += e; }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) sum += __shfl_xor(sum, off);
        const float inv = 1.0f / sum;
        bf16_t* pr = reinterpret_cast<bf16_t*>(smem + L.o_p + r * L.SP);
        for (int s = lane; s < ((p.S + 31) & ~31); s += 64) pr[s] = (bf16_t)(s < p.S ? T[r * STf + s] * inv : 0.0f);
        if (lane == 0 && q0 + r < p.S && p.lse) p.lse[(size_t)n * p.S + q0 + r] = mx + __logf(sum);
    }
    f32x16 acc[4];
    sp_apply(acc, smem + L.o_p, L.SP, smem + L.o_m, L.RM, V, ld, p.S, p.C, tid);      // (its first barrier orders the P writes)
    sp_store(acc, p.out + (size_t)n * p.S * p.C, p.C, q0, p.S, p.C, tid);
}

// ---- backward, shared body.  KEYS = false: block = 32 queries -> dQ (and delta); KEYS = true: block = 32 keys -> dK, dV ---------
template <bool KEYS>
__global__ __launch_bounds__(SNT) void spatial_attn_bwd_kernel(SpParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const SpLds L(p.S, p.C);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nb = (p.S + 31) / 32;
    const int n = blockIdx.x / nb, r0 = (blockIdx.x % nb) * 32;
    const int ld = 3 * p.C;
    const bf16_t* Q = p.qkv + (size_t)n * p.S * ld;
    const bf16_t* K = Q + p.C;
    const bf16_t* V = Q + 2 * p.C;
    const bf16_t* dO = p.dout + (size_t)n * p.S * p.C;
    const float* lse = p.lse + (size_t)n * p.S;
    float* delta = p.delta + (size_t)n * p.S;
    float* T = reinterpret_cast<float*>(smem + L.o_t);
    const int STf = L.ST / 4;
    const int Spad = (p.S + 31) & ~31;

    // ---- P (block rows x all columns): rows = queries (KEYS = false) or keys (KEYS = true)
    sp_stage_rows(smem + L.o_x, L.RX, KEYS ? K : Q, ld, r0, p.S, p.C, tid);
    __syncthreads();
    sp_prod(T, STf, smem + L.o_x, L.RX, KEYS ? Q : K, ld, p.S, p.C, wave, lane);
    __syncthreads();
    for (int r = wave * 8; r < wave * 8 + 8; ++r) {
        bf16_t* pr = reinterpret_cast<bf16_t*>(smem + L.o_p + r * L.SP);
        const bool row_ok = r0 + r < p.S;
        const float lrow = (!KEYS && row_ok) ? lse[r0 + r] : 0.0f;
        for (int s = lane; s < Spad; s += 64) {
            float e = 0.0f;
            if (row_ok && s < p.S) e = __expf(T[r * STf + s] * p.scale - (KEYS ? lse[s] : lrow));
            pr[s] = (bf16_t)e;
        }
    }
    __syncthreads();                              // T is about to be overwritten
    // ---- dP: prod(dO_blk, V) (queries) or prod(V_blk, dO) (keys)
    sp_stage_rows(smem + L.o_x, L.RX, KEYS ? V : dO, KEYS ? ld : p.C, r0, p.S, p.C, tid);
    __syncthreads();
    sp_prod(T, STf, smem + L.o_x, L.RX, KEYS ? dO : V, KEYS ? p.C : ld, p.S, p.C, wave, lane);
    __syncthreads();
    for (int r = wave * 8; r < wave * 8 + 8; ++r) {
        const bf16_t* pr = reinterpret_cast<const bf16_t*>(smem + L.o_p + r * L.SP);
        bf16_t* dr = reinterpret_cast<bf16_t*>(smem + L.o_p2 + r * L.SP);
        float drow = 0.0f;
        if constexpr (!KEYS) {                    // delta_q = sum_keys P dP  (= sum_c dO O), published for the dK/dV kernel
            float acc = 0.0f;
            for (int s = lane; s < p.S; s += 64) acc += (float)pr[s] * T[r * STf + s];
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off);
            drow = acc;
            if (lane == 0 && r0 + r < p.S) delta[r0 + r] = acc;
        }
        for (int s = lane; s < Spad; s += 64) {
            float v = 0.0f;
            if (s < p.S) v = p.scale * (float)pr[s] * (T[r * STf + s] - (KEYS ? delta[s] : drow));
            dr[s] = (bf16_t)v;
        }
    }
    f32x16 acc[4];
    if constexpr (!KEYS) {
        sp_apply(acc, smem + L.o_p2, L.SP, smem + L.o_m, L.RM, K, ld, p.S, p.C, tid);                 // dQ = dS K
        sp_store(acc, p.dqkv + (size_t)n * p.S * ld, ld, r0, p.S, p.C, tid);
    } else {
        sp_apply(acc, smem + L.o_p, L.SP, smem + L.o_m, L.RM, dO, p.C, p.S, p.C, tid);                // dV = P^T dO
        sp_store(acc, p.dqkv + (size_t)n * p.S * ld + 2 * p.C, ld, r0, p.S, p.C, tid);
        sp_apply(acc, smem + L.o_p2, L.SP, smem + L.o_m, L.RM, Q, ld, p.S, p.C, tid);                 // dK = dS^T Q
        sp_store(acc, p.dqkv + (size_t)n * p.S * ld + p.C, ld, r0, p.S, p.C, tid);
    }
}

int sp_check(const char* what, const void* a, const void* b, int dtype, int N, int S, int C) {
    if (!a || !b) MAS_FAIL(MAS_EINVAL, "%s: null argument", what);
    if (dtype != MAS_BF16) MAS_FAIL(MAS_EUNSUPPORTED, "%s: bf16 only (the fp32 parity mode keeps the library GEMM path)", what);
    if (N <= 0 || S <= 0 || S > S_MAX || C <= 0 || C > C_MAX || (C % 32)) MAS_FAIL(MAS_EUNSUPPORTED, "%s: needs S <= %d tokens and C %% 32 == 0, C <= %d (got S=%d C=%d)", what, S_MAX, C_MAX, S, C);
    return MAS_OK;
}

template <typename K>
int sp_set_lds(K kern, int bytes, const char* what, mas_devmask_t& mask) {
    unsigned long long bit;
    if (mas_attr_needed(mask, &bit)) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
            MAS_FAIL(MAS_ELAUNCH, "%s: cannot set dynamic LDS size", what);
        mas_attr_done(mask, bit);
    }
    (void)bytes;
    return MAS_OK;
}

}  // namespace

extern "C" int mas_spatial_attn_fwd(const void* qkv, void* out, float* lse, int dtype, int N, int S, int C, void* stream) {
    MAS_ENTER();
    if (int rc = sp_check("spatial_attn_fwd", qkv, out, dtype, N, S, C)) return rc;
    SpParams p{};
    p.qkv = (const bf16_t*)qkv; p.out = (bf16_t*)out; p.lse = lse; p.N = N; p.S = S; p.C = C; p.scale = 1.0f / sqrtf((float)C);
    const SpLds L(S, C);
    static mas_devmask_t mask{0};
    if (int rc = sp_set_lds(spatial_attn_fwd_kernel, L.total, "spatial_attn_fwd", mask)) return rc;
    hipLaunchKernelGGL(spatial_attn_fwd_kernel, dim3((unsigned)(N * ((S + 31) / 32))), dim3(SNT), (size_t)L.total, reinterpret_cast<hipStream_t>(stream), p);
    MAS_CHECK_LAUNCH("spatial_attn_fwd");
    return MAS_OK;
}

extern "C" int mas_spatial_attn_bwd(const void* qkv, const void* dout, const float* lse, float* delta, void* dqkv, int dtype, int N, int S,
                                    int C, void* stream) {
    MAS_ENTER();
    if (int rc = sp_check("spatial_attn_bwd", qkv, dout, dtype, N, S, C)) return rc;
    if (!lse || !delta || !dqkv) MAS_FAIL(MAS_EINVAL, "spatial_attn_bwd: null argument");
    SpParams p{};
    p.qkv = (const bf16_t*)qkv; p.dout = (const bf16_t*)dout; p.dqkv = (bf16_t*)dqkv; p.lse = const_cast<float*>(lse); p.delta = delta;
    p.N = N; p.S = S; p.C = C; p.scale = 1.0f / sqrtf((float)C);
    const SpLds L(S, C);
    static mas_devmask_t m0{0}, m1{0};
    if (int rc = sp_set_lds(spatial_attn_bwd_kernel<false>, L.total, "spatial_attn_bwd", m0)) return rc;
    if (int rc = sp_set_lds(spatial_attn_bwd_kernel<true>, L.total, "spatial_attn_bwd", m1)) return rc;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const dim3 grid((unsigned)(N * ((S + 31) / 32)));
    hipLaunchKernelGGL(spatial_attn_bwd_kernel<false>, grid, dim3(SNT), (size_t)L.total, s, p);      // dQ, delta
    MAS_CHECK_LAUNCH("spatial_attn_bwd(dq)");
    hipLaunchKernelGGL(spatial_attn_bwd_kernel<true>, grid, dim3(SNT), (size_t)L.total, s, p);       // dK, dV (reads delta)
    MAS_CHECK_LAUNCH("spatial_attn_bwd(dkv)");
    return MAS_OK;
}
