// Single-head spatial self-attention core of AttnBlock for gfx950 (reference models/modules.py:174-187: w = softmax_keys(q^T k * C^-1/2),
// h = v w^T over the h*w tokens of one image), forward and backward, bf16 storage / fp32 accumulate.
// Replaces the two torch.bmm + softmax (and, backward, four bmm + the softmax gradient) of round 1 -- ~60 small library launches
// per VQ-IMG step -- by 1 forward + 2 backward launches per AttnBlock.  S = h*w <= 256 tokens, C <= 512 channels (the reference's
// blocks are 16x16x512 for VQ-IMG, 8x8x512 for VQ-SEG), so one 32-token block's whole score row fits LDS and no online softmax is
// needed.  q, k, v are read in place from the fused [N, S, 3C] projection (q | k | v on the channel axis).
//
// Every kernel is built from two tile products (8 waves = two per SIMD since the end of round 4, NW below; MFMA 32x32x16 bf16):
//   prod : T[32 rows][S]   = X_blk[32][C] . Y_all[S][C]^T     (contraction over channels: both operands channel-contiguous, so
//                             Y rows stream straight from global memory as MFMA A fragments, X_blk rows come from LDS; T STAYS IN
//                             THE ACCUMULATORS: wave w holds column tile w (tiles w and w+4 with 4 waves), a lane 16 (32) columns of one row)
//   apply: O[32 rows][C]   = P[32][S] . M_all[S][C]           (contraction over tokens: M tiles are staged in their natural
//                             [token][channel] layout and read with the LDS transpose read ds_read_b64_tr_b16 -- per-lane
//                             addressing as in conv_wgrad.hip -- P rows (bf16) come from LDS)
//   forward        : T = prod(Q_blk, K); P = softmax(scale T), lse;  O_blk  = apply(P, V)
//   backward, dQ   : P = exp(scale prod(Q_blk, K) - lse); dP = prod(dO_blk, V); delta = rowsum(P dP); dS = scale P (dP - delta);
//                    dQ_blk = apply(dS, K)
//   backward, dK/dV: the same with the roles of queries and keys swapped (block = 32 KEYS, columns = queries, lse / delta indexed
//                    by column): dV_blk = apply(P^T, dO), dK_blk = apply(dS^T, Q).
// With 256 work-groups of 4 waves the kernels are bound by exposed memory round trips (1.2 us each, measured with -DSP_TRACE), not
// by MFMA (2 us per work-group) or bandwidth: round 4 keeps 16-32 KiB per wave in flight (prod: next 256-channel step prefetched,
// apply: M tiles four ahead and issued before the softmax) and does the softmax / dS arithmetic on the accumulators instead of
// bouncing fp32 scores through LDS row by row: forward 58 -> 23 us, backward 72 + 95 -> 31 + 43 us (profiles/r04_spatial_attn.txt).
// No atomics: every output element is written exactly once (deterministic).
#include "mas_common.h"
#include <math.h>

namespace {

#ifndef SP_NW
#define SP_NW 8                                 // waves per work-group: 8 (two per SIMD; shipped) or 4 (-DSP_NW=4: the A/B of profiles/r04_spatial_attn.txt, section 8)
#endif
constexpr int NW = SP_NW, SNT = 64 * NW;
constexpr int S_MAX = 256, C_MAX = 512;
constexpr int TPW = S_MAX / 32 / NW;            // 32-column score tiles per wave: 2 | 1
constexpr int CPW = C_MAX / 32 / NW;            // 32-channel output tiles per wave: 4 | 2
constexpr int UPT = 32 * (C_MAX / 8) / SNT;     // 16-byte units of a 32-row tile per thread: 8 | 4
static_assert(NW == 4 || NW == 8, "4 or 8 waves");

// -DSP_TRACE (tools/build_file_variant.sh, never the shipped build): wave 0 of one work-group stamps the 100 MHz wall clock at the phase
// boundaries of the forward kernel; tools/kbench.py sp_attn prints the differences.
#ifdef SP_TRACE
__device__ long long g_sp_trace[32];
#define SP_T(i) do { if (blockIdx.x == 9 && threadIdx.x == 0) g_sp_trace[i] = wall_clock64(); } while (0)
#else
#define SP_T(i) do { } while (0)
#endif

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

// LDS map (bytes); row strides padded so that 32 consecutive rows do not share banks.  The scores never touch LDS in fp32 (round 4):
// they stay in the MFMA accumulators through the softmax / dS arithmetic, only the bf16 P / dS operand of the second product is staged.
struct SpLds {
    int RX;      // X_blk row stride: roundup(C,128)*2 + 16 (channels [C, roundup(C,128)) are zero: the products run whole 128-channel groups)
    int SP;      // P / dS row stride: roundup(S,32)*2 + 16
    int RM;      // M tile row stride: C*2 + 64 (the transpose read wants 4 consecutive rows x 64 B to tile a 256-byte bank row)
    int o_x, o_p, o_p2, o_m0, o_m1, o_r, total;
    __host__ __device__ SpLds(int S, int C) {
        const int Sp = (S + 31) & ~31;           // the products write whole 32-column tiles
        RX = ((C + 127) & ~127) * 2 + 16; SP = Sp * 2 + 16; RM = C * 2 + 64;
        o_x = 0; o_p = o_x + 32 * RX; o_p2 = o_p + 32 * SP; o_m0 = o_p2 + 32 * SP; o_m1 = o_m0 + 32 * RM; o_r = o_m1 + 32 * RM;
        total = o_r + 2 * NW * 32 * 4;           // two [waves][32 rows] fp32 reduction pads
    }
};

// 32 rows [r0, r0+32) of a [rows][ld] bf16 matrix (channel window [0, C)), <= 8 16-byte units per thread: global -> registers ...
// BRANCH-FREE on purpose: a load inside a conditional block makes the number of younger loads unknown to the compiler's wait-count
// pass, and every later wait becomes vmcnt(0) -- which silently serialises a prefetch pipeline (all 62 waits of the first version of
// this kernel were vmcnt(0)).  Out-of-range units / rows read a clamped, valid address; sp_rows_commit zeroes the rows >= S, and a
// clamped unit just rewrites the last real unit with the same bytes.
// (unit -> (row, 16-byte column) without a division per unit: at one wave per SIMD the address VALU work is not hidden by anything,
//  and tid + 256 k divided by a runtime C / 8, twice per tile, was a third of the apply loop)
struct SpMap {
    int r, cu, dr, dc, upr;
    __device__ __forceinline__ SpMap(int C, int tid) { upr = C / 8; r = tid / upr; cu = tid - r * upr; dr = SNT / upr; dc = SNT - dr * upr; }
};
__device__ __forceinline__ void sp_rows_fetch(u32x4 (&pf)[UPT], const bf16_t* src, int ld, int r0, int S, const SpMap& mp) {
    int r = mp.r, cu = mp.cu;
#pragma unroll
    for (int k = 0; k < UPT; ++k) {
        const int rr = min(r, 31), cc = r > 31 ? mp.upr - 1 : cu;          // units past the tile re-read its last unit
        pf[k] = *reinterpret_cast<const u32x4*>(src + (size_t)min(r0 + rr, S - 1) * ld + cc * 8);
        cu += mp.dc; r += mp.dr;
        if (cu >= mp.upr) { cu -= mp.upr; ++r; }
    }
}
// ... -> LDS with row stride RS (rows >= S are zero)
__device__ __forceinline__ void sp_rows_commit(const u32x4 (&pf)[UPT], unsigned char* dst, int RS, int r0, int S, const SpMap& mp) {
    int r = mp.r, cu = mp.cu;
#pragma unroll
    for (int k = 0; k < UPT; ++k) {
        const int rr = min(r, 31), cc = r > 31 ? mp.upr - 1 : cu;
        const bool ok = r0 + rr < S;
        const u32x4 v = {ok ? pf[k][0] : 0u, ok ? pf[k][1] : 0u, ok ? pf[k][2] : 0u, ok ? pf[k][3] : 0u};
        *reinterpret_cast<u32x4*>(dst + rr * RS + cc * 16) = v;
        cu += mp.dc; r += mp.dr;
        if (cu >= mp.upr) { cu -= mp.upr; ++r; }
    }
}

// Which score columns a lane holds: wave w owns the 32-column tiles w and w + 4 (accumulator slots mt[0], mt[1]); register r of
// slot ti is column m = mt[ti] * 32 + (r & 3) + 8 (r >> 2) + 4 g of block row n = l31  (g = lane >> 5, l31 = lane & 31).
// The 8 work-groups of an image run on one XCD at the same time and would otherwise walk K / V in lockstep: every line would be
// requested by all 8 while the first miss is still on the fabric, and all of them would sit out the full 1.2 us on every step
// (27 GB/s per CU measured = the L1 miss queue x 128 B / that latency, with 8 CUs spending it on the SAME lines).  `rot` (the block's
// index within its image) de-phases them: odd blocks start with their upper column tile, bit 1 flips the order of the two channel
// halves (here), and the apply walks its token tiles starting at tile `rot` -- the lines one block waits for are hits for the rest.
__device__ __forceinline__ void sp_tiles(int (&mt)[TPW], int wave, int S, int rot) {
    if constexpr (TPW == 2) {
        const bool swap = (wave + 4 < (S + 31) / 32) && (rot & 1);
        mt[0] = swap ? wave + 4 : wave;
        mt[TPW - 1] = swap ? wave : wave + 4;
    } else {
        mt[0] = wave;                            // 8 waves: one column tile each
    }
}
__device__ __forceinline__ int sp_col(const int (&mt)[TPW], int ti, int r, int g) { return mt[ti] * 32 + (r & 3) + 8 * (r >> 2) + 4 * g; }

// sc[ti][r] = sum_c X[r0 + n][c] * Y[m][c]: the 32 block rows are staged into xs (LDS) by this call, the rows of Y stream straight
// from global memory as MFMA A fragments (lane = m), B operand (lane = n) from LDS; the result stays in the accumulators.
// At one wave per SIMD nothing hides anything, so the loop is written for the two things that were found to cost (SP_TRACE + the ISA):
// * memory-level parallelism: a step is 128 channels of one column tile (8 fragments, 8 KiB per wave); a ring of four fragment
//   buffers keeps THREE steps in flight behind the one being multiplied, and the first three are issued before the block rows are
//   committed to LDS, so that round trip overlaps too;
// * straight-line code: every load is unconditional (sp_rows_fetch explains why), the block rows are zero-padded to whole groups so
//   that a step has no per-fragment branch, and a step that does not exist for this wave / this C still loads (one clamped line) and
//   only skips its MFMAs under a wave-uniform branch.  The first version, with two 16-fragment buffers selected by `k & 1 ? a0 : a1`
//   and per-fragment bounds branches, compiled to 2700 v_accvgpr moves per product.
__device__ __forceinline__ void sp_prod(f32x16 (&sc)[TPW], const int (&mt)[TPW], int rot, unsigned char* xs, int RX, const bf16_t* X, int ldx,
                                        int r0, const bf16_t* Y, int ld, int S, int C, int tid) {
    const int lane = tid & 63, g = lane >> 5, l31 = lane & 31;
    const int n_mt = (S + 31) / 32, n_grp = (C + 127) / 128;            // 1..4 channel groups
    const int my_tiles = (mt[0] < n_mt) + (TPW == 2 ? (mt[TPW - 1] < n_mt) : 0);      // 0, 1 or 2 (S <= 256); slot 0 is the valid one when 1
    const int gr = (rot >> 1) & 3;                                      // (the blocks of an image start on different groups)
    const SpMap mp(C, tid);
    u32x4 xr_[UPT];
    sp_rows_fetch(xr_, X, ldx, r0, S, mp);
    bf16x8 a[4][8];
    auto fetch = [&](bf16x8 (&f)[8], int k) {
        const int ti = k >> 2, grp = ((k & 3) + gr) & 3;
        const int m = mt[ti] < n_mt ? min(mt[ti] * 32 + l31, S - 1) : 0;          // columns >= S: a real row's bytes, masked by every consumer
        const bf16_t* yr = Y + (size_t)m * ld + 8 * g;
#pragma unroll
        for (int u = 0; u < 8; ++u) f[u] = *reinterpret_cast<const bf16x8*>(yr + min(grp * 128 + 16 * u, C - 16));
    };
    fetch(a[0], 0); fetch(a[1], 1); fetch(a[2], 2);
    SP_T(16);
    __syncthreads();                              // whoever read xs before is done
    sp_rows_commit(xr_, xs, RX, r0, S, mp);
    if (C & 127) {                                // zero channels [C, roundup(C, 128)) of the 32 rows
        const int padu = (128 - (C & 127)) / 8;
        for (int u = tid; u < 32 * padu; u += SNT) *reinterpret_cast<u32x4*>(xs + (u / padu) * RX + (C / 8 + u % padu) * 16) = u32x4{0u, 0u, 0u, 0u};
    }
    __syncthreads();
    SP_T(17);
#pragma unroll
    for (int ti = 0; ti < TPW; ++ti)
#pragma unroll
        for (int r = 0; r < 16; ++r) sc[ti][r] = 0.0f;
    const unsigned char* xr = xs + l31 * RX + 16 * g;
#pragma unroll
    for (int k = 0; k < 4 * TPW; ++k) {
        if (k + 3 < 4 * TPW) fetch(a[(k + 3) & 3], k + 3);
        const int grp = ((k & 3) + gr) & 3;
        if ((k >> 2) < my_tiles && grp < n_grp) {
            bf16x8 b[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) b[u] = *reinterpret_cast<const bf16x8*>(xr + (grp * 128 + 16 * u) * 2);
#pragma unroll
            for (int u = 0; u < 8; ++u) mma16(sc[k >> 2], a[k & 3][u], b[u]);
        }
        SP_T(18 + k);
    }
}

// bf16 rows of P / dS for the second product: lane (g, l31) owns 4 consecutive columns per accumulator quad -> one 8-byte LDS write
__device__ __forceinline__ void sp_put_rows(unsigned char* ps, int SP, const f32x16 (&v)[TPW], int S, const int (&mts)[TPW], int g, int l31) {
    const int n_mt = (S + 31) / 32;
#pragma unroll
    for (int ti = 0; ti < TPW; ++ti) {
        const int mt = mts[ti];
        if (mt < n_mt) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const bf16x4 o = {(bf16_t)v[ti][4 * q], (bf16_t)v[ti][4 * q + 1], (bf16_t)v[ti][4 * q + 2], (bf16_t)v[ti][4 * q + 3]};
                *reinterpret_cast<bf16x4*>(ps + l31 * SP + (mt * 32 + 8 * q + 4 * g) * 2) = o;
            }
        }
    }
}

// row-wise reduction of one value per lane over ALL columns: the two lane halves (xor 32), then the 4 waves through an LDS pad.
template <bool MAX>
__device__ __forceinline__ float sp_row_reduce(float v, float* pad, int wave, int g, int l31) {
    const float o = __shfl_xor(v, 32);
    v = MAX ? fmaxf(v, o) : v + o;
    if (g == 0) pad[wave * 32 + l31] = v;
    __syncthreads();
    float acc = pad[l31];
#pragma unroll
    for (int w = 1; w < NW; ++w) acc = MAX ? fmaxf(acc, pad[w * 32 + l31]) : acc + pad[w * 32 + l31];
    return acc;
}

// O[n][c] = sum_s P[n][s] * M[s][c]: P (bf16 [32][S], LDS), M global [S][ld]; result acc tiles: wave w owns channel tiles
// ct = w, w+4, ... (<= 4 of them, C <= 512): acc[i][r] = D[c = ct*32 + (r&3)+8(r>>2)+4g][n = l31].
// The 32-token tiles of M go global -> registers -> LDS, FOUR tiles ahead: sp_apply_issue puts tiles 0..3 in flight (the callers do
// that before their softmax / dS arithmetic, M does not depend on it), sp_apply_run commits tile t+1 into the LDS buffer tile t-1 just
// left, re-arms its registers with tile t+4 and runs tile t's MFMAs -- one barrier per tile, and a tile's round trip (1.2 us measured)
// is spread over four tiles of work instead of being exposed 8 times (P.V phase of the forward: 11.3 -> 6.8 us).
__device__ __forceinline__ int sp_rot_tile(int t, int rot, int nt) { const int tt = t + rot; return tt >= nt ? tt - nt : tt; }   // rot < nt

__device__ __forceinline__ void sp_apply_issue(u32x4 (&pf)[4][UPT], const bf16_t* M, int ld, int S, int C, int rot, int tid) {
    const int nt = (S + 31) / 32;
    const SpMap mp(C, tid);
#pragma unroll
    for (int t = 0; t < 4; ++t)
        if (t < nt) sp_rows_fetch(pf[t], M, ld, sp_rot_tile(t, rot, nt) * 32, S, mp);
}

__device__ __forceinline__ void sp_apply_run(f32x16 (&acc)[CPW], u32x4 (&pf)[4][UPT], const unsigned char* ps, int SP, unsigned char* ms0,
                                             unsigned char* ms1, int RM, const bf16_t* M, int ld, int S, int C, int rot, int tid) {
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 5, l31 = lane & 31, G16 = (lane >> 4) & 1, sl = lane & 15;
    const int n_ct = C / 32, nt = (S + 31) / 32;
    const SpMap mp(C, tid);
#pragma unroll
    for (int i = 0; i < CPW; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;
    __syncthreads();                              // the callers' P / dS rows are written; earlier readers of ms0 / ms1 are done
    sp_rows_commit(pf[0], ms0, RM, sp_rot_tile(0, rot, nt) * 32, S, mp);
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        if (t < nt) {
            const int s0 = sp_rot_tile(t, rot, nt) * 32;
            const unsigned char* cur = (t & 1) ? ms1 : ms0;
            __syncthreads();                      // tile t is visible; every wave has left tile t-1's buffer
            // (operands first, all of them, then the commit / re-arm traffic, then the 8 MFMAs: with the reads inside the per-channel-
            //  tile bounds branch each MFMA waited out its own LDS round trip -- 1.1 us per tile for 0.12 us of MFMA)
            bf16x8 bq[2], aq[2][CPW];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                bq[ks] = *reinterpret_cast<const bf16x8*>(ps + l31 * SP + (s0 + ks * 16 + 8 * g) * 2);     // P[n][s0 + 16 ks + 8 g ..+7]
                // transpose-read lane addressing: token (16 ks + 8 g + (sl >> 2)) (+4 for the second read), channels ct*32 + 16*G16 + 4*(sl&3) ..+3
                const unsigned char* a_lane = cur + (ks * 16 + 8 * g + (sl >> 2)) * RM + (16 * G16 + 4 * (sl & 3)) * 2;
#pragma unroll
                for (int i = 0; i < CPW; ++i) {
                    const int ct = min(wave + NW * i, n_ct - 1);         // a channel tile past C recomputes the last real one; sp_store drops it
                    aq[ks][i] = sp_tr_frag(a_lane + ct * 64, a_lane + ct * 64 + 4 * RM);
                }
            }
            if (t + 1 < nt) sp_rows_commit(pf[(t + 1) & 3], (t & 1) ? ms0 : ms1, RM, sp_rot_tile(t + 1, rot, nt) * 32, S, mp);
            if (t + 4 < nt) sp_rows_fetch(pf[t & 3], M, ld, sp_rot_tile(t + 4, rot, nt) * 32, S, mp);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int i = 0; i < CPW; ++i) mma16(acc[i], aq[ks][i], bq[ks]);
            SP_T(6 + t);
        }
    }
    __syncthreads();                              // the last tile's reads are done: the caller may reuse both buffers
}

// writes acc tiles to out[n][c] (rows r0 + l31 < S), 4 consecutive channels (8 bytes) per accumulator quad
__device__ __forceinline__ void sp_store(const f32x16 (&acc)[CPW], bf16_t* out, int ld, int r0, int S, int C, int tid) {
    const int lane = tid & 63, wave = tid >> 6, g = lane >> 5, l31 = lane & 31;
    if (r0 + l31 >= S) return;
#pragma unroll
    for (int i = 0; i < CPW; ++i) {
        const int ct = wave + NW * i;
        if (ct >= C / 32) continue;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            bf16x4 o = {(bf16_t)acc[i][4 * q], (bf16_t)acc[i][4 * q + 1], (bf16_t)acc[i][4 * q + 2], (bf16_t)acc[i][4 * q + 3]};
            *reinterpret_cast<bf16x4*>(out + (size_t)(r0 + l31) * ld + ct * 32 + 8 * q + 4 * g) = o;
        }
    }
}

// Work-group -> (image, 32-row block).  The hardware deals work-groups to the 8 XCDs round-robin (id % 8), and each XCD has its own
// 4 MiB L2: with the plain id / nb decode the 8 row blocks of one image land on 8 different XCDs, every L2 sees the K and V of every
// image (32 x 512 KiB at the benched shape) and all of it streams from the fabric 8 times.  Here XCD x takes the x-th contiguous
// eighth of the (image, block) list, so the blocks that share an image's K / V run on the same L2 at the same time.
__device__ __forceinline__ int sp_logical_block(int total) {
    const int w = blockIdx.x;
    if (total % 8) return w;
    return (w % 8) * (total / 8) + w / 8;
}

// ---- forward: one work-group per (image, 32-query block) -----------------------------------------------------------------------
__global__ __launch_bounds__(SNT) void spatial_attn_fwd_kernel(SpParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const SpLds L(p.S, p.C);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), g = lane >> 5, l31 = lane & 31;
    const int nqb = (p.S + 31) / 32;
    const int blk = sp_logical_block((int)gridDim.x);
    const int n = blk / nqb, q0 = (blk % nqb) * 32;
    const int ld = 3 * p.C;
    const bf16_t* Q = p.qkv + (size_t)n * p.S * ld;
    const bf16_t* K = Q + p.C;
    const bf16_t* V = Q + 2 * p.C;
    float* pad = reinterpret_cast<float*>(smem + L.o_r);

    const int rot = blk % nqb;
    int mt[TPW];
    sp_tiles(mt, wave, p.S, rot);
    SP_T(0);
    f32x16 sc[TPW];
    sp_prod(sc, mt, rot, smem + L.o_x, L.RX, Q, ld, q0, K, ld, p.S, p.C, tid);
    SP_T(1);
    u32x4 pf[4][UPT];
    sp_apply_issue(pf, V, ld, p.S, p.C, rot, tid);                 // four V tiles travel while the softmax runs
    SP_T(2);
    // softmax over keys, in the accumulators: lane (g, l31) holds 32 of query l31's scores
    float mx = -1e30f;
#pragma unroll
    for (int ti = 0; ti < TPW; ++ti)
#pragma unroll
        for (int r = 0; r < 16; ++r)
            if (sp_col(mt, ti, r, g) < p.S) mx = fmaxf(mx, sc[ti][r] * p.scale);
    mx = sp_row_reduce<true>(mx, pad, wave, g, l31);
    float sum = 0.0f;
#pragma unroll
    for (int ti = 0; ti < TPW; ++ti)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float e = sp_col(mt, ti, r, g) < p.S ? __expf(sc[ti][r] * p.scale - mx) : 0.0f;
            sc[ti][r] = e;
            sum += e;
        }
    sum = sp_row_reduce<false>(sum, pad + NW * 32, wave, g, l31);
    const float inv = 1.0f / sum;
#pragma unroll
    for (int ti = 0; ti < TPW; ++ti)
#pragma unroll
        for (int r = 0; r < 16; ++r) sc[ti][r] *= inv;
    sp_put_rows(smem + L.o_p, L.SP, sc, p.S, mt, g, l31);
    if (wave == 0 && g == 0 && q0 + l31 < p.S && p.lse) p.lse[(size_t)n * p.S + q0 + l31] = mx + __logf(sum);
    SP_T(3);
    f32x16 acc[CPW];
    sp_apply_run(acc, pf, smem + L.o_p, L.SP, smem + L.o_m0, smem + L.o_m1, L.RM, V, ld, p.S, p.C, rot, tid);
    SP_T(4);
    sp_store(acc, p.out + (size_t)n * p.S * p.C, p.C, q0, p.S, p.C, tid);
    SP_T(5);
}

// ---- backward, shared body.  KEYS = false: block = 32 queries -> dQ (and delta); KEYS = true: block = 32 keys -> dK, dV ---------
template <bool KEYS>
__global__ __launch_bounds__(SNT) void spatial_attn_bwd_kernel(SpParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const SpLds L(p.S, p.C);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), g = lane >> 5, l31 = lane & 31;
    const int nb = (p.S + 31) / 32;
    const int blk = sp_logical_block((int)gridDim.x);
    const int n = blk / nb, r0 = (blk % nb) * 32;
    const int ld = 3 * p.C;
    const bf16_t* Q = p.qkv + (size_t)n * p.S * ld;
    const bf16_t* K = Q + p.C;
    const bf16_t* V = Q + 2 * p.C;
    const bf16_t* dO = p.dout + (size_t)n * p.S * p.C;
    const float* lse = p.lse + (size_t)n * p.S;
    float* delta = p.delta + (size_t)n * p.S;
    float* pad = reinterpret_cast<float*>(smem + L.o_r);
    const bool row_ok = r0 + l31 < p.S;

    // ---- P (block rows x all columns): rows = queries (KEYS = false) or keys (KEYS = true); lse belongs to the QUERY
    const int rot = blk % nb;
    int mt[TPW];
    sp_tiles(mt, wave, p.S, rot);
    f32x16 pv[TPW], dp[TPW];
    sp_prod(pv, mt, rot, smem + L.o_x, L.RX, KEYS ? K : Q, ld, r0, KEYS ? Q : K, ld, p.S, p.C, tid);
    const float lrow = (!KEYS && row_ok) ? lse[r0 + l31] : 0.0f;
#pragma unroll
    for (int ti = 0; ti < TPW; ++ti)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = sp_col(mt, ti, r, g);
            const float l = KEYS ? lse[min(m, p.S - 1)] : lrow;
            pv[ti][r] = (row_ok && m < p.S) ? __expf(pv[ti][r] * p.scale - l) : 0.0f;
        }
    if constexpr (KEYS) sp_put_rows(smem + L.o_p, L.SP, pv, p.S, mt, g, l31);         // dV = P^T dO wants P itself
    // ---- dP: prod(dO_blk, V) (queries) or prod(V_blk, dO) (keys)
    sp_prod(dp, mt, rot + 2, smem + L.o_x, L.RX, KEYS ? V : dO, KEYS ? ld : p.C, r0, KEYS ? dO : V, KEYS ? p.C : ld, p.S, p.C, tid);
    u32x4 pf[4][UPT];
    if constexpr (!KEYS) sp_apply_issue(pf, K, ld, p.S, p.C, rot, tid);               // K tiles travel under the delta / dS arithmetic
    float drow = 0.0f;
    if constexpr (!KEYS) {                        // delta_q = sum_keys P dP  (= sum_c dO O), published for the dK/dV kernel
        float part = 0.0f;
#pragma unroll
        for (int ti = 0; ti < TPW; ++ti)
#pragma unroll
            for (int r = 0; r < 16; ++r) part += pv[ti][r] * dp[ti][r];
        drow = sp_row_reduce<false>(part, pad, wave, g, l31);
        if (wave == 0 && g == 0 && row_ok) delta[r0 + l31] = drow;
    }
#pragma unroll
    for (int ti = 0; ti < TPW; ++ti)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = sp_col(mt, ti, r, g);
            float d = drow;
            if constexpr (KEYS) d = delta[min(m, p.S - 1)];
            dp[ti][r] = p.scale * pv[ti][r] * (dp[ti][r] - d);         // dS (pv is 0 outside the valid rows / columns)
        }
    if constexpr (KEYS) sp_apply_issue(pf, dO, p.C, p.S, p.C, rot, tid);              // (after the delta loads: nothing waits behind them)
    sp_put_rows(smem + L.o_p2, L.SP, dp, p.S, mt, g, l31);
    f32x16 acc[CPW];
    if constexpr (!KEYS) {
        sp_apply_run(acc, pf, smem + L.o_p2, L.SP, smem + L.o_m0, smem + L.o_m1, L.RM, K, ld, p.S, p.C, rot, tid); // dQ = dS K
        sp_store(acc, p.dqkv + (size_t)n * p.S * ld, ld, r0, p.S, p.C, tid);
    } else {
        sp_apply_run(acc, pf, smem + L.o_p, L.SP, smem + L.o_m0, smem + L.o_m1, L.RM, dO, p.C, p.S, p.C, rot, tid); // dV = P^T dO
        sp_apply_issue(pf, Q, ld, p.S, p.C, rot, tid);                                                              // (Q tiles travel under the store)
        sp_store(acc, p.dqkv + (size_t)n * p.S * ld + 2 * p.C, ld, r0, p.S, p.C, tid);
        sp_apply_run(acc, pf, smem + L.o_p2, L.SP, smem + L.o_m0, smem + L.o_m1, L.RM, Q, ld, p.S, p.C, rot, tid); // dK = dS^T Q
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

#ifdef SP_TRACE
extern "C" int mas_sp_trace(long long* out16) { return (int)hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_sp_trace), sizeof(long long) * 32); }
#endif

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
