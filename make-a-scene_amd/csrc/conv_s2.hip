// Stride-2 3x3 convolutions (reference models/modules.py:62-81 Downsample: F.pad(x, (0, 1, 0, 1)) + Conv2d(stride 2, padding 0)).
//
// wgrad_s2_kernel -- their weight gradient  dW[co][kh][kw][ci] = sum over (n, i, j) of dy[n, i, j, co] * x[n, 2 i + kh, 2 j + kw, ci]
// on the transpose-read GEMM of conv_wgrad_dma.hip / conv1x1.hip: K = output pixels, both operands staged [pixel][channel] by LDS-DMA,
// fragments by ds_read_b64_tr_b16 -- whose per-lane addresses make the stride free: the B fragment of tap (kh, kw) reads patch pixel
// (2 r + kh, 2 c + kw).  conv_wgrad.hip's stride-2 instance (round 1: transposed register staging, 4x4-channel micro-tiles) ran at
// ~300 TFLOP/s: 0.40 ms for 128 -> 128 @256^2 where streaming x once takes 0.1 ms (profiles/r03_conv_shapes.txt).
//   * work-group = 8 waves = one 128 (co) x 64 (ci) x 9-tap accumulator block (wave: 32 x 32 x 9 = 144 VGPRs), as conv_wgrad_dma.hip;
//   * tile = 4 x 16 output pixels: dy tile 64 pixels x 256 B (16 KiB, 64-byte blocks ^ (pixel & 3)), x patch 9 x 33 pixels x 128 B
//     (38 KiB, 64-byte blocks ^ ((column >> 1) & 1): the four pixels of a transpose read sit 256 B apart, so this read is 2-way
//     bank-conflicted -- the kernel is bound by the 54 KiB of DMA per 36 MFMAs of a wave, not by LDS);
//   * double-buffered, one barrier per tile; split-K over tiles into slabs for mas_wgrad_reduce (fixed order: bitwise reproducible);
//     bias gradient = one more MFMA per k-step against an all-ones operand.
#include "mas_common.h"
#include <type_traits>

namespace {

typedef __attribute__((ext_vector_type(4))) int s2_i32x4;
typedef __attribute__((ext_vector_type(4))) short s2_s16x4;
__device__ __forceinline__ void s2_dma16(s2_i32x4 rs, unsigned lds, int vo) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" :: "s"(lds), "v"(vo), "s"(rs) : "memory", "m0");
}
__device__ __forceinline__ s2_i32x4 s2_rsrc(const void* base, unsigned bytes) {
    const unsigned long long a = (unsigned long long)base;
    s2_i32x4 r = {(int)(unsigned)a, (int)(unsigned)(a >> 32), (int)bytes, 0x00020000};
    r[0] = __builtin_amdgcn_readfirstlane(r[0]); r[1] = __builtin_amdgcn_readfirstlane(r[1]);
    r[2] = __builtin_amdgcn_readfirstlane(r[2]); r[3] = __builtin_amdgcn_readfirstlane(r[3]);
    return r;
}
__device__ __forceinline__ bf16x8 s2_tr(const unsigned char* a0, const unsigned char* a1) {
    const s2_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s2_s16x4*)a0);
    const s2_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s2_s16x4*)a1);
    const __attribute__((ext_vector_type(8))) short v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return *reinterpret_cast<const bf16x8*>(&v);
}

struct S2WgradParams {
    const unsigned char* x; const unsigned char* dy; float* part; float* part_bias;
    int N, H, W, Cin, Ho, Wo, Cout;
    int tiles_h, tiles_w, n_tiles, n_co_t, n_ci_t, nsplit;
};

constexpr int S2_NT = 512, S2_TH = 4, S2_TW = 16, S2_PH = 9, S2_PW = 33, S2_NPP = S2_PH * S2_PW;   // 297 patch pixels
constexpr int S2_DY = 64 * 256;                // 16 KiB
constexpr int S2_XP = 38 * 1024;               // 297 pixels x 128 B -> 38 DMA pieces of 8 pixels
constexpr int S2_STAGE = S2_DY + S2_XP;
constexpr int S2_LDS = 2 * S2_STAGE;           // 108 KiB
constexpr int S2_OOB = (int)0x80000000;

__global__ __launch_bounds__(S2_NT, 1) void wgrad_s2_kernel(S2WgradParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char s2_smem[];
    const unsigned lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) unsigned char*)s2_smem;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 5, l31 = lane & 31, G16 = (lane >> 4) & 1, sl = lane & 15;
    const int wco = (wave & 3) * 32, wci = (wave >> 2) * 32;
    int bid = blockIdx.x;
    const int split = bid % p.nsplit; bid /= p.nsplit;
    const int ci_t = bid % p.n_ci_t, co_t = bid / p.n_ci_t;
    const int co0 = co_t * 128, ci0 = ci_t * 64;

    const s2_i32x4 rs_dy = s2_rsrc(p.dy, (unsigned)((size_t)p.N * p.Ho * p.Wo * p.Cout * 2));
    const s2_i32x4 rs_x = s2_rsrc(p.x, (unsigned)((size_t)p.N * p.H * p.W * p.Cin * 2));

    // ---- DMA plan.  dY: 16 pieces of 4 pixels x 256 B; wave w moves pieces 2 w, 2 w + 1 (tile row w >> 1, columns 8 (w & 1) + 4 j + (lane >> 4));
    //      physical 64-byte block (lane >> 2) & 3 = logical block ^ (column & 3).  Patch: pieces wave + 8 k (k < 5, piece < 38) of 8 pixels x 128 B;
    //      lane: pixel 8 piece + (lane >> 3), physical block (lane >> 2) & 1 = logical block ^ ((patch column >> 1) & 1), 16-byte slot lane & 3.
    const int lp = lane >> 4;
    const int dsrc = ((((lane >> 2) & 3) ^ lp) << 6) + ((lane & 3) << 4);
    auto issue = [&](int t, int stage) {
        const int tw_i = t % p.tiles_w; const int q = t / p.tiles_w;
        const int th_i = q % p.tiles_h, n = q / p.tiles_h;
        const int oh0 = th_i * S2_TH, ow0 = tw_i * S2_TW;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int piece = 2 * wave + j;
            const int oh = oh0 + (piece >> 2), ow = ow0 + 4 * (piece & 3) + lp;
            const bool ok = oh < p.Ho && ow < p.Wo;
            s2_dma16(rs_dy, __builtin_amdgcn_readfirstlane(lds0 + stage * S2_STAGE + piece * 1024),
                     ok ? (((n * p.Ho + oh) * p.Wo + ow) * p.Cout + co0) * 2 + dsrc : S2_OOB);
        }
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            const int piece = wave + 8 * k;
            if (piece < 38) {
                const int px = piece * 8 + (lane >> 3);
                const int pr = px / S2_PW, pc = px - pr * S2_PW;
                const int ih = 2 * oh0 + pr, iw = 2 * ow0 + pc;
                const bool ok = px < S2_NPP && ih < p.H && iw < p.W;                 // (rows / columns past the map = the one-sided zero padding)
                const int blk = ((lane >> 2) & 1) ^ ((pc >> 1) & 1);
                s2_dma16(rs_x, __builtin_amdgcn_readfirstlane(lds0 + stage * S2_STAGE + S2_DY + piece * 1024),
                         ok ? (((n * p.H + ih) * p.W + iw) * p.Cin + ci0) * 2 + (blk << 6) + ((lane & 3) << 4) : S2_OOB);
            }
        }
    };

    // ---- transpose-read lane addressing: pixel column 8 g + t4 (+ 4) of tile row r; A: channels wco + 16 G16 + 4 (sl & 3) ..+3 of the dY row,
    //      B: channels wci + ... of patch pixel (2 r + kh, 2 column + kw)
    const int t4 = sl >> 2;
    const int a_off = (8 * g + t4) * 256 + (((wco >> 5) ^ t4) << 6) + 32 * G16 + 8 * (sl & 3);
    int b_off[3];                                  // per kw (tile row 0, kh = 0); + (2 r + kh) * 33 * 128 per row; second read + 8 * 128
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
        const int col = 2 * (8 * g + t4) + kw;
        b_off[kw] = S2_DY + col * 128 + ((((wci >> 5) ^ (col >> 1)) & 1) << 6) + 32 * G16 + 8 * (sl & 3);
    }

    f32x16 acc[9], accb;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
#pragma unroll
    for (int r = 0; r < 16; ++r) accb[r] = 0.0f;
    bf16x8 ones;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = (bf16_t)1.0f;
    const bool do_bias = p.part_bias != nullptr && ci_t == 0 && wci == 0;       // (wave-uniform)

    const int n_mine = (p.n_tiles - split + p.nsplit - 1) / p.nsplit;
    if (n_mine > 0) issue(split, 0);
    auto tile = [&](int it, auto stage_c) {
        constexpr int ST = decltype(stage_c)::value;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (it + 1 < n_mine) issue(split + (it + 1) * p.nsplit, ST ^ 1);
        const unsigned char* sb = s2_smem + ST * S2_STAGE;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const unsigned char* a0 = sb + a_off + r * 16 * 256;
            const bf16x8 afr = s2_tr(a0, a0 + 4 * 256);
#pragma unroll
            for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) {
                    const unsigned char* b0 = sb + b_off[kw] + (2 * r + kh) * S2_PW * 128;
                    mma16(acc[kh * 3 + kw], afr, s2_tr(b0, b0 + 8 * 128));          // D[co][ci]
                }
            if (do_bias) mma16(accb, afr, ones);
        }
    };
    for (int it = 0; it < n_mine; it += 2) {
        tile(it, std::integral_constant<int, 0>{});
        if (it + 1 < n_mine) tile(it + 1, std::integral_constant<int, 1>{});
    }

    float* pw = p.part + (size_t)split * ((size_t)p.Cout * 9 * p.Cin);
    const int ci = ci0 + wci + l31;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = co0 + wco + acc_row(lane, r);
            pw[((size_t)co * 9 + t) * p.Cin + ci] = acc[t][r];
        }
    if (do_bias && l31 == 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) p.part_bias[(size_t)split * p.Cout + co0 + wco + acc_row(lane, r)] = accb[r];
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// conv_s2_fwd_kernel -- Downsample's forward:  y[n, i, j, co] = sum w[co][kh][kw][ci] * x[n, 2 i + kh, 2 j + kw, ci] + bias[co].
// conv_fwd.hip's stride-2 instance stages a (2 x 8 + 1) x (2 x 16 + 2)-pixel x 64-channel patch (74 KiB) per 128 output pixels through
// registers and runs at ~300 TFLOP/s (0.56 ms for 128 -> 128 @256^2).  A stride-2 patch is 4.4x the bytes per output pixel of a stride-1
// one, so the tile cannot be the wide kernel's: here a STAGE is one filter row kh of one 32-channel chunk -- the 8 input rows
// 2 r + kh of the 8 x 32-pixel output tile (8 x 65 pixels x 64 B = 33 KiB; rows are re-fetched per kh, from L2) and that row's 3 taps of
// weights for 128 couts (24 KiB, gathered by the DMA from the 64-channel K64 image: a lane asks for LOGICAL 16-byte slot 4 h + s of
// its row) -- double-buffered (116 KiB), both by LDS-DMA, one barrier per stage, 24 MFMAs per wave and stage.
//   waves: 2 (64 couts) x 4 (output rows 2 p, 2 p + 1); B fragment of pixel (r, c), tap kw = patch pixel r * 65 + 2 c + kw: one
//   ds_read_b128 whose 16-byte slot is ^ ((pixel >> 1) & 3) (2-way bank conflict: neighbouring lanes sit 128 B apart); weight rows
//   slot ^ ((row >> 2) & 3) as in conv3x3_wide.hip.  Epilogue as conv3x3_stream.hip (permlane32_swap: 8 consecutive couts per lane).
struct S2FwdParams {
    const unsigned char* x; const unsigned char* w; const float* bias; unsigned char* y;
    int N, H, W, Cin, Ho, Wo, Cout, rows_pad, n_chunks32, tiles_h, tiles_w, n_ct;
};
constexpr int F2_TH = 8, F2_TW = 32, F2_PW = 65, F2_NPP = F2_TH * F2_PW;      // 520 patch pixels per stage
constexpr int F2_PATCH = 33 * 1024;             // 520 x 64 B -> 33 DMA pieces of 16 pixels
constexpr int F2_WT = 128 * 64;                 // one tap: 128 couts x 64 B
constexpr int F2_STAGE = F2_PATCH + 3 * F2_WT;  // 58368 B
constexpr int F2_LDS = 2 * F2_STAGE;

__global__ __launch_bounds__(S2_NT, 1) void conv_s2_fwd_kernel(S2FwdParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char s2_smem[];
    const unsigned lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) unsigned char*)s2_smem;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 5, l31 = lane & 31;
    const int wave_c = wave & 1, wave_p = wave >> 1;
    int t = blockIdx.x;
    const int ct = t % p.n_ct; t /= p.n_ct;
    const int tw_i = t % p.tiles_w; t /= p.tiles_w;
    const int th_i = t % p.tiles_h, n = t / p.tiles_h;
    const int oh0 = th_i * F2_TH, ow0 = tw_i * F2_TW, c0 = ct * 128;

    const s2_i32x4 rs_x = s2_rsrc(p.x, (unsigned)((size_t)p.N * p.H * p.W * p.Cin * 2));
    const s2_i32x4 rs_w = s2_rsrc(p.w, (unsigned)((size_t)(p.Cin / 64 + (p.Cin % 64 ? 1 : 0)) * 9 * p.rows_pad * 128));

    // ---- DMA plan.  Patch piece wave + 8 k (k < 5, piece < 33): pixel P = 16 piece + (lane >> 2) = row P / 65, column P % 65; physical
    //      16-byte slot lane & 3 holds LOGICAL slot ^ ((P >> 1) & 3).  Weight piece wave + 8 k (k < 3) = tap piece / 8, rows
    //      16 (piece & 7) + (lane >> 2); physical slot lane & 3 holds logical slot ^ ((row >> 2) & 3), fetched from the K64 image's row
    //      (128 B, its own swizzle (row >> 1) & 7) at logical 16-byte slot 4 h + s of the 64-channel chunk.
    int xo[5];                                     // per piece: byte offset of (input row 2 (oh0 + r), column 2 ow0 + pc, channel slot) or out of range
    unsigned xok = 0;
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        const int piece = wave + 8 * k, P = piece * 16 + (lane >> 2);
        const int pr = P / F2_PW, pc = P - pr * F2_PW;
        const int ih = 2 * (oh0 + pr), iw = 2 * ow0 + pc;
        const bool ok = piece < 33 && P < F2_NPP && iw < p.W;
        xo[k] = ((n * p.H + ih) * p.W + iw) * p.Cin * 2 + ((((lane & 3) ^ ((P >> 1) & 3))) << 4);
        if (ok && ih < p.H) xok |= 1u << k;            // row 2 (oh0 + r) + 0 inside the map
        if (ok && ih + 1 < p.H) xok |= 1u << (8 + k);   // ... + 1
        if (ok && ih + 2 < p.H) xok |= 1u << (16 + k);  // ... + 2
    }
    int wo[3], ws0[3], ws1[3];                     // row part of the source offset; swizzled slot for the low / high 32-channel half
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int piece = wave + 8 * k, tap_w = piece >> 3, row = 16 * (piece & 7) + (lane >> 2);
        const int ls = (lane & 3) ^ ((row >> 2) & 3);                          // logical slot (0..3) inside the 32-channel half
        wo[k] = (tap_w * p.rows_pad + c0 + row) * 128;
        ws0[k] = (ls ^ ((row >> 1) & 7)) << 4;
        ws1[k] = ((4 + ls) ^ ((row >> 1) & 7)) << 4;
    }
    auto issue = [&](int st, int stage) {
        const int c32 = st / 3, kh = st - 3 * c32;
        const int c64 = c32 >> 1, h = c32 & 1;
        const int xs = kh * p.W * p.Cin * 2 + c32 * 64;                        // uniform
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            if (wave + 8 * k < 33) {
                const bool ok = (xok >> (8 * kh + k)) & 1u;
                s2_dma16(rs_x, __builtin_amdgcn_readfirstlane(lds0 + stage * F2_STAGE + (wave + 8 * k) * 1024), ok ? xo[k] + xs : S2_OOB);
            }
        }
        const int wb = (c64 * 9 + kh * 3) * p.rows_pad * 128;                  // uniform: chunk c64, taps kh * 3 ..
#pragma unroll
        for (int k = 0; k < 3; ++k)
            s2_dma16(rs_w, __builtin_amdgcn_readfirstlane(lds0 + stage * F2_STAGE + F2_PATCH + (wave + 8 * k) * 1024), wo[k] + wb + (h ? ws1[k] : ws0[k]));
    };

    // ---- fragment addresses
    int aoff[2];                                   // weight row wave_c * 64 + 32 i + l31 (tap kw: + kw * F2_WT), logical slot 2 kk + g
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = wave_c * 64 + 32 * i + l31;
        aoff[i] = F2_PATCH + row * 64 + ((g ^ ((row >> 2) & 3)) << 4);
    }
    int boff[2][3];                                // output row 2 wave_p + j, column l31, tap kw
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
            const int P = (2 * wave_p + j) * F2_PW + 2 * l31 + kw;
            boff[j][kw] = P * 64 + ((g ^ ((P >> 1) & 3)) << 4);
        }

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    const int n_st = p.n_chunks32 * 3;
    issue(0, 0);
    auto stage_fn = [&](int st, auto stage_c) {
        constexpr int ST = decltype(stage_c)::value;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (st + 1 < n_st) issue(st + 1, ST ^ 1);
        const unsigned char* sb = s2_smem + ST * F2_STAGE;
#pragma unroll
        for (int kw = 0; kw < 3; ++kw)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                bf16x8 afr[2], bfr[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) afr[i] = *reinterpret_cast<const bf16x8*>(sb + ((aoff[i] + kw * F2_WT) ^ (kk << 5)));
#pragma unroll
                for (int j = 0; j < 2; ++j) bfr[j] = *reinterpret_cast<const bf16x8*>(sb + (boff[j][kw] ^ (kk << 5)));
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) mma16(acc[i][j], afr[i], bfr[j]);           // D[cout][pixel]
            }
    };
    for (int st = 0; st < n_st; st += 2) {
        stage_fn(st, std::integral_constant<int, 0>{});
        if (st + 1 < n_st) stage_fn(st + 1, std::integral_constant<int, 1>{});
    }

    // ---- epilogue
    const unsigned out_bytes = (unsigned)((size_t)p.N * p.Ho * p.Wo * p.Cout * 2);
    const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, out_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc(p.bias ? (void*)p.bias : (void*)p.y, 0, p.bias ? (unsigned)(p.Cout * 4) : 0u, 0x00020000);
    f32x4 bv[2][2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int qp = 0; qp < 2; ++qp) {
            const int cb = (c0 + wave_c * 64 + i * 32 + 16 * qp + 8 * g) * 4;
            bv[i][qp][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_b, cb, 0, 0));
            bv[i][qp][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_b, cb + 16, 0, 0));
        }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int oh = oh0 + 2 * wave_p + j, ow = ow0 + l31;
        const int obase = (oh < p.Ho && ow < p.Wo) ? (((n * p.Ho + oh) * p.Wo + ow) * p.Cout + c0 + wave_c * 64 + 8 * g) * 2 : S2_OOB;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int qp = 0; qp < 2; ++qp) {
                float v[8];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float qa = acc[i][j][(2 * qp) * 4 + e], qb = acc[i][j][(2 * qp + 1) * 4 + e];
                    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(qa), __float_as_uint(qb), false, false);
                    v[e] = __uint_as_float(r[0]); v[4 + e] = __uint_as_float(r[1]);
                }
                u32x4 o;
                bf16_t* ob = reinterpret_cast<bf16_t*>(&o);
#pragma unroll
                for (int e = 0; e < 8; ++e) ob[e] = (bf16_t)(v[e] + bv[i][qp][e >> 2][e & 3]);
                __builtin_amdgcn_raw_buffer_store_b128(o, rs_y, obase + (i * 32 + qp * 16) * 2, 0, 0);
            }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// conv_s2_dgrad_kernel -- Downsample's data gradient  dx[n, u, v, ci] = sum over (kh, kw, co) with u - kh, v - kw even of
//   w[co][kh][kw][ci] * dy[n, (u - kh) / 2, (v - kw) / 2, co]
// without the zero-stuffed tensor (round 1: a stride-1 convolution over 4x the pixels, three quarters of its MFMAs multiplying
// zeros).  A work-group owns ONE parity class (p, q) = (u & 1, v & 1) of an 8 x 32 block of (a, b) = (u >> 1, v >> 1): 256 dx pixels
// x 128 cins whose taps are kh in {0, 2} (p = 0) or {1} (p = 1), likewise kw -- 4, 2, 2 or 1 taps, the exact FLOPs.  Structure of the
// forward kernel above: stage = (32-cout chunk, kh): the 8 dy rows a - (kh >> 1) x 33 columns b - 1 .. b + 31 (17 KiB) + the class's
// 1-2 taps of weights from the TRANSPOSED K64 image (rows = cin, taps flipped: mas_pack_conv_weight_layout), double-buffered LDS-DMA.
// Heavy classes are dispatched first.
struct S2DgradParams {
    const unsigned char* dy; const unsigned char* w; unsigned char* dx;
    int N, H, W, Cin, Ho, Wo, Cout, rows_pad, n_chunks32, tiles_h, tiles_w, n_ct, per_class;
};
constexpr int G2_PW = 33, G2_NPP = 8 * G2_PW;   // 264 patch pixels per stage
constexpr int G2_PATCH = 17 * 1024;
constexpr int G2_STAGE = G2_PATCH + 2 * F2_WT;  // 33792 B
constexpr int G2_LDS = 2 * G2_STAGE;

__global__ __launch_bounds__(S2_NT, 1) void conv_s2_dgrad_kernel(S2DgradParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char s2_smem[];
    const unsigned lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) unsigned char*)s2_smem;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 5, l31 = lane & 31;
    const int wave_c = wave & 1, wave_p = wave >> 1;
    const int cls = (int)blockIdx.x / p.per_class;                  // 0: (0,0) 4 taps | 1: (0,1) | 2: (1,0) | 3: (1,1) 1 tap
    const int pp = cls >> 1, qq = cls & 1;
    int t = (int)blockIdx.x - cls * p.per_class;
    const int ct = t % p.n_ct; t /= p.n_ct;
    const int tw_i = t % p.tiles_w; t /= p.tiles_w;
    const int th_i = t % p.tiles_h, n = t / p.tiles_h;
    const int a0 = th_i * F2_TH, b0 = tw_i * F2_TW, ci0 = ct * 128;
    const int nkh = pp ? 1 : 2, nkw = qq ? 1 : 2;

    const s2_i32x4 rs_y = s2_rsrc(p.dy, (unsigned)((size_t)p.N * p.Ho * p.Wo * p.Cout * 2));
    const s2_i32x4 rs_w = s2_rsrc(p.w, (unsigned)((size_t)((p.Cout + 63) / 64) * 9 * p.rows_pad * 128));

    // ---- DMA plan.  Patch piece wave + 8 k (k < 3, piece < 17): pixel P = 16 piece + (lane >> 2) = row P / 33 (dy row a0 + row - dh),
    //      column P % 33 (dy column b0 - 1 + column); slot swizzle as the forward kernel.  Weights: piece wave + 8 k = tap slot k, rows
    //      16 wave + (lane >> 2) of the 128-cin tile.
    int yo[3];
    unsigned yok = 0;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int piece = wave + 8 * k, P = piece * 16 + (lane >> 2);
        const int pr = P / G2_PW, pc = P - pr * G2_PW;
        const int ar = a0 + pr, bc = b0 - 1 + pc;
        const bool ok = piece < 17 && P < G2_NPP && bc >= 0 && bc < p.Wo;
        yo[k] = ((n * p.Ho + ar) * p.Wo + bc) * p.Cout * 2 + ((((lane & 3) ^ ((P >> 1) & 3))) << 4);
        if (ok && ar < p.Ho) yok |= 1u << k;                              // dh = 0: dy row a0 + row
        if (ok && ar >= 1 && ar - 1 < p.Ho) yok |= 1u << (8 + k);         // dh = 1: dy row a0 + row - 1
    }
    const int wrow = 16 * wave + (lane >> 2);
    const int wls = (lane & 3) ^ ((wrow >> 2) & 3);
    const int wo_ = (ci0 + wrow) * 128;
    const int ws0 = (wls ^ ((wrow >> 1) & 7)) << 4, ws1 = ((4 + wls) ^ ((wrow >> 1) & 7)) << 4;
    auto issue = [&](int st, int stage) {
        const int c32 = st / nkh, ikh = st - nkh * c32;
        const int kh = pp ? 1 : 2 * ikh, dh = kh >> 1;
        const int c64 = c32 >> 1, h = c32 & 1;
        const int ys = c32 * 64 - dh * p.Wo * p.Cout * 2;                     // uniform
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            if (wave + 8 * k < 17) {
                const bool ok = (yok >> (8 * dh + k)) & 1u;
                s2_dma16(rs_y, __builtin_amdgcn_readfirstlane(lds0 + stage * G2_STAGE + (wave + 8 * k) * 1024), ok ? yo[k] + ys : S2_OOB);
            }
        }
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            if (k < nkw) {
                const int kw = qq ? 1 : 2 * k;
                const int tap_img = (2 - kh) * 3 + (2 - kw);                   // the transposed image stores the taps flipped
                const int wb = ((c64 * 9 + tap_img) * p.rows_pad) * 128;      // uniform
                s2_dma16(rs_w, __builtin_amdgcn_readfirstlane(lds0 + stage * G2_STAGE + G2_PATCH + (wave + 8 * k) * 1024), wo_ + wb + (h ? ws1 : ws0));
            }
        }
    };

    int aoff[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = wave_c * 64 + 32 * i + l31;
        aoff[i] = G2_PATCH + row * 64 + ((g ^ ((row >> 2) & 3)) << 4);
    }
    int boff[2][2];                                // (a, b) row 2 wave_p + j, column l31; tap slot k: dy column b - (kw >> 1) = patch column l31 + 1 - (kw >> 1)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int kw = qq ? 1 : 2 * k;
            const int P = (2 * wave_p + j) * G2_PW + l31 + 1 - (kw >> 1);
            boff[j][k] = P * 64 + ((g ^ ((P >> 1) & 3)) << 4);
        }

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    const int n_st = p.n_chunks32 * nkh;
    issue(0, 0);
    auto stage_fn = [&](int st, auto stage_c) {
        constexpr int ST = decltype(stage_c)::value;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (st + 1 < n_st) issue(st + 1, ST ^ 1);
        const unsigned char* sb = s2_smem + ST * G2_STAGE;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            if (k < nkw) {
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    bf16x8 afr[2], bfr[2];
#pragma unroll
                    for (int i = 0; i < 2; ++i) afr[i] = *reinterpret_cast<const bf16x8*>(sb + ((aoff[i] + k * F2_WT) ^ (kk << 5)));
#pragma unroll
                    for (int j = 0; j < 2; ++j) bfr[j] = *reinterpret_cast<const bf16x8*>(sb + (boff[j][k] ^ (kk << 5)));
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j) mma16(acc[i][j], afr[i], bfr[j]);       // D[cin][pixel]
                }
            }
        }
    };
    for (int st = 0; st < n_st; st += 2) {
        stage_fn(st, std::integral_constant<int, 0>{});
        if (st + 1 < n_st) stage_fn(st + 1, std::integral_constant<int, 1>{});
    }

    const unsigned out_bytes = (unsigned)((size_t)p.N * p.H * p.W * p.Cin * 2);
    const __amdgpu_buffer_rsrc_t rs_o = __builtin_amdgcn_make_buffer_rsrc(p.dx, 0, out_bytes, 0x00020000);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int u = 2 * (a0 + 2 * wave_p + j) + pp, v = 2 * (b0 + l31) + qq;
        const int obase = (u < p.H && v < p.W) ? (((n * p.H + u) * p.W + v) * p.Cin + ci0 + wave_c * 64 + 8 * g) * 2 : S2_OOB;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int qp = 0; qp < 2; ++qp) {
                float vv[8];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float qa = acc[i][j][(2 * qp) * 4 + e], qb = acc[i][j][(2 * qp + 1) * 4 + e];
                    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(qa), __float_as_uint(qb), false, false);
                    vv[e] = __uint_as_float(r[0]); vv[4 + e] = __uint_as_float(r[1]);
                }
                u32x4 o;
                bf16_t* ob = reinterpret_cast<bf16_t*>(&o);
#pragma unroll
                for (int e = 0; e < 8; ++e) ob[e] = (bf16_t)vv[e];
                __builtin_amdgcn_raw_buffer_store_b128(o, rs_o, obase + (i * 32 + qp * 16) * 2, 0, 0);
            }
    }
}

}  // namespace

static bool s2_wgrad_setup(const MasConvDesc* d, S2WgradParams& p) {
    static const int on = mas_env_int("MAS_CONV_S2", 1);
    if (!on) return false;
    if (d->ks != 3 || d->stride != 2 || d->upsample || d->act != MAS_ACT_NONE || d->pad_top != 0 || d->pad_left != 0) return false;
    if (d->in_dtype != MAS_BF16 || d->Cin % 64 || d->Cout % 128) return false;
    if ((long long)d->N * d->H * d->W * d->Cin * 2 >= 0x7fffffffLL || (long long)d->N * d->Ho * d->Wo * d->Cout * 2 >= 0x7fffffffLL) return false;
    p.N = d->N; p.H = d->H; p.W = d->W; p.Cin = d->Cin; p.Ho = d->Ho; p.Wo = d->Wo; p.Cout = d->Cout;
    p.tiles_h = mas_cdiv(d->Ho, S2_TH); p.tiles_w = mas_cdiv(d->Wo, S2_TW); p.n_tiles = d->N * p.tiles_h * p.tiles_w;
    p.n_co_t = d->Cout / 128; p.n_ci_t = d->Cin / 64;
    int ns = mas_cdiv(mas_num_cus(), p.n_co_t * p.n_ci_t);                      // one 108-KiB work-group per CU
    if (ns > p.n_tiles) ns = p.n_tiles;
    if (ns < 1) ns = 1;
    p.nsplit = ns;
    return true;
}

int mas_wgrad_s2_splits(const MasConvDesc* d) {
    S2WgradParams p;
    return s2_wgrad_setup(d, p) ? p.nsplit : 0;
}

// part [nsplit][Cout][3][3][Cin], part_bias [nsplit][Cout] or NULL (see mas_conv_wgrad_partial)
int mas_wgrad_s2_partial(const MasConvDesc* d, const void* x, const void* dy, float* part, float* part_bias, hipStream_t s) {
    S2WgradParams p;
    if (!s2_wgrad_setup(d, p)) return 0;
    p.x = (const unsigned char*)x; p.dy = (const unsigned char*)dy; p.part = part; p.part_bias = part_bias;
    static mas_devmask_t attr{0};               // the dynamic-LDS limit is a per-DEVICE attribute (mas_common.h)
    unsigned long long attr_bit;
    if (mas_attr_needed(attr, &attr_bit)) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_s2_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, S2_LDS) != hipSuccess) { (void)hipGetLastError(); return 0; }
        mas_attr_done(attr, attr_bit);
    }
    hipLaunchKernelGGL(wgrad_s2_kernel, dim3((unsigned)(p.n_co_t * p.n_ci_t * p.nsplit)), dim3(S2_NT), S2_LDS, s, p);
    MAS_CHECK_LAUNCH("wgrad_s2");
    return 1;
}

// Downsample forward (3x3, stride 2, no padding but the one-sided zero row / column): returns 1 if launched, 0 if conv_fwd.hip should take it
int mas_conv_s2_fwd_try(const MasConvDesc* d, const void* x, const void* w_packed, const float* bias, const void* residual, void* y, hipStream_t s) {
    static const int on = mas_env_int("MAS_CONV_S2", 1);
    if (!on || residual) return 0;
    if (d->ks != 3 || d->stride != 2 || d->upsample || d->act != MAS_ACT_NONE || d->pad_top != 0 || d->pad_left != 0) return 0;
    if (d->in_dtype != MAS_BF16 || d->out_dtype != MAS_BF16 || d->w_layout != MAS_WLAYOUT_K64) return 0;
    // (16 <= Wo < 32 idles half of every 8 x 32 tile and still beats conv_fwd.hip's stride-2 instance 4x: the 32 -> 16 Downsample of
    //  VQ-IMG, 256 -> 256 at batch 32, took 103 us there -- round 4)
    if (d->Cin % 32 || d->Cout % 128 || d->Wo < 16) return 0;
    if ((long long)d->N * d->H * d->W * d->Cin * 2 >= 0x7fffffffLL || (long long)d->N * d->Ho * d->Wo * d->Cout * 2 >= 0x7fffffffLL) return 0;
    S2FwdParams p;
    p.x = (const unsigned char*)x; p.w = (const unsigned char*)w_packed; p.bias = bias; p.y = (unsigned char*)y;
    p.N = d->N; p.H = d->H; p.W = d->W; p.Cin = d->Cin; p.Ho = d->Ho; p.Wo = d->Wo; p.Cout = d->Cout;
    p.rows_pad = mas_roundup(d->Cout, 128); p.n_chunks32 = d->Cin / 32;
    p.tiles_h = mas_cdiv(d->Ho, F2_TH); p.tiles_w = mas_cdiv(d->Wo, F2_TW); p.n_ct = d->Cout / 128;
    const long long grid = (long long)d->N * p.tiles_h * p.tiles_w * p.n_ct;
    if (grid > 0x7fffffffLL) return 0;
    static mas_devmask_t attr{0};               // the dynamic-LDS limit is a per-DEVICE attribute (mas_common.h)
    unsigned long long attr_bit;
    if (mas_attr_needed(attr, &attr_bit)) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(conv_s2_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, F2_LDS) != hipSuccess) { (void)hipGetLastError(); return 0; }
        mas_attr_done(attr, attr_bit);
    }
    hipLaunchKernelGGL(conv_s2_fwd_kernel, dim3((unsigned)grid), dim3(S2_NT), F2_LDS, s, p);
    MAS_CHECK_LAUNCH("conv_s2_fwd");
    return 1;
}

static bool s2_dgrad_ok(const MasConvDesc* d) {
    static const int on = mas_env_int("MAS_CONV_S2", 1);
    if (!on || !d) return false;
    if (d->ks != 3 || d->stride != 2 || d->upsample || d->act != MAS_ACT_NONE || d->pad_top != 0 || d->pad_left != 0) return false;
    if (d->in_dtype != MAS_BF16 || d->out_dtype != MAS_BF16) return false;
    if (d->Cout % 32 || d->Cin % 128 || d->W < 32) return false;                 // (a, b) blocks of 8 x 32: W < 64 idles half of every tile (still faster
                                                                                 //  than the zero-stuffed stride-1 path); narrower maps stay there
    if ((long long)d->N * d->H * d->W * d->Cin * 2 >= 0x7fffffffLL || (long long)d->N * d->Ho * d->Wo * d->Cout * 2 >= 0x7fffffffLL) return false;
    return true;
}

// d describes the FORWARD convolution (Downsample: 3x3, stride 2, pads 0, Ho x Wo its output).  1 if mas_conv_s2_dgrad takes it.
extern "C" int mas_conv_s2_dgrad_supported(const MasConvDesc* d) { return s2_dgrad_ok(d) ? 1 : 0; }

// dx [N, H, W, Cin] = data gradient of that convolution from dy [N, Ho, Wo, Cout] and the TRANSPOSED K64 weight image
// (mas_pack_conv_weight_layout(..., transpose = 1, MAS_WLAYOUT_K64)): every dx element written exactly once, no zero-stuffed tensor.
extern "C" int mas_conv_s2_dgrad(const MasConvDesc* d, const void* dy, const void* w_packed_t, void* dx, void* stream) {
    MAS_ENTER();
    if (!d || !dy || !w_packed_t || !dx) MAS_FAIL(MAS_EINVAL, "conv_s2_dgrad: null argument");
    if (!s2_dgrad_ok(d)) MAS_FAIL(MAS_EUNSUPPORTED, "conv_s2_dgrad: unsupported convolution (mas_conv_s2_dgrad_supported == 0)");
    S2DgradParams p;
    p.dy = (const unsigned char*)dy; p.w = (const unsigned char*)w_packed_t; p.dx = (unsigned char*)dx;
    p.N = d->N; p.H = d->H; p.W = d->W; p.Cin = d->Cin; p.Ho = d->Ho; p.Wo = d->Wo; p.Cout = d->Cout;
    p.rows_pad = mas_roundup(d->Cin, 128); p.n_chunks32 = d->Cout / 32;
    p.tiles_h = mas_cdiv((d->H + 1) / 2, F2_TH); p.tiles_w = mas_cdiv((d->W + 1) / 2, F2_TW); p.n_ct = d->Cin / 128;
    const long long per_class = (long long)d->N * p.tiles_h * p.tiles_w * p.n_ct;
    if (4 * per_class > 0x7fffffffLL) MAS_FAIL(MAS_EUNSUPPORTED, "conv_s2_dgrad: grid too large");
    p.per_class = (int)per_class;
    static mas_devmask_t attr{0};
    unsigned long long attr_bit;
    if (mas_attr_needed(attr, &attr_bit)) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(conv_s2_dgrad_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, G2_LDS) != hipSuccess)
            MAS_FAIL(MAS_ELAUNCH, "conv_s2_dgrad: cannot reserve %d bytes of LDS", G2_LDS);
        mas_attr_done(attr, attr_bit);
    }
    hipLaunchKernelGGL(conv_s2_dgrad_kernel, dim3((unsigned)(4 * per_class)), dim3(S2_NT), G2_LDS, reinterpret_cast<hipStream_t>(stream), p);
    MAS_CHECK_LAUNCH("conv_s2_dgrad");
    return MAS_OK;
}
