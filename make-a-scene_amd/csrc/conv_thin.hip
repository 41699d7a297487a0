// The RGB-edge layers of the VQ models: conv_in (3 -> 128, reference models/modules.py:219) and conv_out (128 -> 3, :345), whose 3-channel
// side is zero-padded to one 16-byte slot (8 channels) by ops.norm_act_conv.  Their arithmetic is negligible (2 x 9 x 8 x 128 FLOP per
// pixel) but each streams a 128-channel 256^2 tensor, and the general kernels treat the 8-channel side as a full 64-channel chunk /
// 128-cout tile: 8-16x the MFMA work, 0.32-0.59 ms per launch where the tensor's HBM time is 0.1 ms (profiles/r03_conv_shapes.txt).
//
// wgrad_thin_kernel -- both weight gradients as ONE GEMM shape:  D[(tap, cs)][cb] = sum over pixels of S[pixel + sgn (tap - 1)][cs] * B[pixel][cb]
//   B = the 128-channel tensor (conv_out: the activated input, conv_in: dy), S = the 8-channel one (conv_out: dy, conv_in: x), sgn = -1 / +1.
//   K = pixels (outer dimension of both operands): fragments by the LDS transpose read (ds_read_b64_tr_b16, semantics in conv_wgrad.hip).
//   The trick for the thin side: a transpose-read source lane supplies 4 consecutive channels of ONE pixel at an ARBITRARY address, so the
//   72 rows (tap, cs) of the A operand are read straight out of an (4+2) x (16+2)-pixel halo tile of S -- row quad q of a 16-lane group
//   points at tap 4 i + 2 G16 + (q >> 1), channels 4 (q & 1) ..+3, shifted by the tap: no im2col, no padding of S to 128 channels.
//   Tile = 4 x 16 pixels of B (16 KiB, LDS-DMA, 64-byte blocks ^ (pixel & 3)), 4 waves = the 4 32-channel groups of B, 3 accumulator
//   tiles (96 rows, 72 used) per wave; split-K over tiles into slabs that mas_wgrad_reduce adds in a fixed order; the bias gradient is
//   one more MFMA per k-step against an all-ones operand.  HBM-bound by construction (12 MFMAs per 16 KiB of B).
#include "mas_common.h"
#include <type_traits>

namespace {

typedef __attribute__((ext_vector_type(4))) int th_i32x4;
typedef __attribute__((ext_vector_type(4))) short th_s16x4;
__device__ __forceinline__ void th_dma16(th_i32x4 rs, unsigned lds, int vo) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" :: "s"(lds), "v"(vo), "s"(rs) : "memory", "m0");
}
__device__ __forceinline__ th_i32x4 th_rsrc(const void* base, unsigned bytes) {
    const unsigned long long a = (unsigned long long)base;
    th_i32x4 r = {(int)(unsigned)a, (int)(unsigned)(a >> 32), (int)bytes, 0x00020000};
    r[0] = __builtin_amdgcn_readfirstlane(r[0]); r[1] = __builtin_amdgcn_readfirstlane(r[1]);
    r[2] = __builtin_amdgcn_readfirstlane(r[2]); r[3] = __builtin_amdgcn_readfirstlane(r[3]);
    return r;
}
__device__ __forceinline__ bf16x8 th_tr(const unsigned char* a0, const unsigned char* a1) {
    const th_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) th_s16x4*)a0);
    const th_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) th_s16x4*)a1);
    const __attribute__((ext_vector_type(8))) short v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return *reinterpret_cast<const bf16x8*>(&v);
}

struct ThinWgradParams {
    const unsigned char* big; const unsigned char* small; float* part; float* part_bias;
    int N, H, W, sgn;                          // S pixel = B pixel + sgn * (tap - 1)
    int col_stride, tap_stride, cs_stride;     // slab index of D[(tap, cs)][cb] = cb * col_stride + tap * tap_stride + cs * cs_stride
    int bias_small;                            // the bias gradient sums S (conv_out: dy is the thin tensor) / B (conv_in)
    int tiles_h, tiles_w, n_tiles, nsplit;
};

constexpr int TH_NT = 256, TH_TH = 4, TH_TW = 16;
constexpr int TH_BIG = 64 * 256;               // 64 pixels x 128 channels
constexpr int TH_SMALL = 2048;                 // (4 + 2) x (16 + 2) = 108 pixels x 16 B in two 1-KiB DMA pieces
constexpr int TH_STAGE = TH_BIG + TH_SMALL;
constexpr int TH_LDS = 2 * TH_STAGE;
constexpr int TH_OOB = (int)0x80000000;
constexpr int TH_SLAB = 9 * 8 * 128;

__global__ __launch_bounds__(TH_NT, 2) void wgrad_thin_kernel(ThinWgradParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char th_smem[];
    const unsigned lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) unsigned char*)th_smem;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 5, l31 = lane & 31, G16 = (lane >> 4) & 1, sl = lane & 15;
    const int split = blockIdx.x;
    const th_i32x4 rs_b = th_rsrc(p.big, (unsigned)((size_t)p.N * p.H * p.W * 256));
    const th_i32x4 rs_s = th_rsrc(p.small, (unsigned)((size_t)p.N * p.H * p.W * 16));

    // ---- DMA plan.  B: 16 pieces of 4 pixels x 256 B, wave w moves pieces 4 w .. 4 w + 3 = tile row w; piece 4 w + j holds columns
    //      4 j .. 4 j + 3; lane: column 4 j + (lane >> 4), physical 64-byte block (lane >> 2) & 3 = logical block ^ (column & 3).
    //      S: waves 0 / 1 move halo pixels 64 wave + lane (108 of 128 slots live), 16 B each, unswizzled.
    const int lp = lane >> 4;
    const int lsrc = ((((lane >> 2) & 3) ^ lp) << 6) + ((lane & 3) << 4);
    const int hp = wave * 64 + lane, hr = hp / 18, hc = hp - hr * 18;            // (waves 0, 1 only)
    auto issue = [&](int t, int stage) {
        const int tw_i = t % p.tiles_w; const int q = t / p.tiles_w;
        const int th_i = q % p.tiles_h, n = q / p.tiles_h;
        const int h0 = th_i * TH_TH, w0 = tw_i * TH_TW;
        const int ih = h0 + wave;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int iw = w0 + 4 * j + lp;
            const bool ok = ih < p.H && iw < p.W;
            th_dma16(rs_b, __builtin_amdgcn_readfirstlane(lds0 + stage * TH_STAGE + (wave * 4 + j) * 1024), ok ? ((n * p.H + ih) * p.W + iw) * 256 + lsrc : TH_OOB);
        }
        if (wave < 2) {
            const int sh = h0 + hr - 1, sw = w0 + hc - 1;
            const bool ok = hp < 108 && sh >= 0 && sh < p.H && sw >= 0 && sw < p.W;
            th_dma16(rs_s, __builtin_amdgcn_readfirstlane(lds0 + stage * TH_STAGE + TH_BIG + wave * 1024), ok ? ((n * p.H + sh) * p.W + sw) * 16 : TH_OOB);
        }
    };

    // ---- fragment addressing.  A (rows (tap, cs) of row tile i): source lane -> pixel column 8 g + (sl >> 2) (+ 4 for the second read) of
    //      tile row r, row quad q = sl & 3 -> tap 4 i + 2 G16 + (q >> 1), channels 4 (q & 1) ..+3; halo pixel (r + 1 + sgn (kh - 1),
    //      column + 1 + sgn (kw - 1)).  Taps >= 9 (rows 72..95) read tap 8's data; their rows are never stored.
    const int t4 = sl >> 2, qd = sl & 3;
    int a_off[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        int tap = 4 * i + 2 * G16 + (qd >> 1);
        if (tap > 8) tap = 8;
        const int kh = tap / 3, kw = tap - 3 * kh;
        a_off[i] = TH_BIG + ((1 + p.sgn * (kh - 1)) * 18 + (8 * g + t4 + 1 + p.sgn * (kw - 1))) * 16 + (qd & 1) * 8;
    }
    // B (columns = this wave's 32-channel group): pixel 16 r + 8 g + t4 (+ 4), channels 16 G16 + 4 (sl & 3) ..+3, block wave ^ t4
    const int b_off = (8 * g + t4) * 256 + ((wave ^ t4) << 6) + 32 * G16 + 8 * (sl & 3);

    f32x16 acc[3], accb;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[0][r] = 0.0f; acc[1][r] = 0.0f; acc[2][r] = 0.0f; accb[r] = 0.0f; }
    bf16x8 ones;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = (bf16_t)1.0f;
    const bool bias_s = p.part_bias != nullptr && p.bias_small && wave == 0;     // (wave-uniform)
    const bool bias_b = p.part_bias != nullptr && !p.bias_small;

    const int n_mine = (p.n_tiles - split + p.nsplit - 1) / p.nsplit;           // tiles split, split + nsplit, ...
    if (n_mine > 0) issue(split, 0);
    auto tile = [&](int it, auto stage_c) {
        constexpr int ST = decltype(stage_c)::value;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (it + 1 < n_mine) issue(split + (it + 1) * p.nsplit, ST ^ 1);
        const unsigned char* sb = th_smem + ST * TH_STAGE;
#pragma unroll
        for (int r = 0; r < 4; ++r) {                                            // k-step = tile row r (16 pixels)
            const unsigned char* b0 = sb + b_off + r * 16 * 256;
            const bf16x8 bfr = th_tr(b0, b0 + 4 * 256);
            bf16x8 afr[3];
#pragma unroll
            for (int i = 0; i < 3; ++i) { const unsigned char* a0 = sb + a_off[i] + r * 18 * 16; afr[i] = th_tr(a0, a0 + 4 * 16); }
#pragma unroll
            for (int i = 0; i < 3; ++i) mma16(acc[i], afr[i], bfr);
            if (bias_s) mma16(accb, afr[1], ones);                               // rows 32..39 = centre tap: sum over the tile of S[pixel][cs]
            if (bias_b) mma16(accb, ones, bfr);                                  // every row = sum over the tile of B[pixel][cb]
        }
    };
    for (int it = 0; it < n_mine; it += 2) {
        tile(it, std::integral_constant<int, 0>{});
        if (it + 1 < n_mine) tile(it + 1, std::integral_constant<int, 1>{});
    }

    float* pw = p.part + (size_t)split * TH_SLAB;
    const int cb = wave * 32 + l31;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = 32 * i + acc_row(lane, r);
            const int tap = row >> 3, cs = row & 7;
            if (tap < 9) pw[cb * p.col_stride + tap * p.tap_stride + cs * p.cs_stride] = acc[i][r];
        }
    if (bias_s && l31 == 0) {                                                    // rows 0..7 of row tile 1: lanes 0 (rows 0-3) and 32 (rows 4-7)
#pragma unroll
        for (int r = 0; r < 4; ++r) p.part_bias[(size_t)split * 8 + 4 * g + r] = accb[r];
    }
    if (bias_b && g == 0) p.part_bias[(size_t)split * 128 + cb] = accb[0];     // row 0
}

// ---------------------------------------------------------------------------------------------------------------------------------
// conv_thin_fwd_kernel -- 3x3 / stride 1 / pad 1, 8 -> 128 channels (conv_in's forward, conv_out's data gradient):
//   y[pixel][co] = sum over (tap, ci) of W[co][tap][ci] * x[pixel + tap - 1][ci] + bias[co]
// One GEMM with K = 9 taps x 8 channels = 72 (5 k-steps of 16, the tenth tap zero): the A operand (all 128 couts x 80 k = 20 fragments,
// 80 VGPRs) is read ONCE per wave from the K64 weight image conv_fwd.hip uses (its 64-channel rows hold 8 live channels in logical slot
// 0) and stays in registers; a B fragment is one 16-byte LDS read of the (8 + 2) x (32 + 2)-pixel halo tile of x (half-wave g takes tap
// 2 kk + g).  20 MFMAs per 32 pixels against 9 x 64-channel chunks x 4 cout tiles = 144 in conv_fwd.hip.  Store-bound: the kernel's
// job is to write 256 B per pixel.  Persistent work-groups, halo tiles double-buffered by LDS-DMA.
struct ThinFwdParams {
    const unsigned char* x; const unsigned char* w; const float* bias; unsigned char* y;
    int N, H, W, rows_pad, tiles_h, tiles_w, n_tiles;
};
constexpr int TF_TH = 8, TF_TW = 32, TF_HW = TF_TW + 2, TF_HALO = (TF_TH + 2) * TF_HW;   // 340 pixels x 16 B
constexpr int TF_STAGE = 6 * 1024;             // 6 DMA pieces of 64 pixels
constexpr int TF_LDS = 2 * TF_STAGE;

__global__ __launch_bounds__(TH_NT, 2) void conv_thin_fwd_kernel(ThinFwdParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char th_smem[];
    const unsigned lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) unsigned char*)th_smem;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 5, l31 = lane & 31;
    const th_i32x4 rs_x = th_rsrc(p.x, (unsigned)((size_t)p.N * p.H * p.W * 16));
    const unsigned out_bytes = (unsigned)((size_t)p.N * p.H * p.W * 256);
    const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, out_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc(p.bias ? (void*)p.bias : (void*)p.y, 0, p.bias ? 512u : 0u, 0x00020000);

    // ---- weights: fragment (cout tile i, k-step kk): row 32 i + l31, tap 2 kk + g, channels 0..7 = logical slot 0 of the image row
    bf16x8 afr[4][5];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int kk = 0; kk < 5; ++kk) {
            const int row = 32 * i + l31, tap = 2 * kk + g;
            u32x4 v = {0u, 0u, 0u, 0u};
            if (tap < 9) v = *reinterpret_cast<const u32x4*>(p.w + ((size_t)tap * p.rows_pad + row) * 128 + (((row >> 1) & 7) << 4));
            afr[i][kk] = *reinterpret_cast<const bf16x8*>(&v);
        }
    // ---- B fragment addresses: pixel (tile row 2 wave + j, column l31), tap 2 kk + g -> halo pixel (row + kh, column + kw)
    int b_off[5];
#pragma unroll
    for (int kk = 0; kk < 5; ++kk) {
        int tap = 2 * kk + g;
        if (tap > 8) tap = 8;                                                    // (multiplied by zero weights)
        const int kh = tap / 3, kw = tap - 3 * kh;
        b_off[kk] = ((2 * wave + kh) * TF_HW + l31 + kw) * 16;
    }
    auto issue = [&](int t, int stage) {
        const int tw_i = t % p.tiles_w; const int q = t / p.tiles_w;
        const int th_i = q % p.tiles_h, n = q / p.tiles_h;
        const int h0 = th_i * TF_TH - 1, w0 = tw_i * TF_TW - 1;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int piece = wave + 4 * j;
            if (piece < 6) {
                const int hp = piece * 64 + lane, hr = hp / TF_HW, hc = hp - hr * TF_HW;
                const int ih = h0 + hr, iw = w0 + hc;
                const bool ok = hp < TF_HALO && ih >= 0 && ih < p.H && iw >= 0 && iw < p.W;
                th_dma16(rs_x, __builtin_amdgcn_readfirstlane(lds0 + stage * TF_STAGE + piece * 1024), ok ? ((n * p.H + ih) * p.W + iw) * 16 : TH_OOB);
            }
        }
    };
    const int first = blockIdx.x, stride = gridDim.x;
    if (first < p.n_tiles) issue(first, 0);
    auto tile = [&](int t, auto stage_c) {
        constexpr int ST = decltype(stage_c)::value;
        // this tile's halo pieces have landed; the 16 stores of the previous tile are YOUNGER than them and stay in flight (in-order retirement)
        if (t == first) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");                 // (2 rows x 8 row-wise stores)
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (t + stride < p.n_tiles) issue(t + stride, ST ^ 1);
        const unsigned char* sb = th_smem + ST * TF_STAGE;
        const int tw_i = t % p.tiles_w; const int q = t / p.tiles_w;
        const int th_i = q % p.tiles_h, n = q / p.tiles_h;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            f32x16 acc[4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;
#pragma unroll
            for (int kk = 0; kk < 5; ++kk) {
                const bf16x8 bfr = *reinterpret_cast<const bf16x8*>(sb + b_off[kk] + j * TF_HW * 16);
#pragma unroll
                for (int i = 0; i < 4; ++i) mma16(acc[i], afr[i][kk], bfr);      // D[cout][pixel]
            }
            // epilogue through a wave-private LDS tile [32 pixels][256 B] (16-byte slot ^ (pixel & 15)): the accumulators hold, per lane, 16-byte
            // pieces of 32 DIFFERENT pixel rows -- stored directly (first version) the kernel ran at 3.1 TB/s; read back row-wise, one store
            // instruction writes four whole 256-byte rows
            unsigned char* ot = th_smem + TF_LDS + wave * 8192;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int qp = 0; qp < 2; ++qp) {
                    const int cb = (32 * i + 16 * qp + 8 * g) * 4;
                    const f32x4 b0 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_b, cb, 0, 0));
                    const f32x4 b1 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_b, cb + 16, 0, 0));
                    float v[8];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float qa = acc[i][(2 * qp) * 4 + e], qb = acc[i][(2 * qp + 1) * 4 + e];
                        const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(qa), __float_as_uint(qb), false, false);
                        v[e] = __uint_as_float(r[0]); v[4 + e] = __uint_as_float(r[1]);
                    }
                    u32x4 o;
                    bf16_t* ob = reinterpret_cast<bf16_t*>(&o);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { ob[e] = (bf16_t)(v[e] + b0[e]); ob[4 + e] = (bf16_t)(v[4 + e] + b1[e]); }
                    *reinterpret_cast<u32x4*>(ot + l31 * 256 + (((4 * i + 2 * qp + g) ^ (l31 & 15)) << 4)) = o;
                }
            const int oh = th_i * TF_TH + 2 * wave + j;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int r = 4 * k + (lane >> 4), sl_ = lane & 15;
                const u32x4 o = *reinterpret_cast<const u32x4*>(ot + r * 256 + ((sl_ ^ (r & 15)) << 4));
                const int ow = tw_i * TF_TW + r;
                const int off = (oh < p.H && ow < p.W) ? ((n * p.H + oh) * p.W + ow) * 256 + sl_ * 16 : TH_OOB;
                __builtin_amdgcn_raw_buffer_store_b128(o, rs_y, off, 0, 0);
            }
        }
    };
    for (int t = first; t < p.n_tiles; t += 2 * stride) {
        tile(t, std::integral_constant<int, 0>{});
        if (t + stride < p.n_tiles) tile(t + stride, std::integral_constant<int, 1>{});
    }
}


// ---------------------------------------------------------------------------------------------------------------------------------
// conv_thin_out_kernel (round 5) -- 3x3 / stride 1 / pad 1, C -> 8 channels (conv_out's forward, reference models/modules.py:364 with
// its 3 output channels padded to one 16-byte slot by ops.norm_act_conv):
//   y[pixel][co] = sum over (tap, ci) of W[co][tap][ci] * x[pixel + tap - 1][ci] + bias[co]
// The general kernel ran it on 32-cout tiles against 64-channel chunks (0.26 ms at 128 channels @256^2 x 32 where the input's HBM time is
// 0.09 ms).  Here: the wide kernel's patch machinery at half height -- 8 x 32-pixel tiles, 32-channel chunks of the (8 + 2) x (32 + 2)
// halo patch double-buffered by LDS-DMA (64-byte rows, 16-byte slot ^ ((pixel >> 2) & 3)) -- a wave owns two tile rows (two 32 x 32
// accumulator blocks: rows = pixels, columns = couts, 8 of 32 live), the weights of all chunks sit in LDS for the life of the
// work-group ([chunk][tap][8 couts][64 B], copied once out of the K64 image conv_fwd.hip uses), 36 MFMAs per chunk and wave.  The
// accumulators reach memory through a wave-private LDS tile ([pixel][8 couts]) so that a store instruction writes whole pixels.
// HBM-bound by construction; persistent work-groups.
struct ThinOutParams {
    const unsigned char* x; const unsigned char* w; const float* bias; unsigned char* y;
    int N, H, W, Cin, n_chunks, rows_pad, tiles_h, tiles_w, n_tiles;
};
constexpr int TO_TH = 8, TO_TW = 32, TO_PW = TO_TW + 2, TO_NPIX = (TO_TH + 2) * TO_PW;       // 340 patch pixels
constexpr int TO_NPIECE = 22;                  // 340 pixels x 64 B -> 22 DMA pieces of 1 KiB (16 pixels each)
constexpr int TO_PATCH = TO_NPIECE * 1024;
constexpr int TO_MAXCHUNKS = 4;                // Cin <= 128: 70 KiB of LDS, two work-groups per CU (44 KiB of patch in flight per CU)
constexpr int TO_W = 0;                        // LDS map: weights [n_chunks][9][8][64 B], then 2 patches, then 4 waves x 2 rows x [32 pixels][8 x 4 B]
constexpr int TO_P = TO_MAXCHUNKS * 9 * 512;
constexpr int TO_O = TO_P + 2 * TO_PATCH;
constexpr int TO_LDS = TO_O + 4 * 2 * 1024;

template <typename OutT>
__global__ __launch_bounds__(TH_NT, 2) void conv_thin_out_kernel(ThinOutParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char th_smem[];
    const unsigned lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) unsigned char*)th_smem;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 5, l31 = lane & 31;
    const th_i32x4 rs_x = th_rsrc(p.x, (unsigned)((size_t)p.N * p.H * p.W * p.Cin * 2));

    // ---- weights -> LDS, once: slot (chunk32 q, tap, cout r < 8, logical 16-byte slot ls < 4) from the K64 image
    //      [chunk64][tap][rows_pad][128 B], physical slot = logical ^ ((row >> 1) & 7)
    for (int u = tid; u < p.n_chunks * 9 * 8 * 4; u += TH_NT) {
        const int ls = u & 3, r = (u >> 2) & 7, qt = u >> 5, tap = qt % 9, q = qt / 9;
        const int ls64 = (q & 1) * 4 + ls;
        const u32x4 v = *reinterpret_cast<const u32x4*>(p.w + ((size_t)((q >> 1) * 9 + tap) * p.rows_pad + r) * 128 + ((ls64 ^ ((r >> 1) & 7)) << 4));
        *reinterpret_cast<u32x4*>(th_smem + TO_W + ((q * 9 + tap) * 8 + r) * 64 + ls * 16) = v;
    }
    float bias = 0.0f;
    if (p.bias && l31 < 8) bias = p.bias[l31];

    // ---- patch DMA: wave w moves pieces w, w + 4, ... < 22; lane -> patch pixel 16 piece + (lane >> 2), physical slot lane & 3 (the swizzle
    //      goes on the SOURCE slot); pixels outside the image (the zero padding) and the dead tail of the last piece read as zeros
    auto issue = [&](int t, int chunk, int buf) {
        const int tw_i = t % p.tiles_w; const int qq = t / p.tiles_w;
        const int th_i = qq % p.tiles_h, n = qq / p.tiles_h;
        const int h0 = th_i * TO_TH - 1, w0 = tw_i * TO_TW - 1;
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const int piece = wave + 4 * k;
            if (piece < TO_NPIECE) {
                const int q = piece * 16 + (lane >> 2), pr = q / TO_PW, pc = q - pr * TO_PW;
                const int ih = h0 + pr, iw = w0 + pc;
                const bool ok = q < TO_NPIX && ih >= 0 && ih < p.H && iw >= 0 && iw < p.W;
                const int sl = (lane & 3) ^ ((q >> 2) & 3);
                th_dma16(rs_x, __builtin_amdgcn_readfirstlane(lds0 + TO_P + buf * TO_PATCH + piece * 1024),
                         ok ? (((n * p.H + ih) * p.W + iw) * p.Cin + chunk * 32 + sl * 8) * 2 : TH_OOB);
            }
        }
    };
    // ---- fragment addresses.  Pixel operand: patch pixel P = (2 wave + j + kh) * 34 + l31 + kw, logical slot 2 kk + g (conv3x3_wide.hip);
    //      weight operand: cout l31 & 7 (columns 8..31 are multiplied into accumulator columns nobody stores), slot 2 kk + g
    auto b_addr = [&](int P) { return P * 64 + (((g ^ (P >> 2)) & 3) << 4); };
    const int pj0 = (2 * wave) * TO_PW + l31;
    const int w_off = (l31 & 7) * 64 + g * 16;

    const int first = blockIdx.x, stride = gridDim.x;
    if (first < p.n_tiles) issue(first, 0, 0);
    int buf = 0;
    for (int t = first; t < p.n_tiles; t += stride) {
        f32x16 acc[2];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.0f;
        for (int c = 0; c < p.n_chunks; ++c) {
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");          // this wave's pieces of the chunk have landed (and the weights are written)
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (c + 1 < p.n_chunks) issue(t, c + 1, buf ^ 1);
            else if (t + stride < p.n_tiles) issue(t + stride, 0, buf ^ 1);
            const unsigned char* pb = th_smem + TO_P + buf * TO_PATCH;
            const unsigned char* wb = th_smem + TO_W + c * (9 * 512) + w_off;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int kh = tap / 3, kw = tap - 3 * kh;
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    const bf16x8 wf = *reinterpret_cast<const bf16x8*>(wb + tap * 512 + kk * 32);
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const bf16x8 pf = *reinterpret_cast<const bf16x8*>(pb + (b_addr(pj0 + (j + kh) * TO_PW + kw) ^ (kk << 5)));
                        mma16(acc[j], pf, wf);                                   // D[pixel][cout]
                    }
                }
            }
            buf ^= 1;
        }
        // ---- epilogue: lanes l31 < 8 hold cout l31 of 16 pixels each -> wave-private LDS tile [row j][pixel][8 x fp32] -> whole pixels to memory
        const int tw_i = t % p.tiles_w; const int qq = t / p.tiles_w;
        const int th_i = qq % p.tiles_h, n = qq / p.tiles_h;
        float* ot = reinterpret_cast<float*>(th_smem + TO_O + wave * 2048);
        if (l31 < 8) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) ot[(j * 32 + acc_row(lane, r)) * 8 + l31] = acc[j][r] + bias;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                        // (LDS operations of one wave complete in order)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int oh = th_i * TO_TH + 2 * wave + j;
            if constexpr (sizeof(OutT) == 4) {                                   // 32 B per pixel: lane -> (pixel lane >> 1, half lane & 1)
                const int px = lane >> 1, ow = tw_i * TO_TW + px;
                const f32x4 v = *reinterpret_cast<const f32x4*>(ot + (j * 32 + px) * 8 + (lane & 1) * 4);
                if (oh < p.H && ow < p.W) *reinterpret_cast<f32x4*>(p.y + (((size_t)(n * p.H + oh) * p.W + ow) * 8 + (lane & 1) * 4) * 4) = v;
            } else {                                                              // 16 B per pixel: lanes 0..31
                const int ow = tw_i * TO_TW + l31;
                const f32x4 a = *reinterpret_cast<const f32x4*>(ot + (j * 32 + l31) * 8), b = *reinterpret_cast<const f32x4*>(ot + (j * 32 + l31) * 8 + 4);
                u32x4 o;
                bf16_t* ob = reinterpret_cast<bf16_t*>(&o);
#pragma unroll
                for (int e = 0; e < 4; ++e) { ob[e] = (bf16_t)a[e]; ob[4 + e] = (bf16_t)b[e]; }
                if (g == 0 && oh < p.H && ow < p.W) *reinterpret_cast<u32x4*>(p.y + ((size_t)(n * p.H + oh) * p.W + ow) * 16) = o;
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

}  // namespace

static bool thin_wgrad_setup(const MasConvDesc* d, ThinWgradParams& p, bool& big_is_x) {
    static const int on = mas_env_int("MAS_CONV_THIN", 1);
    if (!on) return false;
    if (d->ks != 3 || d->stride != 1 || d->upsample || d->act != MAS_ACT_NONE || d->pad_top != 1 || d->pad_left != 1) return false;
    if (d->in_dtype != MAS_BF16 || d->Ho != d->H || d->Wo != d->W) return false;
    if (!((d->Cin == 128 && d->Cout == 8) || (d->Cin == 8 && d->Cout == 128))) return false;
    if ((long long)d->N * d->H * d->W * 256 >= 0x7fffffffLL) return false;
    big_is_x = d->Cin == 128;
    p.N = d->N; p.H = d->H; p.W = d->W;
    p.tiles_h = mas_cdiv(d->H, TH_TH); p.tiles_w = mas_cdiv(d->W, TH_TW);
    p.n_tiles = d->N * p.tiles_h * p.tiles_w;
    if (big_is_x) {      // dW[co = cs][tap][ci = cb]; dy pixel = x pixel - (tap - 1); the bias gradient sums dy = S
        p.sgn = -1; p.cs_stride = 9 * 128; p.tap_stride = 128; p.col_stride = 1; p.bias_small = 1;
    } else {             // dW[co = cb][tap][ci = cs]; x pixel = dy pixel + (tap - 1); the bias gradient sums dy = B
        p.sgn = 1; p.col_stride = 72; p.tap_stride = 8; p.cs_stride = 1; p.bias_small = 0;
    }
    int ns = 2 * mas_num_cus();
    if (ns > p.n_tiles / 4) ns = p.n_tiles / 4;
    if (ns < 1) ns = 1;
    p.nsplit = ns;
    return true;
}

int mas_wgrad_thin_splits(const MasConvDesc* d) {
    ThinWgradParams p; bool bx;
    return thin_wgrad_setup(d, p, bx) ? p.nsplit : 0;
}

// part [nsplit][Cout][3][3][Cin], part_bias [nsplit][Cout] or NULL (see mas_conv_wgrad_partial)
int mas_wgrad_thin_partial(const MasConvDesc* d, const void* x, const void* dy, float* part, float* part_bias, hipStream_t s) {
    ThinWgradParams p; bool bx;
    if (!thin_wgrad_setup(d, p, bx)) return 0;
    p.big = (const unsigned char*)(bx ? x : dy); p.small = (const unsigned char*)(bx ? dy : x); p.part = part; p.part_bias = part_bias;
    hipLaunchKernelGGL(wgrad_thin_kernel, dim3((unsigned)p.nsplit), dim3(TH_NT), TH_LDS, s, p);
    MAS_CHECK_LAUNCH("wgrad_thin");
    return 1;
}

// 3x3 / stride 1 / pad 1 / bf16, 8 -> 128 channels, no prologue, no residual: returns 1 if launched, 0 if conv_fwd.hip should take it
int mas_conv_thin_fwd_try(const MasConvDesc* d, const void* x, const void* w_packed, const float* bias, const void* residual, void* y, hipStream_t s) {
    static const int on = mas_env_int("MAS_CONV_THIN", 1);
    if (!on || residual) return 0;
    if (d->ks != 3 || d->stride != 1 || d->upsample || d->act != MAS_ACT_NONE || d->pad_top != 1 || d->pad_left != 1) return 0;
    if (d->in_dtype != MAS_BF16 || d->out_dtype != MAS_BF16 || d->w_layout != MAS_WLAYOUT_K64) return 0;
    if (d->Cin != 8 || d->Cout != 128 || d->Ho != d->H || d->Wo != d->W) return 0;
    if ((long long)d->N * d->H * d->W * 256 >= 0x7fffffffLL) return 0;
    ThinFwdParams p;
    p.x = (const unsigned char*)x; p.w = (const unsigned char*)w_packed; p.bias = bias; p.y = (unsigned char*)y;
    p.N = d->N; p.H = d->H; p.W = d->W; p.rows_pad = 128;
    p.tiles_h = mas_cdiv(d->H, TF_TH); p.tiles_w = mas_cdiv(d->W, TF_TW); p.n_tiles = d->N * p.tiles_h * p.tiles_w;
    int grid = 8 * mas_num_cus();                                                // two resident work-groups per CU, four rounds
    if (grid > p.n_tiles) grid = p.n_tiles;
    hipLaunchKernelGGL(conv_thin_fwd_kernel, dim3((unsigned)grid), dim3(TH_NT), TF_LDS + 4 * 8192, s, p);
    MAS_CHECK_LAUNCH("conv_thin_fwd");
    return 1;
}

// 3x3 / stride 1 / pad 1 / bf16 in, C -> 8 channels (C = 64 or 128), no prologue, no residual: conv_out's forward.  Returns 1 if
// launched, 0 if conv_fwd.hip should take it
int mas_conv_thin_out_try(const MasConvDesc* d, const void* x, const void* w_packed, const float* bias, const void* residual, void* y, hipStream_t s) {
    static const int on = mas_env_int("MAS_CONV_THIN", 1), on_out = mas_env_int("MAS_CONV_THIN_OUT", 1);
    if (!on || !on_out || residual) return 0;
    if (d->ks != 3 || d->stride != 1 || d->upsample || d->act != MAS_ACT_NONE || d->pad_top != 1 || d->pad_left != 1) return 0;
    if (d->in_dtype != MAS_BF16 || (d->out_dtype != MAS_BF16 && d->out_dtype != MAS_F32) || d->w_layout != MAS_WLAYOUT_K64) return 0;
    if (d->Cout != 8 || d->Cin % 64 != 0 || d->Cin > 32 * TO_MAXCHUNKS || d->Ho != d->H || d->Wo != d->W) return 0;
    if ((long long)d->N * d->H * d->W * d->Cin * 2 >= 0x7fffffffLL) return 0;
    ThinOutParams p;
    p.x = (const unsigned char*)x; p.w = (const unsigned char*)w_packed; p.bias = bias; p.y = (unsigned char*)y;
    p.N = d->N; p.H = d->H; p.W = d->W; p.Cin = d->Cin; p.n_chunks = d->Cin / 32; p.rows_pad = 128;
    p.tiles_h = mas_cdiv(d->H, TO_TH); p.tiles_w = mas_cdiv(d->W, TO_TW); p.n_tiles = d->N * p.tiles_h * p.tiles_w;
    int grid = 8 * mas_num_cus();                                                // two resident work-groups per CU (70 KiB of LDS each), four rounds
    if (grid > p.n_tiles) grid = p.n_tiles;
    static mas_devmask_t attr_mask{0};
    unsigned long long attr_bit;
    if (mas_attr_needed(attr_mask, &attr_bit)) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(conv_thin_out_kernel<float>), hipFuncAttributeMaxDynamicSharedMemorySize, TO_LDS) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void*>(conv_thin_out_kernel<bf16_t>), hipFuncAttributeMaxDynamicSharedMemorySize, TO_LDS) != hipSuccess)
            MAS_FAIL(MAS_ELAUNCH, "conv_thin_out: cannot set dynamic LDS size %d", TO_LDS);
        mas_attr_done(attr_mask, attr_bit);
    }
    if (d->out_dtype == MAS_F32) hipLaunchKernelGGL(conv_thin_out_kernel<float>, dim3((unsigned)grid), dim3(TH_NT), TO_LDS, s, p);
    else hipLaunchKernelGGL(conv_thin_out_kernel<bf16_t>, dim3((unsigned)grid), dim3(TH_NT), TO_LDS, s, p);
    MAS_CHECK_LAUNCH("conv_thin_out");
    return 1;
}
