// 3x3 / stride-1 / bf16 convolution weight gradient with LDS-DMA staging (the shapes that carry the FLOPs: Cin % 64 == 0,
// Cout % 128 == 0 -- autograd of reference models/modules.py:93,100,113 and the Upsample convs :49).
//
// Same math and MFMA operand scheme as conv_wgrad_tr_kernel (conv_wgrad.hip): dW[tap][co][ci] = sum_pixels dY[pixel][co] *
// A[pixel (+) tap][ci], both operands staged in their natural [pixel][channel] layout and fetched with the LDS transpose read
// (ds_read_b64_tr_b16), one 8-wave work-group per 128(co) x 64(ci) x 9-tap accumulator block, split-K over 8x16-pixel tiles,
// fp32 atomics at the end.  What changes is how the operands get to LDS.  The round-1 kernel moved them through registers
// (global -> VGPR -> ds_write), single-buffered, two work-group barriers per 72 MFMAs; its timeline showed 38 %
// of a tile spent in the issue of the next tile's loads, the staging stores and the barriers.  Here:
//   * dY tiles (128 pixels x 256 B) are double-buffered and the input halo patch (180 pixels x 128 B) triple-buffered in LDS,
//     filled by `buffer_load_dwordx4 ... lds` (no VGPRs, no VALU, no ds_write; the descriptor's bounds check writes the zero
//     padding), issued one / two tiles ahead behind COUNTED vmcnt waits -- ONE barrier per tile;
//   * rows are unpadded (a DMA piece is 1 KiB of consecutive LDS), so the bank spreading of the transpose reads comes from an
//     XOR swizzle of the 64-byte block index instead of the 64-byte row padding: block ^ (pixel & 3) for the 256-byte dY rows,
//     block ^ ((pixel >> 1) & 1) for the 128-byte patch rows (any 4 consecutive pixels then tile the 256-byte bank row);
//   * the GroupNorm(+SiLU) prologue is applied IN PLACE to the patch of the NEXT tile (each wave activates the pieces it
//     DMA'd itself: no extra barrier), the bias-gradient column sums are read back from the LDS dY tile; the per-image
//     scale/shift rows come by DMA too, two tiles ahead, into a wave-private 2 x 1 KiB slot;
//   * the DMA instructions are written in inline assembly.  With the clang builtin the compiler knows a buffer_load ... lds
//     is pending and puts an `s_waitcnt vmcnt(0)` in front of the first ds_read_b64_tr_b16 that follows (its transpose-read
//     intrinsic carries no memory operand to disambiguate) -- which waits for the look-ahead that was issued a few hundred
//     cycles earlier, every tile.  Ordering is entirely by the counted waits below.
#include "mas_common.h"

namespace {

struct DmaWgradParams {
    const unsigned char* x; const float* ss; const unsigned char* dy; float* dw; float* dbias;
    int N, H, W, Cin, Ho, Wo, Cout;
    int Hl, Wl, pad_top, pad_left, act, upsample;
    int tiles_h, tiles_w, n_pt, n_co_t, n_ci_t, nsplit;
};

constexpr int D_NT = 512, D_THW = 8, D_TWW = 16, D_PH = 10, D_PW = 18, D_NPP = 180;
constexpr int D_DY = 128 * 256;                // one dY tile: 128 pixels x 128 couts x 2 B
constexpr int D_XP = 23 * 1024;                // one patch: 180 pixels x 128 B -> 23 DMA pieces of 8 pixels
constexpr int D_SS = 8 * 2 * 1024;             // scale/shift rows: per wave 2 x (64 channels x 2 floats = 512 B, in a 1 KiB DMA piece)
constexpr int D_LDS = 2 * D_DY + 3 * D_XP + D_SS;     // 152576 B
constexpr int D_OOB = (int)0x80000000;

typedef __attribute__((ext_vector_type(4))) short d_s16x4;
__device__ __forceinline__ bf16x8 d_tr_frag(const unsigned char* a0, const unsigned char* a1) {
    const d_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) d_s16x4*)a0);
    const d_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) d_s16x4*)a1);
    const __attribute__((ext_vector_type(8))) short v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return *reinterpret_cast<const bf16x8*>(&v);
}

#define D_WAIT(N) asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory")

typedef __attribute__((ext_vector_type(4))) int d_i32x4;
// one LDS-DMA piece: 64 lanes x 16 B from descriptor `rs` at per-lane byte offset `vo` (out of range -> zeros) to LDS byte
// address `lds` (wave-uniform) + 16 lane
__device__ __forceinline__ void d_dma16(d_i32x4 rs, unsigned lds, int vo) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" :: "s"(lds), "v"(vo), "s"(rs) : "memory", "m0");
}
__device__ __forceinline__ d_i32x4 d_rsrc(const void* base, unsigned bytes) {
    const unsigned long long a = (unsigned long long)base;
    d_i32x4 r = {(int)(unsigned)a, (int)(unsigned)(a >> 32), (int)bytes, 0x00020000};
    r[0] = __builtin_amdgcn_readfirstlane(r[0]); r[1] = __builtin_amdgcn_readfirstlane(r[1]);
    r[2] = __builtin_amdgcn_readfirstlane(r[2]); r[3] = __builtin_amdgcn_readfirstlane(r[3]);
    return r;
}

template <bool ACT>
__global__ __launch_bounds__(D_NT) void conv_wgrad_dma_kernel(DmaWgradParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* const dyb = smem;               // [2][D_DY]
    unsigned char* const xb = smem + 2 * D_DY;     // [3][D_XP]
    unsigned char* const ssb = smem + 2 * D_DY + 3 * D_XP;        // [8 waves][2][1024]
    const unsigned lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) unsigned char*)smem;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 5, l31 = lane & 31, G16 = (lane >> 4) & 1, sl = lane & 15;
    const int wco = (wave & 3) * 32, wci = (wave >> 2) * 32;

    int bid = blockIdx.x;
    const int split = bid % p.nsplit; bid /= p.nsplit;
    const int ci_t = bid % p.n_ci_t; const int co_t = bid / p.n_ci_t;
    const int co0 = co_t * 128, ci0 = ci_t * 64;
    const int n_mine = (p.n_pt - split + p.nsplit - 1) / p.nsplit;     // tiles of this work-group: split, split + nsplit, ...

    const d_i32x4 rs_dy = d_rsrc(p.dy, (unsigned)((size_t)p.N * p.Ho * p.Wo * p.Cout * 2));
    const d_i32x4 rs_x = d_rsrc(p.x, (unsigned)((size_t)p.N * p.H * p.W * p.Cin * 2));
    const d_i32x4 rs_ss = d_rsrc(p.ss, ACT ? (unsigned)((size_t)p.N * p.Cin * 8) : 0u);

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;

    auto coords = [&](int i, int& n, int& h0, int& w0) {          // i-th tile of this work-group (clamped: harmless re-reads at the end)
        int t = split + (i < n_mine ? i : n_mine - 1) * p.nsplit;
        const int tw_i = t % p.tiles_w; t /= p.tiles_w;
        const int th_i = t % p.tiles_h; n = t / p.tiles_h;
        h0 = th_i * D_THW; w0 = tw_i * D_TWW;
    };
    // dY: wave w moves pieces 4w .. 4w+3 (4 pixels x 256 B each); lane -> pixel 4 piece + (lane >> 4), physical 64-byte block
    // (lane >> 2) & 3 holding logical block ^ (pixel & 3), 16-byte slot lane & 3
    auto dy_issue = [&](int i, int buf, int k0 = 0, int k1 = 4) {
        int n, h0, w0;
        coords(i, n, h0, w0);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (k < k0 || k >= k1) continue;
            const int piece = wave * 4 + k;
            const int pix = piece * 4 + (lane >> 4);
            const int ho = h0 + (pix >> 4), wo = w0 + (pix & 15);
            const int blk = ((lane >> 2) & 3) ^ (pix & 3);
            const bool ok = (ho < p.Ho) && (wo < p.Wo);
            const int vo = ok ? (int)((((size_t)(n * p.Ho + ho) * p.Wo + wo) * p.Cout + co0 + blk * 32 + (lane & 3) * 8) * 2) : D_OOB;
#ifdef D_ABL_NODMA
            if (p.N != -12345) continue;
#endif
            d_dma16(rs_dy, __builtin_amdgcn_readfirstlane(lds0 + buf * D_DY + piece * 1024), vo);
        }
    };
    // patch: wave w moves pieces w, w+8, w+16 (8 pixels x 128 B each; wave 7 repeats piece 22 as its third); lane -> patch pixel
    // 8 piece + (lane >> 3), physical block (lane >> 2) & 1 holding logical block ^ ((pixel >> 1) & 1), slot lane & 3
    auto x_piece = [&](int k) { return (k < 2 || wave < 7) ? wave + 8 * k : 22; };
    auto x_pix = [&](int k, int h0, int w0, int& P, int& ih, int& iw) -> bool {
        P = x_piece(k) * 8 + (lane >> 3);
        const int pr = (P * 3641) >> 16, pc = P - pr * D_PW;     // P / 18 for P < 3641
        ih = h0 + pr - p.pad_top; iw = w0 + pc - p.pad_left;
        const bool inb = (P < D_NPP) && (ih >= 0) && (ih < p.Hl) && (iw >= 0) && (iw < p.Wl);
        if (p.upsample) { ih >>= 1; iw >>= 1; }
        return inb;
    };
    auto x_issue = [&](int i, int buf, int k0 = 0, int k1 = 3) {
        int n, h0, w0;
        coords(i, n, h0, w0);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            if (k < k0 || k >= k1) continue;
            int P, ih, iw;
            const bool inb = x_pix(k, h0, w0, P, ih, iw);
            const int blk = ((lane >> 2) & 1) ^ ((P >> 1) & 1);
            const int vo = inb ? (int)((((size_t)(n * p.H + ih) * p.W + iw) * p.Cin + ci0 + blk * 32 + (lane & 3) * 8) * 2) : D_OOB;
#ifdef D_ABL_NODMA
            if (p.N != -12345) continue;
#endif
            d_dma16(rs_x, __builtin_amdgcn_readfirstlane(lds0 + 2 * D_DY + buf * D_XP + x_piece(k) * 1024), vo);
        }
    };
    // GroupNorm(+SiLU) in place: this thread takes LOGICAL 16-byte slot lane & 7 (channels ci0 + 8 (lane & 7) ..+7) of the pixels of
    // its own wave's pieces; padding pixels were written as zeros by the DMA and stay zero
    // (scale, shift) of the 64 channels of tile i's image: 512 B, lanes 0..31 of one DMA piece, into this wave's slot i & 1
    auto ss_issue = [&](int i) {
        int n, h0, w0;
        coords(i, n, h0, w0);
        const int vo = lane < 32 ? (int)(((size_t)n * p.Cin + ci0) * 8 + lane * 16) : D_OOB;
        d_dma16(rs_ss, __builtin_amdgcn_readfirstlane(lds0 + 2 * D_DY + 3 * D_XP + (wave * 2 + (i & 1)) * 1024), vo);
    };
    auto x_activate = [&](int i, int buf) {
        int n, h0, w0;
        coords(i, n, h0, w0);
        f32x4 rss[4];
        {
            const f32x4* sp = reinterpret_cast<const f32x4*>(ssb + (wave * 2 + (i & 1)) * 1024 + (lane & 7) * 64);
#pragma unroll
            for (int q = 0; q < 4; ++q) rss[q] = sp[q];
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            if (k == 2 && wave == 7) continue;                    // piece 22 is wave 6's
            int P, ih, iw;
            if (!x_pix(k, h0, w0, P, ih, iw)) continue;
            const int u = lane & 7;
            unsigned char* dst = xb + buf * D_XP + P * 128 + ((((u >> 2) ^ ((P >> 1) & 1))) << 6) + (u & 3) * 16;
            u32x4 v = *reinterpret_cast<const u32x4*>(dst);
            bf16_t* tv = reinterpret_cast<bf16_t*>(&v);
            if (p.act == MAS_ACT_AFFINE_SILU) {
#pragma unroll
                for (int e = 0; e < 8; ++e) tv[e] = (bf16_t)silu_f((float)tv[e] * rss[e >> 1][(e & 1) * 2] + rss[e >> 1][(e & 1) * 2 + 1]);
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) tv[e] = (bf16_t)((float)tv[e] * rss[e >> 1][(e & 1) * 2] + rss[e >> 1][(e & 1) * 2 + 1]);
            }
            *reinterpret_cast<u32x4*>(dst) = v;
        }
    };

    // ---- transpose-read lane addressing (conv_wgrad.hip tr_probe semantics): lane -> pixel 8 g + (sl >> 2) (+4 for the second read),
    //      channels 16 G16 + 4 (sl & 3) ..+3 of the wave's 32-channel tile; swizzled 64-byte block as above
    const int t4 = sl >> 2;
    const int a_lane = (8 * g + t4) * 256 + (((wco >> 5) ^ t4) << 6) + 32 * G16 + 8 * (sl & 3);
    int b_off[3];                                  // per kw, for EVEN patch rows; odd rows: ^ 64
#pragma unroll
    for (int kw = 0; kw < 3; ++kw)
        b_off[kw] = (8 * g + t4 + kw) * 128 + ((((wci >> 5) ^ ((kw + t4) >> 1)) & 1) << 6) + 32 * G16 + 8 * (sl & 3);

    float bsum[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) bsum[e] = 0.0f;
    const bool do_bias = (p.dbias != nullptr) && (ci_t == 0);

    // ---- prologue: (scale/shift of tiles 0 and 1,) dY(0), x(0), x(1); for the prologue variant everything landed and x(0) activated
    if constexpr (ACT) { ss_issue(0); ss_issue(1); }
    dy_issue(0, 0);
    x_issue(0, 0);
    x_issue(1, 1);
    if constexpr (ACT) {
        D_WAIT(0);
        x_activate(0, 0);
    }

    for (int i = 0; i < n_mine; ++i) {
        const int dsel = i & 1, xsel = i % 3;
        // every wave: its own DMA for tile i has landed (dY(i): issued one tile ago; x(i): two tiles ago), its activation of x(i) is in
        // LDS; after the barrier all of tile i is visible and the buffers of tile i-1 are free.  In flight across it: x(i+1) (3 pieces)
        // (the scale/shift of tile i+1 was issued BEFORE dY(i) and has landed with it)
        asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if constexpr (ACT) ss_issue(i + 2);                       // slot i & 1: tile i's values were last read one tile ago
        asm volatile("" ::: "memory");
#ifndef D_SPREAD_ISSUE
        dy_issue(i + 1, dsel ^ 1);
        x_issue(i + 2, (i + 2) % 3);
        asm volatile("" ::: "memory");
#endif
        if constexpr (ACT) {
#ifdef D_SPREAD_ISSUE
            D_WAIT(0);                                            // (spread issue: nothing younger than x(i+1) / the scale-shift is in flight yet)
#else
            D_WAIT(8);                                            // x(i+1) has landed; 1 + 4 + 3 younger pieces fly
#endif
            x_activate(i + 1, (i + 1) % 3);
            asm volatile("" ::: "memory");
        }
        const unsigned char* dys = dyb + dsel * D_DY;
        const unsigned char* xs = xb + xsel * D_XP;
        if (do_bias) {                                            // column sums of the dY tile: channel unit tid & 15, pixels (tid >> 4) + 32 j
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int pix = (tid >> 4) + 32 * j, cu = tid & 15;
                const u32x4 raw = *reinterpret_cast<const u32x4*>(dys + pix * 256 + (((cu >> 2) ^ (pix & 3)) << 6) + (cu & 3) * 16);
                const bf16_t* rv = reinterpret_cast<const bf16_t*>(&raw);
#pragma unroll
                for (int e = 0; e < 8; ++e) bsum[e] += (float)rv[e];
            }
        }
        // ---- MFMA: for every patch row, every tap that touches it (sliding window of dY rows)
        bf16x8 aw[3];
#pragma unroll
        for (int pr = 0; pr < D_PH; ++pr) {
            bf16x8 bf[3];
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const unsigned char* b0 = xs + pr * (D_PW * 128) + (b_off[kw] ^ ((pr & 1) << 6));      // patch pixels (pr, kw + 8g + j)
                bf[kw] = d_tr_frag(b0, b0 + 4 * 128);
            }
            if (pr < D_THW) {
                const unsigned char* a0 = dys + pr * (16 * 256) + a_lane;                                 // dY pixels (pr, 8g + j)
                aw[pr % 3] = d_tr_frag(a0, a0 + 4 * 256);
            }
#pragma unroll
            for (int kh = 0; kh < 3; ++kh) {
                const int rr = pr - kh;
                if (rr < 0 || rr >= D_THW) continue;
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) mma16(acc[kh * 3 + kw], aw[rr % 3], bf[kw]);
            }
#ifdef D_SPREAD_ISSUE   // one DMA piece of the look-ahead tiles after each of the first 7 patch rows (same VMEM order: dY x4, then x x3)
            if (pr < 4) dy_issue(i + 1, dsel ^ 1, pr, pr + 1);
            else if (pr < 7) x_issue(i + 2, (i + 2) % 3, pr - 4, pr - 3);
#endif
        }
    }
    D_WAIT(0);                                                    // the look-ahead DMA of tiles past the end must land before the LDS is reused / the block exits

    const int ci = ci0 + wci + l31;
#ifdef D_ABL_NOATOM     // timing experiment only
    if (p.N != -12345) {
        float t = 0.0f;
        for (int k = 0; k < 9; ++k) for (int r = 0; r < 16; ++r) t += acc[k][r];
        if (t == 123.456f) p.dw[0] = t;
        return;
    }
#endif
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = co0 + wco + acc_row(lane, r);
            atomicAdd(p.dw + ((size_t)co * 9 + t) * p.Cin + ci, acc[t][r]);
        }
    if (do_bias) {
        __syncthreads();
        float* red = reinterpret_cast<float*>(smem);
        for (int k = tid; k < 128; k += D_NT) red[k] = 0.0f;
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 8; ++e) atomicAdd(&red[(tid & 15) * 8 + e], bsum[e]);
        __syncthreads();
        for (int k = tid; k < 128; k += D_NT) atomicAdd(p.dbias + co0 + k, red[k]);
    }
}

template <bool ACT>
int launch_dma(DmaWgradParams p, hipStream_t s) {
    auto kern = conv_wgrad_dma_kernel<ACT>;
    static mas_devmask_t attr_mask{0};
    unsigned long long attr_bit;
    if (mas_attr_needed(attr_mask, &attr_bit)) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, D_LDS) != hipSuccess)
            MAS_FAIL(MAS_ELAUNCH, "conv_wgrad_dma: cannot set dynamic LDS size %d", D_LDS);
        mas_attr_done(attr_mask, attr_bit);
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)(p.n_co_t * p.n_ci_t * p.nsplit)), dim3(D_NT), D_LDS, s, p);
    MAS_CHECK_LAUNCH("conv_wgrad_dma");
    return MAS_OK;
}

}  // namespace

// Returns 1 if the shape qualifies and the launch was made, 0 if the caller should use the other kernels, < 0 on error.
int mas_conv_wgrad_dma_try(const MasConvDesc* d, const void* x, const float* scale_shift, const void* dy, float* dw, float* dbias,
                           hipStream_t s) {
    static const int mode = mas_env_int("MAS_WGRAD_DMA", 1);
    if (!mode) return 0;
    if (d->ks != 3 || d->stride != 1 || d->in_dtype != MAS_BF16) return 0;
    if (d->Cin % 64 != 0 || d->Cout % 128 != 0 || d->act > MAS_ACT_AFFINE_SILU) return 0;
    const long long xb = (long long)d->N * d->H * d->W * d->Cin * 2, yb = (long long)d->N * d->Ho * d->Wo * d->Cout * 2;
    if (xb >= 0x7fffffffLL || yb >= 0x7fffffffLL) return 0;
    DmaWgradParams p;
    p.x = (const unsigned char*)x; p.ss = scale_shift; p.dy = (const unsigned char*)dy; p.dw = dw; p.dbias = dbias;
    p.N = d->N; p.H = d->H; p.W = d->W; p.Cin = d->Cin; p.Ho = d->Ho; p.Wo = d->Wo; p.Cout = d->Cout;
    p.Hl = d->upsample ? 2 * d->H : d->H; p.Wl = d->upsample ? 2 * d->W : d->W;
    p.pad_top = d->pad_top; p.pad_left = d->pad_left; p.act = d->act; p.upsample = d->upsample;
    p.tiles_h = mas_cdiv(p.Ho, D_THW); p.tiles_w = mas_cdiv(p.Wo, D_TWW);
    p.n_pt = p.N * p.tiles_h * p.tiles_w;
    p.n_co_t = d->Cout / 128; p.n_ci_t = d->Cin / 64;
    const int out_tiles = p.n_co_t * p.n_ci_t;
    // One register-file-filling work-group per CU.  Under a co-running RCCL collective (data-parallel backward) some CUs are taken
    // and a grid of exactly one work-group per CU runs the displaced ones as a second FULL round; MAS_WGRAD_OVERSUB=2 (bench.py sets
    // it for N > 1) halves the work-groups so the hardware rebalances at half-round granularity, for 2x the split-K atomics
    // (+3 % of this kernel on an idle GPU).
    static const int oversub = mas_env_int("MAS_WGRAD_OVERSUB", 1);
    int nsplit = mas_cdiv(mas_num_cus() * (oversub > 0 ? oversub : 1), out_tiles);
    if (nsplit > p.n_pt) nsplit = p.n_pt;
    if (nsplit < 1) nsplit = 1;
    p.nsplit = nsplit;
    const int rc = d->act != MAS_ACT_NONE ? launch_dma<true>(p, s) : launch_dma<false>(p, s);
    return rc == MAS_OK ? 1 : rc;
}
