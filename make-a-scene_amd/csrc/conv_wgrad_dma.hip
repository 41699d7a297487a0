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
    float* part; float* part_bias;                 // split-K partials [nsplit][Cout][9][Cin] / [nsplit][Cout] (plain stores, fixed-order reduce
                                                   // by mas_wgrad_reduce) instead of fp32 atomics into dw / dbias
    int N, H, W, Cin, Ho, Wo, Cout;
    int Hl, Wl, pad_top, pad_left, act, upsample;
    int tiles_h, tiles_w, n_pt, n_co_t, n_ci_t, nsplit;
    unsigned long long* dbg;                       // -DD_TIMELINE builds only
    // KS = 2 (the sub-pixel form of Upsample + conv, see below): dy is the HIGH-resolution tensor walked as four phase images
    int dy_px, dy_row, dy_ph_px, dy_ph_row;        // bytes between low-resolution neighbours of a phase image; offset of phase column / row
    unsigned dy_img, dy_bytes;                     // bytes of one image of dy / of the whole tensor
};

constexpr int D_THW = 8, D_TWW = 16, D_PH = 10, D_PW = 18, D_NPP = 180;
constexpr int D_DY = 128 * 256;                // one dY tile: 128 pixels x 128 couts x 2 B
constexpr int D_XP = 23 * 1024;                // one patch: 180 pixels x 128 B -> 23 DMA pieces of 8 pixels
constexpr int D_SS = 2 * 1024;                 // scale/shift rows of two tiles (64 channels x 2 floats = 512 B, in a 1 KiB DMA piece)
constexpr int D_LDS = 2 * D_DY + 3 * D_XP + D_SS;     // 138240 B
constexpr int D_OOB = (int)0x80000000;

typedef __attribute__((ext_vector_type(4))) short d_s16x4;
__device__ __forceinline__ bf16x8 d_tr_frag(const unsigned char* a0, const unsigned char* a1) {
#ifdef D_ABL_NOREAD     // timing experiment only: fragments without LDS reads
    { const unsigned k = (unsigned)(size_t)a0 | 0x3f803f80u; u32x4 q = {k, k, k, k}; return *reinterpret_cast<const bf16x8*>(&q); }
#endif
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

template <int I> struct d_ic { static constexpr int v = I; };

// SPLIT = 1: 8 waves, each a 32(co) x 32(ci) x 9-tap accumulator (144 registers, 2 waves per SIMD).
// SPLIT = 2 (-DD_SPLIT2 experiment builds): 16 waves; waves w and w + 8 share a 32 x 32 tile and split its taps 5 / 4 (80 / 64
//            accumulator registers, 4 waves per SIMD).  The idea: a wave issues in order and one LDS read costs it ~20 cycles, one
//            DMA piece 60-180 (tools/probes/lds_rate.hip), so two waves per SIMD need ~5.2k cycles of issue time per tile against
//            4.6k cycles of MFMA pipe and serialise on top of that (measured 7.5k); four waves would each carry half the instruction
//            stream.  It does not fit: 80 + 24 fragment registers leave 24 of the 128-register cap for everything else, the
//            compiler spills 88 registers into the loop.
// KS = 2 (round 5): the weight gradient of `Upsample` + 3x3 (reference models/modules.py:44-59) in its sub-pixel form (conv_up2.hip): for
// output phase (a, b) the phase image dy_ab[i][j] = dy[2i + a][2j + b] is correlated with the 2x2 window x[i + a - 1 + r][j + b - 1 + s] of the
// LOW-resolution input -- 16 tap-correlations per low-resolution pixel instead of 36.  Same kernel: the grid carries the phase
// (blockIdx % 4), the patch origin moves by (a, b) (pad_top = 1 - a, pad_left = 1 - b), the dY DMA plan walks dy at pixel stride 2, a
// tile accumulates 4 taps instead of 9 (its step 9 is idle), and the slab of a work-group is [Cout][2x2][Cin] behind the phase's slabs;
// mas_wgrad_reduce_up2 folds the 4 x 4 phase taps back into the 3x3 gradient.  The launch is HBM-bound (dy once, x four times = the
// same bytes as the 3x3 form, 2.25x fewer MFMAs).
template <bool ACT, int SPLIT, int KS = 3>
__global__ __launch_bounds__(512 * SPLIT) void conv_wgrad_dma_kernel(DmaWgradParams p) {
    static_assert(KS == 3 || (KS == 2 && SPLIT == 1 && !ACT), "the 2x2 phase form: prologue-free, 8 waves");
    constexpr int NW = 8 * SPLIT, NT = 512 * SPLIT;
    constexpr int DYK = 4 / SPLIT;                 // dY pieces per wave
    constexpr int XK = SPLIT == 1 ? 3 : 2;         // patch piece slots per wave (piece = wave + NW k, valid below 23)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* const dyb = smem;               // [2][D_DY]
    unsigned char* const xb = smem + 2 * D_DY;     // [3][D_XP]
    unsigned char* const ssb = smem + 2 * D_DY + 3 * D_XP;        // [2][1024]
    const unsigned lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) unsigned char*)smem;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wt = wave & 7, half = wave >> 3;
    const int g = lane >> 5, l31 = lane & 31, G16 = (lane >> 4) & 1, sl = lane & 15;
    const int wco = (wt & 3) * 32, wci = (wt >> 2) * 32;

    int bid = blockIdx.x;
    const int phase = KS == 2 ? bid & 3 : 0;       // (a, b) = (phase >> 1, phase & 1)
    if constexpr (KS == 2) bid >>= 2;
    const int pad_top = KS == 2 ? 1 - (phase >> 1) : p.pad_top, pad_left = KS == 2 ? 1 - (phase & 1) : p.pad_left;
    const int split = bid % p.nsplit; bid /= p.nsplit;
    const int ci_t = bid % p.n_ci_t; const int co_t = bid / p.n_ci_t;
    const int co0 = co_t * 128, ci0 = ci_t * 64;
    const int n_mine = (p.n_pt - split + p.nsplit - 1) / p.nsplit;     // tiles of this work-group: split, split + nsplit, ...

    const d_i32x4 rs_dy = d_rsrc(p.dy, KS == 2 ? p.dy_bytes : (unsigned)((size_t)p.N * p.Ho * p.Wo * p.Cout * 2));
    const d_i32x4 rs_x = d_rsrc(p.x, (unsigned)((size_t)p.N * p.H * p.W * p.Cin * 2));
    const d_i32x4 rs_ss = d_rsrc(p.ss, ACT ? (unsigned)((size_t)p.N * p.Cin * 8) : 0u);

    // ---- tile cursor: the decode (image, tile row, tile column) is advanced incrementally (no division in the loop) and parks on
    //      the last tile (harmless re-reads at the end)
    // ud / ux: byte offset of the tile's first dY pixel / of input pixel (h0, w0) (both for channel co0 / ci0), fd / fx: the tile and
    // its halo lie inside the image -- the DMA offsets of such a tile are `lane constant + ud / ux` (one VALU add per piece); edge
    // tiles take the general per-lane address code.  (Round 2 recomputed the full address of every piece -- two integer multiplies
    // and ~20 VALU per piece, 7 pieces per tile -- which made ~150 of the ~320 VALU instructions a wave issues per 72 MFMAs.)
    struct TC { int n, h0, w0, idx, ud, ux; bool fd, fx; };
    const int ups = p.upsample ? 1 : 0;
    auto finish_tc = [&](TC& c) {
        if constexpr (KS == 2) c.ud = (int)((unsigned)c.n * p.dy_img) + (c.h0 + wave) * p.dy_row + c.w0 * p.dy_px + (phase >> 1) * p.dy_ph_row + (phase & 1) * p.dy_ph_px + co0 * 2;
        else c.ud = (((c.n * p.Ho + c.h0 + wave) * p.Wo + c.w0) * p.Cout + co0) * 2;
        c.ux = (((c.n * p.H + (c.h0 >> ups)) * p.W + (c.w0 >> ups)) * p.Cin + ci0) * 2;
        c.fd = (c.h0 + D_THW <= p.Ho) && (c.w0 + D_TWW <= p.Wo);
        c.fx = (c.h0 >= pad_top) && (c.h0 - pad_top + D_PH <= p.Hl) && (c.w0 >= pad_left) && (c.w0 - pad_left + D_PW <= p.Wl);
    };
    const int adv_w = (p.nsplit % p.tiles_w) * D_TWW, adv_q = p.nsplit / p.tiles_w;
    const int adv_h = (adv_q % p.tiles_h) * D_THW, adv_n = adv_q / p.tiles_h;
    const int lim_w = p.tiles_w * D_TWW, lim_h = p.tiles_h * D_THW;
    auto first_tile = [&]() {
        int t = split;
        TC c; c.w0 = (t % p.tiles_w) * D_TWW; t /= p.tiles_w;
        c.h0 = (t % p.tiles_h) * D_THW; c.n = t / p.tiles_h; c.idx = 0;
        finish_tc(c);
        return c;
    };
    auto next_tile = [&](TC c) {
        if (c.idx + 1 >= n_mine) return c;
        c.idx += 1;
        c.w0 += adv_w; int cy = c.w0 >= lim_w; c.w0 -= cy ? lim_w : 0;
        c.h0 += adv_h + (cy ? D_THW : 0); cy = c.h0 >= lim_h; c.h0 -= cy ? lim_h : 0;
        c.n += adv_n + cy;
        finish_tc(c);
        return c;
    };
    auto fresh_lane = [&]() { int l = lane; asm volatile("" : "+v"(l)); return l; };   // per-piece address terms are recomputed,
                                                                                       // not kept in registers across the tile loop
    // lane parts of the fast-path offsets.  dY piece DYK wave + k: pixel row `wave` (SPLIT 1: 16 wave + 4 k + (lane >> 4) < 16 (wave + 1)),
    // column 4 k + (lane >> 4): lane part = column (lane >> 4), block, slot; + k * 8 Cout bytes per piece.  Patch piece wave + NW k:
    // pixel P = 8 (wave + NW k) + (lane >> 3) -> (pr, pc) relative to the tile's (h0 - pad_top, w0 - pad_left); with the Upsample fold
    // (h0, w0 even) the source pixel is ((h0 >> 1) + ((pr - pad_top) >> 1), ...).  Dead lanes (P >= 180) carry the out-of-range marker.
    static_assert(SPLIT == 1 || true, "");
    const int dy_px = KS == 2 ? p.dy_px : p.Cout * 2;            // bytes between horizontally adjacent pixels of the (phase) image
    const int dyL = (lane >> 4) * dy_px + (((((lane >> 2) & 3) ^ ((lane >> 4) & 3)) * 32) + (lane & 3) * 8) * 2;
    int xL[XK], prpc[XK];                          // prpc: patch row | patch column << 8 of the lane's pixel (the bounds test of edge tiles)
#pragma unroll
    for (int k = 0; k < XK; ++k) {
        const int P = (wave + NW * k) * 8 + (lane >> 3);
        const int pr = (P * 3641) >> 16, pc = P - pr * D_PW;
        prpc[k] = pr | (pc << 8);
        const int r = (pr - pad_top) >> ups, cc = (pc - pad_left) >> ups;               // arithmetic shifts: -1 >> 1 == -1
        const int blk = ((lane >> 2) & 1) ^ ((P >> 1) & 1);
        xL[k] = (P < D_NPP) ? ((r * p.W + cc) * p.Cin + blk * 32 + (lane & 3) * 8) * 2 : D_OOB;
    }

    // dY: wave w moves pieces DYK w .. DYK w + DYK - 1 (4 pixels x 256 B each); lane -> pixel 4 piece + (lane >> 4), physical 64-byte
    // block (lane >> 2) & 3 holding logical block ^ (pixel & 3), 16-byte slot lane & 3
    auto dy_issue = [&](const TC& c, int buf, int k) {
        const int piece = wave * DYK + k;
        int vo;
        if constexpr (SPLIT == 1) {                 // pixel row `wave`, column 4 k + (lane >> 4): lane constant + uniform base
            vo = dyL + (c.ud + k * 4 * dy_px);
            if (!c.fd) {                            // ragged tile (uniform branch): rows / columns past the map read as zeros
                const bool ok = (c.h0 + wave < p.Ho) && (c.w0 + 4 * k + (lane >> 4) < p.Wo);
                vo = ok ? vo : D_OOB;
            }
        } else {
            const int lane = fresh_lane();
            const int pix = piece * 4 + (lane >> 4);
            const int ho = c.h0 + (pix >> 4), wo = c.w0 + (pix & 15);
            const int blk = ((lane >> 2) & 3) ^ (pix & 3);
            const bool ok = (ho < p.Ho) && (wo < p.Wo);
            vo = ok ? (int)((((size_t)(c.n * p.Ho + ho) * p.Wo + wo) * p.Cout + co0 + blk * 32 + (lane & 3) * 8) * 2) : D_OOB;
        }
#ifdef D_ABL_NODMA
        if (p.N != -12345) return;
#endif
        d_dma16(rs_dy, __builtin_amdgcn_readfirstlane(lds0 + buf * D_DY + piece * 1024), vo);
    };
    // patch: wave w moves pieces w, w + NW, .. below 23 (8 pixels x 128 B each); lane -> patch pixel 8 piece + (lane >> 3), physical
    // block (lane >> 2) & 1 holding logical block ^ ((pixel >> 1) & 1), slot lane & 3
    auto x_inb = [&](int k, const TC& c) -> bool {                // is the lane's patch pixel a pixel of the image (else: zero padding)
        const int pr = prpc[k] & 0xff, pc = prpc[k] >> 8;
        return (unsigned)(c.h0 - pad_top + pr) < (unsigned)p.Hl && (unsigned)(c.w0 - pad_left + pc) < (unsigned)p.Wl;
    };
    auto x_issue = [&](const TC& c, int buf, int k) {
        if (wave + NW * k >= 23) return;
        int vo = xL[k] + c.ux;                      // (dead lanes: the marker stays out of range after the add)
        if (!c.fx) vo = x_inb(k, c) ? vo : D_OOB;   // edge tile (uniform branch): halo pixels outside the image read as zeros
#ifdef D_ABL_NODMA
        if (p.N != -12345) return;
#endif
        d_dma16(rs_x, __builtin_amdgcn_readfirstlane(lds0 + 2 * D_DY + buf * D_XP + (wave + NW * k) * 1024), vo);
    };
    // (scale, shift) of the 64 channels of a tile's image: 512 B, lanes 0..31 of one DMA piece, ONE copy per work-group (wave 0 moves
    // it BEFORE the barrier that precedes its first use)
    auto ss_issue = [&](const TC& c, int par) {
        if (wave != 0) return;
        const int lane = fresh_lane();
        const int vo = lane < 32 ? (int)(((size_t)c.n * p.Cin + ci0) * 8 + lane * 16) : D_OOB;
        d_dma16(rs_ss, __builtin_amdgcn_readfirstlane(lds0 + 2 * D_DY + 3 * D_XP + par * 1024), vo);
    };
    // GroupNorm(+SiLU) in place: this thread takes LOGICAL 16-byte slot lane & 7 (channels ci0 + 8 (lane & 7) ..+7) of the pixels of
    // its own wave's pieces; padding pixels were written as zeros by the DMA and stay zero
    f32x4 rss[4];                                  // the lane's 8 (scale, shift) pairs of the tile being activated
    auto ss_fetch = [&](int par) {
        const f32x4* sp = reinterpret_cast<const f32x4*>(ssb + par * 1024 + (lane & 7) * 64);
#pragma unroll
        for (int q = 0; q < 4; ++q) rss[q] = sp[q];
    };
    auto x_activate = [&](const TC& c, int buf, int k) {
        if (wave + NW * k >= 23) return;
        const int P = (wave + NW * k) * 8 + (lane >> 3);
        if (P >= D_NPP || !(c.fx || x_inb(k, c))) return;
        const int u = lane & 7;
        unsigned char* dst = xb + buf * D_XP + P * 128 + ((((u >> 2) ^ ((P >> 1) & 1))) << 6) + (u & 3) * 16;
        u32x4 v = *reinterpret_cast<const u32x4*>(dst);
        if (p.act == MAS_ACT_AFFINE_SILU) {
#pragma unroll
            for (int q = 0; q < 4; ++q)                           // rss[q] = (scale, shift) of channels 2q, 2q+1
                v[q] = act_pair_bf16<true>(v[q], f32x2{rss[q][0], rss[q][2]}, f32x2{rss[q][1], rss[q][3]});
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q)
                v[q] = act_pair_bf16<false>(v[q], f32x2{rss[q][0], rss[q][2]}, f32x2{rss[q][1], rss[q][3]});
        }
        *reinterpret_cast<u32x4*>(dst) = v;
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
    const bool do_bias = (p.dbias != nullptr || p.part_bias != nullptr) && (ci_t == 0);

    // ---- prologue: all of tile 0, the scale/shift rows of tiles 0 and 1 and the dY pieces of tile 1 that steady state issues in
    //      steps 8, 9 of the previous iteration; everything landed, tile 0 activated, barrier
    TC cA = first_tile(), cN = next_tile(cA);
    if constexpr (ACT) { ss_issue(cA, 0); ss_issue(cN, 1); }
#pragma unroll
    for (int k = 0; k < DYK; ++k) dy_issue(cA, 0, k);
#pragma unroll
    for (int k = 0; k < XK; ++k) x_issue(cA, 0, k);
#pragma unroll
    for (int k = 0; k < DYK && k < 2; ++k) dy_issue(cN, 1, k);
    D_WAIT(0);
    if constexpr (ACT) {
        __builtin_amdgcn_s_barrier();                             // the scale/shift row is wave 0's piece
        asm volatile("" ::: "memory");
        ss_fetch(0);
#pragma unroll
        for (int k = 0; k < XK; ++k) x_activate(cA, 0, k);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    cA = cN;                                       // inside iteration j: cA = tile j+1 (fetched in steps 0..7), cN = tile j+2 (steps 7..9)

#ifdef D_TIMELINE      // s_memtime stamps of iterations 40 and 41 of work-group 100: [0] arrive, [1] released, [2 + s] MFMAs of step s issued
    unsigned long long tsv[12];
#define DTS(k) asm volatile("s_memtime %0" : "=s"(tsv[k]) :: "memory")
#else
#define DTS(k)
#endif

    // ---- schedule.  A tile is 10 steps (patch rows); the taps of this wave are [LO, HI).  SPLIT = 1 prefetches the patch fragments
    // of step s+1 during step s (double-buffered registers); SPLIT = 2 reads them at the start of their step (the SIMD's other three
    // waves cover the latency).  The ONE work-group barrier per tile sits between steps 7 and 8 -- not at the tile boundary -- so the
    // thin steps 8, 9 of tile j and 0, 1 of tile j+1 run through without a pipeline drain.  It publishes tile j+1: its first two dY
    // pieces were issued in steps 8, 9 of iteration j-1, the other pieces in steps 0.. of iteration j, the in-place activation runs
    // after a vmcnt(0) in the steps that follow; its scale/shift row was issued (wave 0) in step 7 of iteration j-1, BEFORE that
    // iteration's barrier.  Every wave waits for its own pieces (vmcnt(0)) before it arrives.
    // Buffers: dY(t) in t & 1 (last read in step 7 of its tile, before the barrier that precedes the first write of dY(t+2));
    // patch(t) in t % 3 (read until step 8 of iteration t, AFTER that iteration's barrier, hence the third buffer).
    f32x16 acc[SPLIT == 1 ? KS * KS : 5];
    auto run = [&](auto HALF_T) {
        constexpr int HALF = decltype(HALF_T)::v;
        constexpr int LO = SPLIT == 1 ? 0 : (HALF ? 5 : 0), HI = SPLIT == 1 ? KS * KS : (HALF ? 9 : 5);
#ifdef D_NO_ACT_PREFETCH   // A/B builds: round 2's rule (the prologue variant spent those 12 registers on address temporaries)
        constexpr bool PREFETCH = SPLIT == 1 && !ACT;
#else
        constexpr bool PREFETCH = SPLIT == 1;             // (round 3: the leaner address code freed the 12 registers for the prologue variant too)
#endif
#pragma unroll
        for (int t = 0; t < HI - LO; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
        bf16x8 aw[3], bf[PREFETCH ? 2 : 1][3];
        int bxo[3][2], xsel = 0;
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) { bxo[kw][0] = 2 * D_DY + b_off[kw]; bxo[kw][1] = 2 * D_DY + (b_off[kw] ^ 64); }
        auto kw_needed = [&](int pr, int kw) {                    // does step pr of this wave use the patch fragment kw ?
            bool need = false;
#pragma unroll
            for (int kh = 0; kh < KS; ++kh) {
                const int t = kh * KS + kw, rr = pr - kh;
                if (kw < KS && t >= LO && t < HI && rr >= 0 && rr < D_THW) need = true;
            }
            return need;
        };
        // patch fragment addresses: bxo[kw][parity of the patch row] = LDS byte offset of the CURRENT tile's patch buffer + the lane's
        // swizzled offset; a read is then `base register + compile-time offset` (row * 2304, +512 for the second half) -- round 2
        // formed every address with two VALU adds (120 per tile).  The bases move to the next buffer of the ring once per tile.
        auto load_b = [&](int pr) {                               // -> bf[PREFETCH ? pr & 1 : 0], from the buffer bxo points at
#ifdef D_KW_REUSE      // experiment builds only -- measured SLOWER (0.594 vs 0.584 ms at 128->128 @256^2; step 63.52 vs 63.28 ms, one box):
            if constexpr (SPLIT == 1) {
                // The three kw fragments of a patch row are windows of the SAME 10 pixels (8 g + kw .. 8 g + kw + 7): three transpose reads
                // (pixels +0..3, +4..7, +8..11 of the lane's channel; a dword = two pixels) instead of six -- kw = 2 is the window shifted
                // by one dword, kw = 1 by half a dword (four v_alignbit).  80 -> 50 LDS reads per tile, but +70 VALU: the odd-register
                // window of kw = 2 is not a legal MFMA operand (tuples are even-aligned), so it is copied, and an issued VALU instruction
                // costs this kernel as much as an issued LDS read (R2.2's cost model).  (SPLIT 1: a step needs all three kw or none.)
                if (!kw_needed(pr, 0)) return;
                const unsigned char* b0 = smem + bxo[0][pr & 1] + pr * (D_PW * 128);
                const d_s16x4 r0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) d_s16x4*)b0);
                const d_s16x4 r1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) d_s16x4*)(b0 + 4 * 128));
                const d_s16x4 r2 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) d_s16x4*)(b0 + 8 * 128));
                const u32x2 a = *reinterpret_cast<const u32x2*>(&r0), b = *reinterpret_cast<const u32x2*>(&r1), c = *reinterpret_cast<const u32x2*>(&r2);
                const u32x4 f0 = {a[0], a[1], b[0], b[1]}, f2 = {a[1], b[0], b[1], c[0]};
                const u32x4 f1 = {__builtin_amdgcn_alignbit(a[1], a[0], 16), __builtin_amdgcn_alignbit(b[0], a[1], 16),
                                  __builtin_amdgcn_alignbit(b[1], b[0], 16), __builtin_amdgcn_alignbit(c[0], b[1], 16)};
                bf[PREFETCH ? (pr & 1) : 0][0] = *reinterpret_cast<const bf16x8*>(&f0);
                bf[PREFETCH ? (pr & 1) : 0][1] = *reinterpret_cast<const bf16x8*>(&f1);
                bf[PREFETCH ? (pr & 1) : 0][2] = *reinterpret_cast<const bf16x8*>(&f2);
                return;
            }
#endif
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                if (!kw_needed(pr, kw)) continue;
                const unsigned char* b0 = smem + bxo[kw][pr & 1] + pr * (D_PW * 128);                  // patch pixels (pr, kw + 8g + j)
                bf[PREFETCH ? (pr & 1) : 0][kw] = d_tr_frag(b0, b0 + 4 * 128);
            }
        };
        auto advance_bases = [&]() {                              // ring of three patch buffers
            const int dxn = xsel == 2 ? -2 * D_XP : D_XP;
            xsel = xsel == 2 ? 0 : xsel + 1;
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) { bxo[kw][0] += dxn; bxo[kw][1] += dxn; }
        };
        auto load_a = [&](const unsigned char* dys, int pr) {     // -> aw[pr % 3]
            const unsigned char* a0 = dys + pr * (16 * 256) + a_lane;                                 // dY pixels (pr, 8g + j)
            aw[pr % 3] = d_tr_frag(a0, a0 + 4 * 256);
        };
        auto mfma_kh = [&](int pr, int kh) {
            const int rr = pr - kh;
            if (rr < 0 || rr >= D_THW || kh >= KS) return;
#pragma unroll
            for (int kw = 0; kw < KS; ++kw) {
                const int t = kh * KS + kw;
                if (t < LO || t >= HI) continue;
#ifdef D_ABL_NOMFMA     // timing experiment only
                acc[t - LO][0] += (float)aw[rr % 3][0] + (float)bf[PREFETCH ? (pr & 1) : 0][kw][0];
#else
                mma16(acc[t - LO], aw[rr % 3], bf[PREFETCH ? (pr & 1) : 0][kw]);
#endif
            }
        };
        if constexpr (PREFETCH) load_b(0);
        load_a(dyb, 0);
        for (int j = 0; j < n_mine; ++j) {
#ifdef D_TIMELINE
            if (j >= 41 && j < 43 && blockIdx.x == 100 && lane == 0 && p.dbg) {
#pragma unroll
                for (int k = 0; k < 12; ++k) p.dbg[((j - 41) * NW + wave) * 16 + k] = tsv[k];
            }
            asm volatile("" ::: "memory");
#endif
            const unsigned char* dys = dyb + (j & 1) * D_DY;
            const unsigned char* xs = xb + (j % 3) * D_XP;
            const int xn = (j + 1) % 3, dn = (j + 1) & 1;         // buffers of tile j+1
#pragma unroll
            for (int pr = 0; pr < D_PH; ++pr) {
                if (pr == 8) {                                    // ---- the barrier: tile j+1 complete and visible, dY(j) dead
                    DTS(0);
                    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#ifndef D_ABL_NOBAR
                    __builtin_amdgcn_s_barrier();
#endif
                    asm volatile("" ::: "memory");
                    DTS(1);
                }
                if constexpr (PREFETCH) {
                    if (pr + 1 < D_PH) load_b(pr + 1);
                    else { advance_bases(); load_b(0); }          // step 9: row 0 of the next tile (every read of this tile's patch has been issued)
                } else {
                    load_b(pr);
                    if (pr == D_PH - 1) advance_bases();
                }
                __builtin_amdgcn_sched_barrier(0);
                mfma_kh(pr, 2);                                   // the oldest dY row first: its registers take the next fragment
                __builtin_amdgcn_sched_barrier(0);
                // ---- side work of this step.  DMA pieces of a wave in issue order: dY 0 .. DYK-1, then patch 0 .. XK-1, at steps
                //      8, 9 (tile j+2) and 0, 1, 2, .. (tile j+1)
#pragma unroll
                for (int q = 0; q < DYK + XK; ++q) {
                    if ((8 + q) % D_PH != pr) continue;
                    const bool late = q < 2;                      // issued at the end of the previous iteration
                    if (q < DYK) dy_issue(late ? cN : cA, late ? (j & 1) : dn, q);
                    else x_issue(late ? cN : cA, late ? (j + 2) % 3 : xn, q - DYK);
                }
                constexpr int ACT0 = (8 + DYK + XK) % D_PH + (SPLIT == 1 ? 0 : 2);     // first activation step: 5 (SPLIT 1), 4 (SPLIT 2)
                if (pr == (SPLIT == 1 ? (ACT ? 4 : 5) : 2) && do_bias) {   // column sums of the dY tile: channel unit tid & 15, pixels (tid >> 4) + (NT / 16) q
#pragma unroll
                    for (int q = 0; q < 2048 / NT; ++q) {
                        const int pix = (tid >> 4) + (NT / 16) * q, cu = tid & 15;
                        const u32x4 raw = *reinterpret_cast<const u32x4*>(dys + pix * 256 + (((cu >> 2) ^ (pix & 3)) << 6) + (cu & 3) * 16);
                        const bf16_t* rv = reinterpret_cast<const bf16_t*>(&raw);
#pragma unroll
                        for (int e = 0; e < 8; ++e) bsum[e] += (float)rv[e];
                    }
                }
                if constexpr (ACT) {
                    if (pr == ACT0) { D_WAIT(0); ss_fetch(dn); }  // this wave's pieces of tile j+1 (the oldest issued a tile ago) have landed
                    if (pr >= ACT0 && pr < ACT0 + XK) x_activate(cA, xn, pr - ACT0);
                }
                if (pr == 7) {
                    cN = next_tile(cA);
                    if constexpr (ACT) ss_issue(cN, j & 1);       // tile j+2's row; slot j & 1 was last read in iteration j-1
                }
                __builtin_amdgcn_sched_barrier(0);
                mfma_kh(pr, 1);
                __builtin_amdgcn_sched_barrier(0);
                if (pr + 1 < D_THW) load_a(dys, pr + 1);
                else if (pr == D_PH - 1) load_a(dyb + dn * D_DY, 0);   // step 9: dY row 0 of the next tile (slot 0: row 6 is done)
                __builtin_amdgcn_sched_barrier(0);
                mfma_kh(pr, 0);
#ifdef D_TIMELINE
                __builtin_amdgcn_sched_barrier(0);
                DTS(2 + pr);
                if (pr == D_PH - 1) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
                __builtin_amdgcn_sched_barrier(0);
            }
            cA = cN;
        }
        D_WAIT(0);                                                // the look-ahead DMA of tiles past the end must land before the LDS is reused / the block exits

        const int ci = ci0 + wci + l31;
#ifdef D_ABL_NOATOM     // timing experiment only
        if (p.N != -12345) {
            float t = 0.0f;
            for (int k = 0; k < HI - LO; ++k) for (int r = 0; r < 16; ++r) t += acc[k][r];
            if (t == 123.456f) p.dw[0] = t;
            return;
        }
#endif
        if (p.part) {
            // this work-group's own slab of the partial table: plain coalesced stores (a half-wave writes 128 consecutive bytes).  The
            // fp32 atomics they replace cost 45-60 us per launch on every shape but the largest (18.9 M read-modify-writes at the L2:
            // profiles/r03_wgrad_commit.txt) and made the sums depend on arrival order
            float* pw = p.part + (size_t)(phase * p.nsplit + split) * ((size_t)p.Cout * (KS * KS) * p.Cin);
#pragma unroll
            for (int t = LO; t < HI; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int co = co0 + wco + acc_row(lane, r);
                    pw[((size_t)co * (KS * KS) + t) * p.Cin + ci] = acc[t - LO][r];
                }
        } else {
#pragma unroll
            for (int t = LO; t < HI; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int co = co0 + wco + acc_row(lane, r);
                    atomicAdd(p.dw + ((size_t)co * 9 + t) * p.Cin + ci, acc[t - LO][r]);
                }
        }
    };
    if (SPLIT == 1 || half == 0) run(d_ic<0>{}); else run(d_ic<1>{});
#ifdef D_ABL_NOATOM
    if (p.N != -12345) return;
#endif
    if (do_bias) {                                 // thread (tid & 15) = channel unit, (tid >> 4) = pixel phase: fixed-order sum over the phases
        __syncthreads();
        float* red = reinterpret_cast<float*>(smem);               // [NT / 16][128]
#pragma unroll
        for (int e = 0; e < 8; ++e) red[(tid >> 4) * 128 + (tid & 15) * 8 + e] = bsum[e];
        __syncthreads();
        if (tid < 128) {
            float t = 0.0f;
            for (int k = 0; k < NT / 16; ++k) t += red[k * 128 + tid];
            if (p.part_bias) p.part_bias[(size_t)(phase * p.nsplit + split) * p.Cout + co0 + tid] = t;
            else atomicAdd(p.dbias + co0 + tid, t);
        }
    }
}

template <bool ACT, int SPLIT, int KS = 3>
int launch_dma(DmaWgradParams p, hipStream_t s) {
    auto kern = conv_wgrad_dma_kernel<ACT, SPLIT, KS>;
    static mas_devmask_t attr_mask{0};
    unsigned long long attr_bit;
    if (mas_attr_needed(attr_mask, &attr_bit)) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, D_LDS) != hipSuccess)
            MAS_FAIL(MAS_ELAUNCH, "conv_wgrad_dma: cannot set dynamic LDS size %d", D_LDS);
        mas_attr_done(attr_mask, attr_bit);
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)(p.n_co_t * p.n_ci_t * p.nsplit * (KS == 2 ? 4 : 1))), dim3(512 * SPLIT), D_LDS, s, p);
    MAS_CHECK_LAUNCH(KS == 2 ? "conv_wgrad_up2" : "conv_wgrad_dma");
    return MAS_OK;
}

// MAS_WGRAD_CUS: how many CUs the persistent 3x3 weight-gradient grid is sized for.  0 (the library's default): all of them.  n > 0: n.
// -1: three quarters -- what mas_hip.ops sets when it runs the weight gradient on a second stream beside the GroupNorm backward passes
// (MAS_WGRAD_STREAM=1, the default since late round 6): with one work-group on EVERY CU (138 KB of LDS each) the 6 us finalize launch between
// the two GroupNorm passes is not placed until the weight gradient retires and the apply pass never overlaps; with a quarter of the CUs free it
// starts at once and the passes run beside the weight gradient: step -1.35 ms at 192 of 256 CUs (208: -0.8, 176: -1.15, 160: -1.0, 128: 0;
// profiles/r06_wgrad_stream.txt).  Alone, a 192-CU grid is 0.6 ms per step SLOWER: the two knobs go together.
static int wgrad_cus() {
    const int n = mas_env_int("MAS_WGRAD_CUS", 0);       // read per call (one getenv): the host side drops the 3/4 grid when its side stream is refused
    const int all = mas_num_cus();
    if (n == -1) return all * 3 / 4 > 0 ? all * 3 / 4 : all;
    return (n > 0 && n < all) ? n : all;
}

}  // namespace

static bool dma_setup(const MasConvDesc* d, DmaWgradParams& p) {
    static const int mode = mas_env_int("MAS_WGRAD_DMA", 1);
    if (!mode) return false;
    if (d->ks != 3 || d->stride != 1 || d->in_dtype != MAS_BF16) return false;
    if (d->Cin % 64 != 0 || d->Cout % 128 != 0 || d->act > MAS_ACT_AFFINE_SILU) return false;
    const long long xb = (long long)d->N * d->H * d->W * d->Cin * 2, yb = (long long)d->N * d->Ho * d->Wo * d->Cout * 2;
    if (xb >= 0x7fffffffLL || yb >= 0x7fffffffLL) return false;
    p.N = d->N; p.H = d->H; p.W = d->W; p.Cin = d->Cin; p.Ho = d->Ho; p.Wo = d->Wo; p.Cout = d->Cout;
    p.Hl = d->upsample ? 2 * d->H : d->H; p.Wl = d->upsample ? 2 * d->W : d->W;
    p.pad_top = d->pad_top; p.pad_left = d->pad_left; p.act = d->act; p.upsample = d->upsample;
    p.tiles_h = mas_cdiv(p.Ho, D_THW); p.tiles_w = mas_cdiv(p.Wo, D_TWW);
    p.n_pt = p.N * p.tiles_h * p.tiles_w;
    p.n_co_t = d->Cout / 128; p.n_ci_t = d->Cin / 64;
    p.dbg = nullptr;
    const int out_tiles = p.n_co_t * p.n_ci_t;
    // One register-file-filling work-group per CU.  Under a co-running RCCL collective (data-parallel backward) some CUs are taken
    // and a grid of exactly one work-group per CU runs the displaced ones as a second FULL round; MAS_WGRAD_OVERSUB=2 (bench.py sets
    // it for N > 1) halves the work-groups so the hardware rebalances at half-round granularity, for 2x the split-K partials.
    static const int oversub = mas_env_int("MAS_WGRAD_OVERSUB", 1);
    int nsplit = mas_cdiv(wgrad_cus() * (oversub > 0 ? oversub : 1), out_tiles);
    if (nsplit > p.n_pt) nsplit = p.n_pt;
    if (nsplit < 1) nsplit = 1;
    p.nsplit = nsplit;
    return true;
}

static int dma_launch(DmaWgradParams& p, hipStream_t s) {
#ifdef D_TIMELINE
    if (const char* e = getenv("MAS_DBG_PTR")) p.dbg = reinterpret_cast<unsigned long long*>(strtoull(e, nullptr, 0));
#endif
    const bool act = p.act != MAS_ACT_NONE;
#ifdef D_SPLIT2         // experiment builds only: the 16-wave tap-split variant does not fit the 128-register cap (88 spilled registers)
    static const int wsplit = mas_env_int("MAS_WGRAD_SPLIT", 2);
    if (wsplit == 2) return act ? launch_dma<true, 2>(p, s) : launch_dma<false, 2>(p, s);
#endif
    return act ? launch_dma<true, 1>(p, s) : launch_dma<false, 1>(p, s);
}

// Returns 1 if the shape qualifies and the launch was made, 0 if the caller should use the other kernels, < 0 on error.
int mas_conv_wgrad_dma_try(const MasConvDesc* d, const void* x, const float* scale_shift, const void* dy, float* dw, float* dbias,
                           hipStream_t s) {
    DmaWgradParams p;
    if (!dma_setup(d, p)) return 0;
    p.x = (const unsigned char*)x; p.ss = scale_shift; p.dy = (const unsigned char*)dy; p.dw = dw; p.dbias = dbias;
    p.part = nullptr; p.part_bias = nullptr;
    const int rc = dma_launch(p, s);
    return rc == MAS_OK ? 1 : rc;
}

// Split-K factor of the LDS-DMA kernel for this convolution (= slabs of the partial table mas_conv_wgrad_partial writes), or 0 when the
// shape does not take that kernel (mas_conv_wgrad with its atomic commit is the path then).
int mas_wgrad1x1_splits(const MasConvDesc* d);                                   // conv1x1.hip
int mas_wgrad1x1_partial(const MasConvDesc* d, const void* x, const void* dy, float* part, float* part_bias, hipStream_t s);

int mas_wgrad_s2_splits(const MasConvDesc* d);                                    // conv_s2.hip
int mas_wgrad_s2_partial(const MasConvDesc* d, const void* x, const void* dy, float* part, float* part_bias, hipStream_t s);
int mas_wgrad_thin_splits(const MasConvDesc* d);                                  // conv_thin.hip
int mas_wgrad_thin_partial(const MasConvDesc* d, const void* x, const void* dy, float* part, float* part_bias, hipStream_t s);

int mas_conv_wgrad_general_splits(const MasConvDesc* d);                          // conv_wgrad.hip: the general kernels in slab mode
int mas_conv_wgrad_general_partial(const MasConvDesc* d, const void* x, const float* scale_shift, const void* dy, float* part, float* part_bias,
                                   hipStream_t s);

// MAS_WGRAD_GENERAL_SLABS=0: shapes without a slab kernel of their own report 0 splits again and take mas_conv_wgrad's fp32 atomics
static bool general_slabs() { static const int on = mas_env_int("MAS_WGRAD_GENERAL_SLABS", 1); return on != 0; }

extern "C" int mas_conv_wgrad_splits(const MasConvDesc* d) {
    DmaWgradParams p;
    if (!d) return 0;
    int k;
    if (d->ks == 1) k = mas_wgrad1x1_splits(d);
    else if (d->stride == 2) k = mas_wgrad_s2_splits(d);
    else if (!dma_setup(d, p)) k = mas_wgrad_thin_splits(d);
    else k = p.nsplit;
    if (k > 0 || !general_slabs()) return k;
    return mas_conv_wgrad_general_splits(d);
}

// The weight gradient as split-K PARTIALS: part [nsplit][Cout][3][3][Cin] fp32 and (when non-NULL) part_bias [nsplit][Cout], every
// element written exactly once by plain stores (no zero-initialisation needed, no atomics); mas_wgrad_reduce sums the slabs in a
// fixed order: bitwise run-to-run deterministic.  nsplit = mas_conv_wgrad_splits(d) (> 0 required).
extern "C" int mas_conv_wgrad_partial(const MasConvDesc* d, const void* x, const float* scale_shift, const void* dy, float* part,
                                      float* part_bias, void* stream) {
    MAS_ENTER();
    if (!d || !x || !dy || !part) MAS_FAIL(MAS_EINVAL, "conv_wgrad_partial: null argument");
    if (d->act != MAS_ACT_NONE && !scale_shift) MAS_FAIL(MAS_EINVAL, "conv_wgrad_partial: act prologue needs scale_shift");
    if (d->ks == 1) {                            // plain GEMM (conv1x1.hip)
        const int rc = mas_wgrad1x1_partial(d, x, dy, part, part_bias, reinterpret_cast<hipStream_t>(stream));
        if (rc == 0 && general_slabs() && mas_conv_wgrad_general_splits(d) > 0)
            return mas_conv_wgrad_general_partial(d, x, scale_shift, dy, part, part_bias, reinterpret_cast<hipStream_t>(stream));
        if (rc == 0) MAS_FAIL(MAS_EUNSUPPORTED, "conv_wgrad_partial: this 1x1 convolution does not take the split-K partial path (mas_conv_wgrad_splits == 0)");
        return rc < 0 ? rc : MAS_OK;
    }
    if (d->stride == 2) {                        // Downsample: conv_s2.hip
        const int rc = mas_wgrad_s2_partial(d, x, dy, part, part_bias, reinterpret_cast<hipStream_t>(stream));
        if (rc == 0 && general_slabs() && mas_conv_wgrad_general_splits(d) > 0)
            return mas_conv_wgrad_general_partial(d, x, scale_shift, dy, part, part_bias, reinterpret_cast<hipStream_t>(stream));
        if (rc == 0) MAS_FAIL(MAS_EUNSUPPORTED, "conv_wgrad_partial: this stride-2 convolution does not take the split-K partial path (mas_conv_wgrad_splits == 0)");
        return rc < 0 ? rc : MAS_OK;
    }
    DmaWgradParams p;
    if (!dma_setup(d, p)) {                      // the RGB-edge layers (8 <-> 128 channels): conv_thin.hip
        const int rc = mas_wgrad_thin_partial(d, x, dy, part, part_bias, reinterpret_cast<hipStream_t>(stream));
        if (rc != 0) return rc < 0 ? rc : MAS_OK;
    }
    if (!dma_setup(d, p) && general_slabs() && mas_conv_wgrad_general_splits(d) > 0)      // every other shape: the general kernels' slab mode
        return mas_conv_wgrad_general_partial(d, x, scale_shift, dy, part, part_bias, reinterpret_cast<hipStream_t>(stream));
    if (!dma_setup(d, p)) MAS_FAIL(MAS_EUNSUPPORTED, "conv_wgrad_partial: this convolution does not take the split-K partial path (mas_conv_wgrad_splits == 0)");
    p.x = (const unsigned char*)x; p.ss = scale_shift; p.dy = (const unsigned char*)dy; p.dw = nullptr; p.dbias = nullptr;
    p.part = part; p.part_bias = part_bias;
    return dma_launch(p, reinterpret_cast<hipStream_t>(stream));
}

// ---- Upsample + conv, sub-pixel form: the weight gradient as split-K partials of the 4 x 2x2 phase correlations (KS = 2 above) --------
// d describes the FORWARD convolution (upsample = 1; H x W the input map, Ho x Wo = 2H x 2W, 3x3, stride 1, pads 1, bf16, no prologue).
static bool up2_wgrad_setup(const MasConvDesc* d, DmaWgradParams& p) {
    static const int mode = mas_env_int("MAS_CONV_UP2_WGRAD", 1), mode_all = mas_env_int("MAS_CONV_UP2", 1);
    if (!mode || !mode_all || !d) return false;
    if (!d->upsample || d->ks != 3 || d->stride != 1 || d->pad_top != 1 || d->pad_left != 1 || d->act != MAS_ACT_NONE) return false;
    if (d->in_dtype != MAS_BF16 || d->Ho != 2 * d->H || d->Wo != 2 * d->W || d->N <= 0 || d->H <= 0 || d->W <= 0) return false;
    if (d->Cin % 64 != 0 || d->Cout % 128 != 0) return false;
    const long long xb = (long long)d->N * d->H * d->W * d->Cin * 2, yb = (long long)d->N * d->Ho * d->Wo * d->Cout * 2;
    if (xb >= 0x7fffffffLL || yb >= 0x7fffffffLL) return false;
    p.N = d->N; p.H = d->H; p.W = d->W; p.Cin = d->Cin; p.Ho = d->H; p.Wo = d->W; p.Cout = d->Cout;      // the tiles walk the LOW-resolution grid
    p.Hl = d->H; p.Wl = d->W; p.pad_top = 1; p.pad_left = 1; p.act = MAS_ACT_NONE; p.upsample = 0;
    p.tiles_h = mas_cdiv(p.Ho, D_THW); p.tiles_w = mas_cdiv(p.Wo, D_TWW);
    p.n_pt = p.N * p.tiles_h * p.tiles_w;
    p.n_co_t = d->Cout / 128; p.n_ci_t = d->Cin / 64;
    p.dbg = nullptr; p.ss = nullptr;
    p.dy_ph_px = d->Cout * 2; p.dy_ph_row = d->Wo * d->Cout * 2; p.dy_px = 2 * p.dy_ph_px; p.dy_row = 2 * p.dy_ph_row;
    p.dy_img = (unsigned)((size_t)d->Ho * d->Wo * d->Cout * 2); p.dy_bytes = (unsigned)yb;
    static const int oversub = mas_env_int("MAS_WGRAD_OVERSUB", 1);
    int nsplit = mas_cdiv(mas_num_cus() * (oversub > 0 ? oversub : 1), p.n_co_t * p.n_ci_t * 4);      // per phase: the four phases fill the chip together
    if (nsplit > p.n_pt) nsplit = p.n_pt;
    if (nsplit < 1) nsplit = 1;
    p.nsplit = nsplit;
    // narrow maps waste the 8 x 16-pixel tiles (and conv_up2's forward keeps them on the 3x3 kernels too): same rule as the forward
    static const int any_width = mas_env_int("MAS_CONV_WIDE_ANY_WIDTH", 0);
    if (!any_width && d->W < 32) return false;
    return true;
}

// slabs PER PHASE (the partial table holds 4 x that many slabs of [Cout][2x2][Cin], the bias table 4 x that many rows), 0 = unsupported
extern "C" int mas_conv_up2_wgrad_splits(const MasConvDesc* d) {
    DmaWgradParams p;
    return up2_wgrad_setup(d, p) ? p.nsplit : 0;
}

extern "C" int mas_conv_up2_wgrad_partial(const MasConvDesc* d, const void* x, const void* dy, float* part, float* part_bias, void* stream) {
    MAS_ENTER();
    if (!d || !x || !dy || !part) MAS_FAIL(MAS_EINVAL, "conv_up2_wgrad_partial: null argument");
    DmaWgradParams p;
    if (!up2_wgrad_setup(d, p)) MAS_FAIL(MAS_EUNSUPPORTED, "conv_up2_wgrad_partial: unsupported convolution (mas_conv_up2_wgrad_splits == 0)");
    p.x = (const unsigned char*)x; p.dy = (const unsigned char*)dy; p.dw = nullptr; p.dbias = nullptr; p.part = part; p.part_bias = part_bias;
    return launch_dma<false, 1, 2>(p, reinterpret_cast<hipStream_t>(stream));
}
