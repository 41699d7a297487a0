// 3x3 / stride-1 / bf16 implicit-GEMM convolution, "wide" geometry, for the shapes that carry the FLOPs
// (Cin % 64 == 0, Cout % 128 == 0, maps at least 32 pixels wide: every ResnetBlock / Upsample conv of levels 256..32 and all
// of their data gradients -- reference models/modules.py:49,93,100,113 and autograd of the same sites).
//
// Why another geometry (evidence: profiles/r02_stream_timeline_v1.txt, r02_stream_pmc_v1.txt).  The stream kernel
// (conv3x3_stream.hip: 16x16-pixel tile, 64-channel chunks, wave tile 64 cout x 64 pixels) spends, per 32-MFMA stage and wave,
// ~1250 cycles in its MFMA block and ~1350 in everything that is paid ONCE PER STAGE whatever the amount of arithmetic behind
// it: the work-group barrier, the issue of the LDS-DMA pieces (92 .. 250 cycles each: the CU's vector-memory path takes
// 434 KB per 256-pixel tile, 288 KB of it the 3x3x128x128 weights that EVERY tile re-fetches from L2), the LDS latency in
// front of the first MFMA.  The matrix pipe is busy 60 % of the cycles and the chip clocks at 1.6 GHz under that load.
// This geometry halves what is paid per MFMA instead of re-arranging it:
//   * tile = 16 x 32 pixels x 128 couts, wave tile = 128 couts x 64 pixels (two tile rows): the weights are fetched once
//     per 512 pixels (half the weight DMA per FLOP), 6 fragment reads feed 8 MFMAs (0.75 ds_read_b128 per MFMA instead of 1);
//   * 32-channel chunks (64-byte LDS rows) keep the double-buffered halo patch at 2 x 39 KiB; a stage is one filter ROW of a
//     chunk (3 taps, 24 KiB of weights, 48 MFMAs per wave): 1.5x the MFMAs per barrier;
//   * 64-byte rows: the 16-byte slot is XOR-swizzled with (row >> 2) & 3 -- the 16 lanes of a ds_read_b128 phase (lanes
//     {0-3,12-15,20-27} / {4-11,16-19,28-31} of consecutive rows) then hit 16 different 16-byte bank groups, for the weight
//     rows (row = cout) and for the patch (row = patch pixel, any tap shift) alike.
// Everything HBM-facing is LDS-DMA (`buffer_load_dwordx4 ... lds`) issued inside a stage for a later one, with COUNTED waits
// (in-order VMEM retirement), as in the stream kernel; the GroupNorm(+SiLU) prologue is applied in place to the raw patch.
// LDS: 2 x 24 KiB weights + 2 x 39 KiB patch = 126 KiB, one 512-thread work-group per CU, 2 waves per SIMD, 256 VGPRs.
#include "mas_common.h"
#include <algorithm>
#include <utility>

namespace {

template <int... I, typename F>
__device__ __forceinline__ void w_static_for(std::integer_sequence<int, I...>, F&& f) { (f(std::integral_constant<int, I>{}), ...); }

struct WideParams {
    unsigned long long* dbg;                   // -DW_TIMELINE builds only
    const unsigned char* x; const float* ss; const unsigned char* w; const float* bias; const unsigned char* res; unsigned char* y;
    float* stats;                              // optional [N][tiles_h * tiles_w][Cout][2]: per-tile sum / sum of squares of the bf16 OUTPUT (the consumer's GroupNorm statistics)
    int N, H, W, Cin, Ho, Wo, Cout;
    int Hl, Wl, pad_top, pad_left, upsample, act;
    int n_chunks, Cout_pad, tiles_h, tiles_w, n_ct;
    unsigned m_ct, m_tw, m_th;                 // ceil(2^32 / d) for d = n_ct, tiles_w, tiles_h: t / d == umulhi(t, m) (host checks t * d < 2^32)
    int xcd_bands;                             // 1: every XCD walks its own contiguous eighth of the tile list (see the prologue)
};

constexpr int W_PWL = 34;                      // patch pitch in pixels (32 + 2)
constexpr int W_NPIX = 18 * 34;                // 612
constexpr int W_NPIECE = 39;                   // 612 pixels x 64 B = 39168 B -> 39 DMA pieces of 1 KiB (16 pixels each)
constexpr int W_PATCH = W_NPIECE * 1024;
constexpr int W_WT = 128 * 64;                 // one tap-step weight tile: 128 couts x 64 B
constexpr int W_WSTAGE = 3 * W_WT;             // one filter row of one chunk
constexpr int W_WBUF = 0;                      // LDS map: [2][W_WSTAGE] weights, then [2][W_PATCH] patch
constexpr int W_PBUF = 2 * W_WSTAGE;
constexpr int W_BIAS = W_PBUF + 2 * W_PATCH;   // [Cout] fp32 bias (zeros without one), staged once per work-group
constexpr int W_MAXCOUT = 2048;
constexpr int W_STAT = W_BIAS + W_MAXCOUT * 4;  // [8 waves][128 couts][2] fp32 per-wave partial statistics of the tile just finished
constexpr int W_NEXT = W_STAT + 8 * 128 * 2 * 4 + 16;   // [512 threads][4] the NEXT tile's output offsets / statistics row (kept out of the VGPRs)
constexpr int W_SS = W_NEXT + 512 * 16;        // [2][Cin <= 512][2] fp32 GroupNorm scale / shift of the current and of the next tile's image
constexpr int W_MAXCIN = 512;
constexpr int W_LDS = W_SS + 2 * W_MAXCIN * 8;   // + {table row, cout offset} of the pending statistics flush
constexpr int W_NSLOT = 5;                     // patch DMA pieces (and 16-byte activation slots) per wave (thread) per chunk
constexpr int W_OOB = (int)0x80000000;

#define W_WAIT_BARRIER(N) do { asm volatile("s_waitcnt vmcnt(" #N ") lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); \
                               asm volatile("" ::: "memory"); } while (0)

#ifdef W_TIMELINE
#define WTS(id) do { __builtin_amdgcn_sched_barrier(0); if (lane == 0 && blockIdx.x == 100 && tl_iter >= 1 && tl_iter < 3 && p.dbg) \
                         p.dbg[((tl_iter - 1) * 8 + wave) * 64 + (id)] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define WTS(id) do {} while (0)
#endif

__device__ __forceinline__ void w_wait_barrier(int n) {   // n is a compile-time constant after unrolling, or selected by a uniform branch
    switch (n) {
        case 0: W_WAIT_BARRIER(0); break;
        case 2: W_WAIT_BARRIER(2); break;
        case 3: W_WAIT_BARRIER(3); break;
        case 5: W_WAIT_BARRIER(5); break;
        case 7: W_WAIT_BARRIER(7); break;
        case 32: W_WAIT_BARRIER(32); break;
        default: W_WAIT_BARRIER(0); break;
    }
}

// ACT: GroupNorm(+SiLU) prologue (in-place activation of the raw patch).  RES: residual add in the epilogue.
// STATS: per-tile sum / sum of squares of the output channels (the consumer's GroupNorm statistics) written to p.stats.
template <bool ACT, bool RES, bool STATS>
__global__ __launch_bounds__(512, 2) void conv3x3_wide_kernel(WideParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* const wbuf = smem + W_WBUF;
    unsigned char* const patch = smem + W_PBUF;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // wave w owns tile rows 2w, 2w+1 and all 128 couts
    const int g = lane >> 5, l31 = lane & 31;

    const size_t img_bytes = (size_t)p.H * p.W * p.Cin * 2;
    const unsigned out_bytes = (unsigned)((size_t)p.N * p.Ho * p.Wo * p.Cout * 2);
    const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, out_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_r = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(RES ? p.res : p.y), 0, RES ? out_bytes : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(p.w), 0,
                                                                           (unsigned)(9 * p.n_chunks * p.Cout_pad * 64), 0x00020000);


    // ---- tiles: persistent work-group, static stride.  Divisions by the (runtime) tile-grid extents are multiply-high by
    //      host-made reciprocals: hipcc's generic 32-bit division keeps ~10 SGPRs of loop-invariant temporaries alive per
    //      divisor, and this kernel has no SGPRs to spare (a spilled SGPR comes back through scratch + vmcnt(0)).
    const int total_tiles = p.N * p.tiles_h * p.tiles_w * p.n_ct;
    struct Tile { int n, h0, w0, c0; };
    auto udiv = [](int t, unsigned m, int d, int& q, int& r) {
        q = (int)__umulhi((unsigned)t, m);
        if (d == 1) q = t;                       // ceil(2^32 / 1) does not fit 32 bits
        r = t - q * d;
    };
    auto decode = [&](int t) {
        Tile tc;
        int q, ct, tw_i, th_i;
        udiv(t, p.m_ct, p.n_ct, q, ct); t = q;
        udiv(t, p.m_tw, p.tiles_w, q, tw_i); t = q;
        udiv(t, p.m_th, p.tiles_h, q, th_i);
        tc.n = q; tc.c0 = ct * 128; tc.h0 = th_i * 16; tc.w0 = tw_i * 32;
        return tc;
    };

    // ---- patch plan -------------------------------------------------------------------------------------------------------
    // slot k of this thread: patch pixel q = 128 k + (tid >> 2) (= pixel (lane >> 2) of DMA piece wave + 8 k; wave 7 has no
    // fifth piece and repeats piece 38, so every wave issues the same number of VMEM operations).  DMA: the LDS image is
    // lane-linear (physical slot lane & 3), the swizzle goes on the SOURCE channel offset.  Activation: the thread takes
    // LOGICAL slot lane & 3 of the same pixel (one fixed set of 8 scale/shift pairs) at its swizzled LDS address.
    auto slot_pix = [&](int k, int& q, int& pr, int& pc) -> bool {
        const int piece = (k < 4) ? wave + 8 * k : (wave < 7 ? 32 + wave : 38);
        q = piece * 16 + (lane >> 2);
        pr = (q * 1928) >> 16;                   // q / 34 for q < 640
        pc = q - pr * W_PWL;
        return q < W_NPIX;
    };
    auto make_plan = [&](const Tile& tc, int (&vo)[W_NSLOT], unsigned& inb_mask, int (&ob)[4]) {
        inb_mask = 0;
#pragma unroll
        for (int j = 0; j < 2; ++j) {           // byte offset of (n, ho, w0 + 4 g, c0 + 4 l31); the epilogue adds the pixel column
            const int ho = tc.h0 + 2 * wave + j;
            ob[j] = (ho < p.Ho) ? (int)((((size_t)(tc.n * p.Ho + ho) * p.Wo + tc.w0 + 4 * g) * p.Cout + tc.c0 + 4 * l31) * 2) : W_OOB;
        }
        ob[2] = p.Wo - tc.w0 - 4 * g;           // pixel columns (relative to this lane's first) that exist
        ob[3] = (tc.n * p.tiles_h + (tc.h0 >> 4)) * p.tiles_w + (tc.w0 >> 5);   // row of the statistics table (uniform)
#pragma unroll
        for (int k = 0; k < W_NSLOT; ++k) {
            int q, pr, pc;
            const bool live = slot_pix(k, q, pr, pc);
            int ih = tc.h0 + pr - p.pad_top, iw = tc.w0 + pc - p.pad_left;
            const bool inb = live && (ih >= 0) && (ih < p.Hl) && (iw >= 0) && (iw < p.Wl);
            if (p.upsample) { ih >>= 1; iw >>= 1; }
            const int sl = (lane & 3) ^ ((q >> 2) & 3);
            vo[k] = inb ? ((ih * p.W + iw) * p.Cin + sl * 8) * 2 : W_OOB;
            inb_mask |= inb ? (1u << k) : 0u;
        }
    };
    auto p_dma = [&](__amdgpu_buffer_rsrc_t rs, const int (&vo)[W_NSLOT], int soff_, int buf, int k0, int cnt) {
        const int soff = __builtin_amdgcn_readfirstlane(soff_);
#pragma unroll
        for (int k = 0; k < W_NSLOT; ++k) {
            if (k < k0 || k >= k0 + cnt) continue;
            const int piece = (k < 4) ? wave + 8 * k : (wave < 7 ? 32 + wave : 38);
#ifdef W_ABL_NOPATCH
            if (p.N != -12345) continue;
#endif
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(patch + buf * W_PATCH + piece * 1024),
                                                     16, vo[k], soff, 0, 0);
        }
    };
    // GroupNorm scale / shift: the whole [Cin][2] row of an image is staged in LDS once per tile (table `sel`), and a chunk's 8
    // pairs per thread are read back right where they are used -- an LDS latency instead of an L2 round trip in front of every
    // activation, and no registers alive across the MFMA blocks
    auto ss_stage = [&](int n, int sel) {
        float* tab = reinterpret_cast<float*>(smem + W_SS) + sel * (W_MAXCIN * 2);
        const float* src = p.ss + (size_t)n * p.Cin * 2;
        for (int k = tid; k < p.Cin * 2; k += 512) tab[k] = src[k];
    };
    float sc[8], sh[8];
    auto ss_fetch = [&](int sel, int ci0) {      // 8 (scale, shift) pairs of this thread's logical channel slot
        const f32x4* sp = reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(smem + W_SS) + sel * (W_MAXCIN * 2) + (ci0 + (lane & 3) * 8) * 2);
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
            const f32x4 v = sp[q4];
            sc[2 * q4] = v[0]; sh[2 * q4] = v[1]; sc[2 * q4 + 1] = v[2]; sh[2 * q4 + 1] = v[3];
        }
    };
    // (side output) the activated values of a stage's 5 slots stay in registers (av) with their store offsets (ao) until the stage's
    // weight DMA has been issued: the 5 stores then are the YOUNGEST vector-memory operations at the next barrier, whose counted wait
    // lets them fly on (a store in front of the DMA would have to be acknowledged before the in-order wait for the DMA returns:
    // measured +0.09 ms per launch).  ALWAYS exactly 5 store instructions per wave (dead slots: out-of-range offset, dropped by the
    // descriptor's bounds check): the counted waits depend on it.
    auto p_activate = [&](unsigned inb_mask, int buf) {   // padding pixels were written as zeros by the DMA and stay zero
#pragma unroll
        for (int k = 0; k < W_NSLOT; ++k) {
            int q, pr, pc;
            const bool live = slot_pix(k, q, pr, pc) && (k < 4 || wave < 7) && ((inb_mask >> k) & 1u);
            if (!live) continue;
            unsigned char* dst = patch + buf * W_PATCH + q * 64 + (((lane & 3) ^ ((q >> 2) & 3)) << 4);
            u32x4 v = *reinterpret_cast<const u32x4*>(dst);
            if (p.act == MAS_ACT_AFFINE_SILU) {
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) v[q4] = act_pair_bf16<true>(v[q4], f32x2{sc[2 * q4], sc[2 * q4 + 1]}, f32x2{sh[2 * q4], sh[2 * q4 + 1]});
            } else {
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) v[q4] = act_pair_bf16<false>(v[q4], f32x2{sc[2 * q4], sc[2 * q4 + 1]}, f32x2{sh[2 * q4], sh[2 * q4 + 1]});
            }
            *reinterpret_cast<u32x4*>(dst) = v;
        }
    };

    // ---- weight stage: 3 tap tiles of 8 KiB; wave w moves piece w (rows 16w..16w+15) of each ---------------------------------
    const int wlane = lane * 16;
    const int wstride = p.Cout_pad * 64;                   // one tap-step of the packed image ([chunk32][tap][Cout_pad][64 B])
    auto w_issue = [&](int t0, int c0, int sel) {          // taps t0, t0+1, t0+2 (t = chunk * 9 + tap) of cout tile c0
        // (readfirstlane: when a uniform term of this sum has been spilled through a VGPR, hipcc otherwise legalises the SGPR
        //  operand with a waterfall loop around every DMA instruction)
        const int soff = __builtin_amdgcn_readfirstlane(t0 * wstride + c0 * 64 + wave * 1024);
        unsigned char* dst = wbuf + sel * W_WSTAGE + wave * 1024;
#ifdef W_ABL_NOW
        if (p.N != -12345) return;
#endif
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (__attribute__((address_space(3))) void*)dst, 16, wlane, soff, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (__attribute__((address_space(3))) void*)(dst + W_WT), 16, wlane, soff + wstride, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (__attribute__((address_space(3))) void*)(dst + 2 * W_WT), 16, wlane, soff + 2 * wstride, 0, 0);
    };

    // ---- per-lane fragment addressing ----------------------------------------------------------------------------------------
    // A (weights): row = 32 i + l31, logical slot 2 kk + g, key (row >> 2) & 3 = (l31 >> 2) & 3 for every i
    const int a_off = l31 * 64 + ((g ^ ((l31 >> 2) & 3)) << 4);          // kk = 0; kk = 1 is a_off ^ 32
    // B (patch): pixel P = (2 wave + j) * 34 + l31 + kh * 34 + kw, logical slot 2 kk + g, key (P >> 2) & 3
    const int pj0 = (2 * wave) * W_PWL + l31;
    auto b_addr = [&](int P) { return P * 64 + (((g ^ (P >> 2)) & 3) << 4); };   // kk = 0; kk = 1 is ^ 32

    // ---- prologue ----------------------------------------------------------------------------------------------------------------
    // Tile walk.  Work-group b runs on XCD b % 8 (observed dispatch order; speed only, never correctness).  With the plain static stride
    // (b, b + G, ...) the 32 CUs of an XCD work on tiles 8 apart: no two of them share a halo, and every 18x34 patch (1.195x its tile)
    // comes over the fabric.  xcd_bands: XCD x owns the contiguous eighth [T x / 8, T (x + 1) / 8) of the tile list and its work-groups
    // walk it in order, so the tiles in flight on one XCD are spatial neighbours (a whole image of a 256^2 map) and their halos meet in
    // that XCD's L2 (profiles/r06_xcd_bands.txt).  Same registers as before: `step` replaces gridDim.x, `tile_end` replaces total_tiles.
    int tile = blockIdx.x;                      // grid <= total_tiles
    int step = (int)gridDim.x, tile_end = total_tiles;
    if (p.xcd_bands && (gridDim.x & 7) == 0) {
        const int x = blockIdx.x & 7;
        step = (int)(gridDim.x >> 3);
        tile = (int)(((long long)total_tiles * x) >> 3) + (int)(blockIdx.x >> 3);
        tile_end = (int)(((long long)total_tiles * (x + 1)) >> 3);
    }
    int c0_cur, c0_nxt, ob_cur[4];
    // ONE plan / image descriptor: the current tile's until its last chunk-B patch has been issued (stage 1 of the last pair),
    // the next tile's from stage 2 of the last pair on (only the in-bounds masks of both tiles are live at the same time)
    int vo[W_NSLOT];
    unsigned inb_cur, inb_nxt;
    int n_cur, n_nxt;                            // image index: the GroupNorm scale/shift rows of the prologue
    __amdgpu_buffer_rsrc_t rs_x;
    {
        const Tile t0 = decode(tile);
        make_plan(t0, vo, inb_cur, ob_cur);
        c0_cur = c0_nxt = t0.c0; n_cur = n_nxt = t0.n;
        *reinterpret_cast<u32x4*>(smem + W_NEXT + tid * 16) = u32x4{(unsigned)ob_cur[0], (unsigned)ob_cur[1], (unsigned)ob_cur[2], (unsigned)ob_cur[3]};
        rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(p.x) + (size_t)t0.n * img_bytes, 0, (unsigned)img_bytes, 0x00020000);
    }
    inb_nxt = inb_cur;
    {   // bias -> LDS (read back in the epilogue through lgkmcnt: a VMEM load there would wait behind the tile's own stores)
        float* bl = reinterpret_cast<float*>(smem + W_BIAS);
        for (int c = tid; c < p.Cout; c += 512) bl[c] = p.bias ? p.bias[c] : 0.0f;
    }
    w_issue(0, c0_cur, 0);
    p_dma(rs_x, vo, 0, 0, 0, W_NSLOT);
    int ss_sel = 0;                              // table of the CURRENT tile's image
    if constexpr (ACT) {
        ss_stage(n_cur, 0);
        W_WAIT_BARRIER(0);                       // the raw patch of chunk 0 has landed for every wave, the table is visible
        ss_fetch(0, 0);
        p_activate(inb_cur, 0);
    }
    // fused GroupNorm statistics (p.stats): the epilogue leaves per-wave partial sums in LDS; after the next work-group barrier 256
    // threads add the 8 waves in a fixed order and write the tile's row of the statistics table (no atomics: deterministic)
    // (the pending tile's table row / cout offset wait in LDS, not in registers: this kernel has no SGPRs to spare)
    auto stats_flush = [&]() {
        if (tid < 256) {
            const float* sl = reinterpret_cast<const float*>(smem + W_STAT);
            const int* meta = reinterpret_cast<const int*>(smem + W_STAT + 8 * 128 * 2 * 4);
            float t = 0.0f;
#pragma unroll
            for (int w = 0; w < 8; ++w) t += sl[w * 256 + tid];
            p.stats[((size_t)meta[0] * p.Cout + meta[1]) * 2 + tid] = t;
        }
    };
    bool stores_in_flight = false;               // the previous tile's 32 epilogue stores may still be in flight at the first wait
    const int n_pairs = p.n_chunks >> 1;
#ifdef W_TIMELINE
    int tl_iter = 0;
#endif

    for (;;) {
        const int next_tile = tile + step;
        const bool has_next = next_tile < tile_end;
        WTS(0);

        f32x16 acc[4][2];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

        for (int pair = 0; pair < n_pairs; ++pair) {
            const bool last_pair = pair + 1 == n_pairs;
            const int ciA = pair * 64;                        // first channel of chunk A (even) of this pair; B = +32
            w_static_for(std::make_integer_sequence<int, 6>{}, [&](auto s_c) {
                constexpr int s = decltype(s_c)::value;
                constexpr int cb = s / 3, kh = s % 3;         // chunk A / B of the pair (= patch buffer), filter row
                constexpr int wsel = s & 1;
                if (pair < 2) WTS(1 + 3 * (pair * 6 + s));
                // ---- barrier: this stage's weights, and the patch chunk it reads, are visible; the previous stage's buffers are free.
                //      vmcnt allowance = VMEM operations issued AFTER the weight DMA in the previous stage
                {
                    constexpr int khp = (kh + 2) % 3;         // filter row of the previous stage
                    constexpr int allow = khp == 0 ? (ACT ? 5 : 3) : (khp == 1 ? (ACT ? 0 : 2) : 0);
                    if (s == 0 && pair == 0 && stores_in_flight) {
                        w_wait_barrier(32); stores_in_flight = false;
                        if constexpr (STATS) stats_flush();      // the finished tile's statistics: one store, older than this stage's weight DMA
                    } else w_wait_barrier(allow);      // one store, older than this stage's weight DMA
                }
                if (pair < 2) WTS(2 + 3 * (pair * 6 + s));
                // ---- the next tile's plan, and GroupNorm(+SiLU) in place on the NEXT chunk's raw patch (all of it has landed: allowance 0 above)
                auto plan_blk = [&]() {
                    if (s == 2 && last_pair && has_next) {        // the next tile's plan (used by the chunk-B stages 3, 4); without a next
                        const Tile nt = decode(next_tile);        // tile the current one is re-fetched, harmlessly
                        int ob_n[4];
                        make_plan(nt, vo, inb_nxt, ob_n);
                        *reinterpret_cast<u32x4*>(smem + W_NEXT + tid * 16) = u32x4{(unsigned)ob_n[0], (unsigned)ob_n[1], (unsigned)ob_n[2], (unsigned)ob_n[3]};
                        c0_nxt = nt.c0; n_nxt = nt.n;
                        if constexpr (ACT) ss_stage(nt.n, ss_sel ^ 1);      // visible after the next barrier, first read three stages later
                        rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(p.x) + (size_t)nt.n * img_bytes, 0,
                                                                 (unsigned)img_bytes, 0x00020000);
                    }
                };
                auto act_blk = [&]() {
                    if constexpr (ACT) {
                        if (kh == 2) {
                            if (cb == 0) ss_fetch(ss_sel, ciA + 32);
                            else if (last_pair) ss_fetch(ss_sel ^ 1, 0);
                            else ss_fetch(ss_sel, ciA + 64);
                            const bool nxt = cb == 1 && last_pair;                 // chunk 0 of the NEXT tile (its plan is in `vo` since stage 2)
                            // (without a next tile this stage re-activates the current tile's chunk 0 with a scale/shift table that was never
                            //  staged -- harmless: the convolution never reads it)
                            p_activate(nxt ? inb_nxt : inb_cur, cb ^ 1);
                            asm volatile("" ::: "memory");
                        }
                    }
                };
                plan_blk(); act_blk();
                // ---- weight DMA for the next stage
                {
                    const int tn = (s < 5) ? (pair * 2 + (s + 1) / 3) * 9 + ((s + 1) % 3) * 3 : (last_pair ? 0 : (pair + 1) * 18);
                    const int c0n = (s == 5 && last_pair) ? c0_nxt : c0_cur;
                    w_issue(tn, c0n, wsel ^ 1);
                }
                asm volatile("" ::: "memory");                // VMEM order = source order: the counted waits depend on it
                // ---- patch DMA for the next chunk (3 pieces in the kh = 0 stage, 2 in the kh = 1 stage)
                if (kh < 2 && !(ACT && kh == 1)) {
                    // plain: 3 + 2 pieces over the kh = 0, 1 stages.  With the prologue all 5 in the kh = 0 stage: the activation that
                    // opens the kh = 2 stage then waits for DMA that has had two stages to land
                    constexpr int k0 = (ACT || kh == 0) ? 0 : 3, cnt = ACT ? 5 : (kh == 0 ? 3 : 2);
                    if (cb == 0) {                            // chunk B of this pair -> buffer 1
                        p_dma(rs_x, vo, (ciA + 32) * 2, 1, k0, cnt);
                    } else {                                  // chunk A of the next pair, or chunk 0 of the next tile -> buffer 0
                        p_dma(rs_x, vo, last_pair ? 0 : (ciA + 64) * 2, 0, k0, cnt);
                    }
                }
                asm volatile("" ::: "memory");
                if (pair < 2) WTS(3 + 3 * (pair * 6 + s));
                // ---- 3 taps x 2 k-steps of 8 MFMAs; fragment reads software-pipelined one k-step ahead
                {
                    const unsigned char* wb = wbuf + wsel * W_WSTAGE;
                    const unsigned char* pb = patch + cb * W_PATCH;
                    bf16x8 afr[2][4], bfr[2][2];
                    auto ld_k = [&](int n, int b) {           // n = 0..5: kw = n >> 1, kk = n & 1
                        const int kw = n >> 1, kx = (n & 1) << 5;
#pragma unroll
                        for (int j = 0; j < 2; ++j)
                            bfr[b][j] = *reinterpret_cast<const bf16x8*>(pb + (b_addr(pj0 + (j + kh) * W_PWL + kw) ^ kx));
#ifdef W_ABL_HALF_A   // timing experiment only (wrong results): 2 of the 4 weight fragments per k-step are not re-read -- 0.5 instead of 0.75
#pragma unroll        // LDS reads per MFMA: what a 128 x 128 register tile would buy (tools/experiments/gpu_r3_41.sh)
                        for (int i = 0; i < 2; ++i)
                            afr[b][i] = *reinterpret_cast<const bf16x8*>(wb + kw * W_WT + i * 2048 + (a_off ^ kx));
                        if (p.N == -12345) { afr[b][2] = afr[b][0]; afr[b][3] = afr[b][1]; }
#else
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                            afr[b][i] = *reinterpret_cast<const bf16x8*>(wb + kw * W_WT + i * 2048 + (a_off ^ kx));
#endif
                    };
                    ld_k(0, 0);
#pragma unroll
                    for (int n = 0; n < 6; ++n) {
                        if (n + 1 < 6) ld_k(n + 1, (n + 1) & 1);
#pragma unroll
                        for (int i = 0; i < 4; ++i)
#pragma unroll
                            for (int j = 0; j < 2; ++j) mma16(acc[i][j], bfr[n & 1][j], afr[n & 1][i]);   // D[pixel][cout]: rows (registers) = pixels, column (lane) = cout
                    }
                    __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);          // DS reads of k-step 0
#pragma unroll
                    for (int n = 0; n + 1 < 6; ++n) {
                        __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);
                        __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
                    }
                    __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
                }
            });
        }

        WTS(60);
#ifdef W_ABL_NOEPI     // timing experiment only
        if (p.N != -12345) {
            float t = 0.0f;
            for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) t += acc[i][j][r];
            if (t == 123.456f) reinterpret_cast<float*>(p.y)[0] = t;
        } else
#endif
        // ---- epilogue.  The weight rows of a cout tile are stored PERMUTED (LDS row 32 i + l holds cout 4 l + i, see
        //      mas_pack_conv_weight_layout), so a lane owns 4 CONSECUTIVE couts (one of each of its 4 accumulator tiles) of the
        //      16 pixels of each register block: one 8-byte store per pixel and lane, and the 32 lanes of a half-wave write one
        //      whole 256-byte NHWC pixel row (8 lanes per 64-byte... every store instruction = two complete rows, perfectly coalesced).
        //      (The MFMA-natural alternative -- lane = pixel, 8 couts = 16 B per lane -- touches 32 different 128-byte lines per
        //      store instruction: measured 0.08 ms of 0.64 on the 128->128 @256^2 launch, profiles/r02_wide_store_ablation.txt.)
        {
            const int row_bytes = p.Cout * 2;
            const f32x4 bv = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(smem + W_BIAS) + c0_cur + 4 * l31);
            // order: residual loads of row 0 | pack row 0 | residual loads of row 1 | stores of row 0 | pack + stores of row 1 --
            // no load is waited for behind a store (in-order vmcnt would add the store's round trip), 32 packed + 32 residual
            // registers at most beside the 64 accumulators of the other row
            u32x2 rv[16], o0[16];
            auto res_load = [&](int j) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int pc = (r & 3) + 8 * (r >> 2);           // pixel column (+ 4 g, folded into ob_cur)
                    rv[r] = __builtin_amdgcn_raw_buffer_load_b64(rs_r, pc < ob_cur[2] ? ob_cur[j] : W_OOB, pc * row_bytes, 0);
                }
            };
            // fused statistics: per lane the sum and the sum of squares of its 4 couts over its pixels, as two packed-fp32 pairs each
            // (v_pk_add_f32 / v_pk_fma_f32: 4 VALU per pixel).  Of the fp32 values BEFORE the bf16 rounding: closer to what
            // torch.nn.GroupNorm computes on the reference's fp32 activations than the statistics of the stored tensor, and free of
            // the unpacking the round-2 version paid (12 VALU per pixel, +0.07 ms per launch).
            f32x2 st_s[2] = {f32x2{0.0f, 0.0f}, f32x2{0.0f, 0.0f}}, st_q[2] = {f32x2{0.0f, 0.0f}, f32x2{0.0f, 0.0f}};
            if constexpr (STATS) {
                // ragged tiles only (wave-uniform branch): a pixel outside the map must not count -- its accumulators are set to
                // -bias, so that (with the residual load of an out-of-range offset returning 0) its value is exactly 0; its store is
                // dropped by the bounds check anyway.  Full tiles (all but the last row / column of tiles) skip this.
                if (!__all(ob_cur[2] > 27 && ob_cur[0] != W_OOB && ob_cur[1] != W_OOB)) {
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int pc = (r & 3) + 8 * (r >> 2);
                            if (!((pc < ob_cur[2]) && (ob_cur[j] != W_OOB))) {
#pragma unroll
                                for (int i = 0; i < 4; ++i) acc[i][j][r] = -bv[i];
                            }
                        }
                }
            }
            auto pack = [&](int j, int r) {
                float v[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = acc[i][j][r] + bv[i];
                if constexpr (RES) {
                    const bf16_t* rb = reinterpret_cast<const bf16_t*>(&rv[r]);
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] += (float)rb[i];
                }
                if constexpr (STATS) {          // (pixels past a ragged edge were zeroed above: they add nothing)
                    const f32x2 a = {v[0], v[1]}, b = {v[2], v[3]};
                    st_s[0] += a; st_s[1] += b;
                    st_q[0] = a * a + st_q[0]; st_q[1] = b * b + st_q[1];
                }
                u32x2 o;
                bf16_t* ob = reinterpret_cast<bf16_t*>(&o);
#pragma unroll
                for (int i = 0; i < 4; ++i) ob[i] = (bf16_t)v[i];
                return o;
            };
#ifndef W_ST_AUX
#define W_ST_AUX 0
#endif
            auto store = [&](int j, int r, u32x2 o) {
                const int pc = (r & 3) + 8 * (r >> 2);
                const bool ok = (pc < ob_cur[2]) && (ob_cur[j] != W_OOB);
                __builtin_amdgcn_raw_buffer_store_b64(o, rs_y, ok ? ob_cur[j] : W_OOB, pc * row_bytes, W_ST_AUX);
            };
            if constexpr (RES) {
                res_load(0);
#pragma unroll
                for (int r = 0; r < 16; ++r) o0[r] = pack(0, r);
                asm volatile("" ::: "memory");
                res_load(1);
                asm volatile("" ::: "memory");
#pragma unroll
                for (int r = 0; r < 16; ++r) store(0, r, o0[r]);
                asm volatile("" ::: "memory");
#pragma unroll
                for (int r = 0; r < 16; ++r) store(1, r, pack(1, r));
            } else {                             // nothing to load: pack and store pixel by pixel (no staging registers)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) store(j, r, pack(j, r));
            }
            stores_in_flight = true;
            if constexpr (STATS) {               // the two half-waves hold different pixels of the same 4 couts; then one LDS row per wave
                float* sl = reinterpret_cast<float*>(smem + W_STAT) + wave * 256 + (4 * l31) * 2;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float ss_ = st_s[i >> 1][i & 1], qq_ = st_q[i >> 1][i & 1];
                    const float a = ss_ + __shfl_xor(ss_, 32), b = qq_ + __shfl_xor(qq_, 32);
                    if (g == 0) { sl[2 * i] = a; sl[2 * i + 1] = b; }
                }
                if (tid == 0) { int* meta = reinterpret_cast<int*>(smem + W_STAT + 8 * 128 * 2 * 4); meta[0] = ob_cur[3]; meta[1] = c0_cur; }
            }
        }
        WTS(61);
#ifdef W_TIMELINE
        ++tl_iter;
#endif
        if (!has_next) break;
        tile = next_tile; c0_cur = c0_nxt; n_cur = n_nxt; inb_cur = inb_nxt; ss_sel ^= 1;
        { const u32x4 nx = *reinterpret_cast<const u32x4*>(smem + W_NEXT + tid * 16); ob_cur[0] = (int)nx[0]; ob_cur[1] = (int)nx[1]; ob_cur[2] = (int)nx[2]; ob_cur[3] = (int)nx[3]; }
    }
    if constexpr (STATS) {                       // the last tile's statistics
        __syncthreads();
        stats_flush();
    }
}

template <bool ACT, bool RES, bool STATS>
int launch_wide(const WideParams& p, hipStream_t s) {
    auto kern = conv3x3_wide_kernel<ACT, RES, STATS>;
    static mas_devmask_t attr_mask{0};
    unsigned long long attr_bit;
    if (mas_attr_needed(attr_mask, &attr_bit)) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, W_LDS) != hipSuccess)
            MAS_FAIL(MAS_ELAUNCH, "conv3x3_wide: cannot set dynamic LDS size %d", W_LDS);
        mas_attr_done(attr_mask, attr_bit);
    }
    const long long tiles = (long long)p.N * p.tiles_h * p.tiles_w * p.n_ct;
    long long resident = 4LL * mas_num_cus();              // 4x oversubscription: see conv_fwd.hip launch_v
    static const int wgs_per_cu = mas_env_int("MAS_CONV_WGS_PER_CU", 0);
    if (wgs_per_cu > 0) resident = (long long)wgs_per_cu * mas_num_cus();
    const unsigned blocks = (unsigned)(tiles < resident ? tiles : resident);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(512), W_LDS, s, p);
    MAS_CHECK_LAUNCH("conv3x3_wide");
    return MAS_OK;
}

}  // namespace

// Does this convolution take the wide kernel (and therefore the MAS_WLAYOUT_K32 weight image)?
bool mas_conv3x3_wide_eligible(const MasConvDesc* d) {
    static const int mode = mas_env_int("MAS_CONV_WIDE", 1);
    if (!mode) return false;
    if (d->ks != 3 || d->stride != 1 || d->in_dtype != MAS_BF16 || d->out_dtype != MAS_BF16) return false;
    if (d->Cin % 64 != 0 || d->Cout % 128 != 0) return false;
    static const int any_width = mas_env_int("MAS_CONV_WIDE_ANY_WIDTH", 0);   // tests: ragged tile columns at small sizes
    if (d->Cout > W_MAXCOUT || (d->act != MAS_ACT_NONE && d->Cin > W_MAXCIN)) return false;
    if (!any_width && (d->Wo < 32 || 4 * d->Wo < 3 * 32 * mas_cdiv(d->Wo, 32))) return false;   // < 75 % of the 32-pixel tile rows used
    const long long img_bytes = (long long)d->H * d->W * d->Cin * 2;
    const long long out_bytes = (long long)d->N * d->Ho * d->Wo * d->Cout * 2;
    if (img_bytes >= 0x7fffffffLL || out_bytes >= 0x7fffffffLL) return false;
    const long long tiles = (long long)d->N * mas_cdiv(d->Ho, 16) * mas_cdiv(d->Wo, 32) * (d->Cout / 128);
    // one tile per CU is enough: 512 -> 512 @32^2 (256 tiles) 0.143 ms on the stream kernel, 0.127 here; step -0.3 ms (profiles/r03_ab_wide_min.txt)
    static const int min_per_cu = mas_env_int("MAS_CONV_WIDE_MIN_TILES_PER_CU", 1);
    if (tiles < (long long)min_per_cu * mas_num_cus() || tiles > 0x7fffffffLL) return false;
    const long long dmax = std::max<long long>(d->Cout / 128, std::max(mas_cdiv(d->Ho, 16), mas_cdiv(d->Wo, 32)));
    if (tiles * dmax >= 0x100000000LL) return false;       // the kernel's multiply-high tile decode is exact below this
    return true;
}

// rows per image of the statistics table the wide kernel can fill for this convolution (its 16x32-pixel tiles)
int mas_conv3x3_wide_stat_rows(const MasConvDesc* d) { return mas_cdiv(d->Ho, 16) * mas_cdiv(d->Wo, 32); }

int mas_conv3x3_wide_launch(const MasConvDesc* d, const void* x, const float* scale_shift, const void* w_packed, const float* bias,
                            const void* residual, void* y, float* stats, hipStream_t s) {
    WideParams p;
    p.dbg = nullptr;
    p.stats = stats;
#ifdef W_TIMELINE
    if (const char* e = getenv("MAS_DBG_PTR")) p.dbg = reinterpret_cast<unsigned long long*>(strtoull(e, nullptr, 0));
#endif
    p.x = (const unsigned char*)x; p.ss = scale_shift; p.w = (const unsigned char*)w_packed; p.bias = bias;
    p.res = (const unsigned char*)residual; p.y = (unsigned char*)y;
    p.N = d->N; p.H = d->H; p.W = d->W; p.Cin = d->Cin; p.Ho = d->Ho; p.Wo = d->Wo; p.Cout = d->Cout;
    p.Hl = d->upsample ? 2 * d->H : d->H; p.Wl = d->upsample ? 2 * d->W : d->W;
    p.pad_top = d->pad_top; p.pad_left = d->pad_left; p.upsample = d->upsample; p.act = d->act;
    p.n_chunks = d->Cin / 32; p.Cout_pad = mas_roundup(d->Cout, 128);
    p.tiles_h = mas_cdiv(d->Ho, 16); p.tiles_w = mas_cdiv(d->Wo, 32); p.n_ct = d->Cout / 128;
    auto magic = [](int dv) { return (unsigned)((0x100000000ULL + (unsigned)dv - 1) / (unsigned)dv); };   // (d = 1 handled in the kernel)
    p.m_ct = magic(p.n_ct); p.m_tw = magic(p.tiles_w); p.m_th = magic(p.tiles_h);
    static const int xcd_bands = mas_env_int("MAS_CONV_XCD_BANDS", 1);
    {   // the banded walk needs every band at least as long as the number of work-groups that walk it
        const long long tiles = (long long)p.N * p.tiles_h * p.tiles_w * p.n_ct;
        static const int wgs_per_cu = mas_env_int("MAS_CONV_WGS_PER_CU", 0);
        const long long resident = (long long)(wgs_per_cu > 0 ? wgs_per_cu : 4) * mas_num_cus();
        const long long blocks = tiles < resident ? tiles : resident;
        p.xcd_bands = (xcd_bands && blocks % 8 == 0 && tiles / 8 >= blocks / 8) ? 1 : 0;
    }
    if (stats) {
        if (d->act != MAS_ACT_NONE) return residual ? launch_wide<true, true, true>(p, s) : launch_wide<true, false, true>(p, s);
        return residual ? launch_wide<false, true, true>(p, s) : launch_wide<false, false, true>(p, s);
    }
    if (d->act != MAS_ACT_NONE) return residual ? launch_wide<true, true, false>(p, s) : launch_wide<true, false, false>(p, s);
    return residual ? launch_wide<false, true, false>(p, s) : launch_wide<false, false, false>(p, s);
}
