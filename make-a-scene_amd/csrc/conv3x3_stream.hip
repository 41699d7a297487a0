// 3x3 / stride-1 / bf16 implicit-GEMM convolution, "stream" structure, for the shapes that carry the FLOPs
// (Cin % 128 == 0, Cout % 128 == 0: every ResnetBlock conv, every Upsample conv and all of their data gradients --
// reference models/modules.py:49,93,100,113 and autograd of the same sites).  Same math, operand layouts and packed
// weight image as conv_fwd.hip (which stays the general kernel); what changes is the SCHEDULE.
//
// Round 1's kernel ran load -> MFMA -> store phases in lock-step: its ablation (profiles/r01_conv_fwd_ablation.txt) showed
// patch loads (+0.16 ms), epilogue stores (+0.14 ms) and weight DMA (+0.07 ms) ADDING to a 0.39 ms MFMA core.  Here a
// tile is a continuous stream of "tap-steps" (one tap x one 64-channel chunk = 16 MFMAs per wave), cut into stages of two
// steps with one work-group barrier per stage, and every HBM-facing operation is issued INSIDE a stage, for a later one:
//   * the halo patch is double-buffered in LDS (2 x 41 KiB): chunk c+1 (or the next tile's chunk 0) is fetched while the
//     MFMAs of chunk c run.  Without a GroupNorm prologue (all data gradients, Upsample convs) the patch is filled by
//     LDS-DMA through a buffer descriptor (`buffer_load_dwordx4 ... lds`): no VGPRs, no VALU, no ds_write, and the
//     descriptor's bounds check writes the zero padding (out-of-range lanes return 0).  With the GN+SiLU prologue it is
//     staged through registers, 3 slots per stage, and committed (affine + SiLU + ds_write_b128) two stages later, beside
//     the MFMAs of the other waves, instead of in a phase of its own between two barriers;
//   * the output tile of tile k is packed to bf16 in registers and its 8 stores are issued one per stage during tile k+1;
//   * weight stages (2 steps = 32 KiB) arrive by LDS-DMA one stage ahead (double buffer).
// All waits are COUNTED (`s_waitcnt vmcnt(N)` + raw `s_barrier`): in-order VMEM retirement finishes the weight DMA of the
// next stage and leaves the N younger patch / store operations in flight across the barrier.
// LDS: 2 x 41 KiB patch + 2 x 32 KiB weights = 146 KiB, one 512-thread work-group per CU, 2 waves per SIMD.
#include "mas_common.h"
#include <utility>

namespace {

template <int... I, typename F>
__device__ __forceinline__ void s_static_for(std::integer_sequence<int, I...>, F&& f) { (f(std::integral_constant<int, I>{}), ...); }

struct StreamParams {
    unsigned long long* dbg;                   // -DS_TIMELINE builds only (tools/timeline_stream.py): s_memtime stamps of one work-group
    const unsigned char* x; const float* ss; const unsigned char* w; const float* bias; const unsigned char* res; unsigned char* y;
    int N, H, W, Cin, Ho, Wo, Cout;
    int Hl, Wl, pad_top, pad_left, upsample, act;
    int n_chunks, Cout_pad, tiles_h, tiles_w, n_ct;
    int xcd_bands;                             // 1: every XCD walks its own contiguous eighth of the tile list (conv3x3_wide.hip has the reasoning)
};

constexpr int S_PWL = 18;                      // patch pitch in pixels ((16-1)+3)
// Patch geometry by tile height TH (16 rows: 324 pixels x 128 B = 41 DMA pieces of 1 KiB, 6 per wave; 8 rows: 180 pixels = 23 pieces,
// 3 per wave).  (A traits struct, not constexpr locals in the kernel: locals that appear in the parameter types of the kernel's
// lambdas make hipcc drop the kernel's host-side handle without a diagnostic -- the launch then fails to link at load time.)
template <int TH> struct SGeo {
    static constexpr int NPIX = (TH + 2) * S_PWL;
    static constexpr int NPIECE = (NPIX * 128 + 1023) / 1024;      // 41 | 23
    static constexpr int PATCH = NPIECE * 1024;
    static constexpr int NSLOT = (NPIECE + 7) / 8;                 // 6 | 3 DMA pieces per wave per chunk
    static constexpr int NJ = TH / 8;                              // 32-pixel fragments per wave: 2 | 1
    static constexpr int PCNT = NSLOT / 3;                         // patch pieces a wave issues per stage (plain path): 2 | 1
    static constexpr int NST = 4 * NJ;                             // deferred stores per lane per tile: 8 | 4, one per stage
};
constexpr int S_WT = 128 * 128;                // one tap-step weight tile: 128 couts x 128 B
constexpr int S_WSTAGE = 2 * S_WT;
constexpr int S_OOB = (int)0x80000000;         // voffset beyond any descriptor's num_records

#ifndef S_ABL_NOBARRIER
#define S_WAIT_BARRIER(N) do { asm volatile("s_waitcnt vmcnt(" #N ") lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); \
                               asm volatile("" ::: "memory"); } while (0)
#else   // timing experiment only (races): what do the 9 work-group barriers per pair cost?
#define S_WAIT_BARRIER(N) do { asm volatile("s_waitcnt vmcnt(" #N ") lgkmcnt(0)" ::: "memory"); asm volatile("" ::: "memory"); } while (0)
#endif

#ifdef S_TIMELINE      // stamps only where no ds_read is in flight (s_memtime returns through lgkmcnt)
#define STS(id) do { if (lane == 0 && blockIdx.x == 100 && tl_iter >= 1 && tl_iter < 3 && p.dbg) \
                         p.dbg[((tl_iter - 1) * 8 + wave) * 64 + (id)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define STS(id) do {} while (0)
#endif

__device__ __forceinline__ void s_wait_barrier(int n) {   // n is compile-time after unrolling, or selected by a uniform branch
    switch (n) {
        case 0: S_WAIT_BARRIER(0); break;
        case 1: S_WAIT_BARRIER(1); break;
        case 2: S_WAIT_BARRIER(2); break;
        case 3: S_WAIT_BARRIER(3); break;
        case 4: S_WAIT_BARRIER(4); break;
        case 7: S_WAIT_BARRIER(7); break;
        case 8: S_WAIT_BARRIER(8); break;
        default: S_WAIT_BARRIER(0); break;
    }
}

__device__ __forceinline__ bf16x8 s_ld_frag(const unsigned char* row, int xs, int kk) {
    // 8 consecutive K elements (k = 16 kk + 8 g + 0..7) of a 128-byte row whose 16-byte slots are XOR-swizzled; xs = (g ^ swizzle) << 4
    return *reinterpret_cast<const bf16x8*>(row + (xs ^ (kk << 5)));
}

// ACT: GroupNorm(+SiLU) prologue -> register-staged patch; otherwise LDS-DMA patch.  DEFER: deferred-store epilogue.
// TH: tile height, 16 (default) or 8.  Round 4: the 16x16-pixel level of VQ-IMG (512 -> 512 at batch 32) has 128 tiles of 16x16 pixels
// x 128 couts -- half the chip idle (VERDICT r3 #4i: 0.70 PF).  With TH = 8 a wave owns 64 couts x 32 pixels (one pixel fragment
// instead of two), the patch is 10 x 18 pixels (23 DMA pieces, 3 per wave, one per stage), the tile has 4 deferred stores per lane
// instead of 8 -- same stage program, same counted waits with the smaller per-stage allowances, twice the tiles.  Prologue-free only.
template <bool ACT, bool DEFER, int TH>
__global__ __launch_bounds__(512, 2) void conv3x3_stream_kernel(StreamParams p) {
    static_assert(TH == 16 || (TH == 8 && !ACT), "the 8-row tile exists for the prologue-free path only");
    using G = SGeo<TH>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* const patch = smem;                    // [2][G::PATCH]
    unsigned char* const wbuf = smem + 2 * G::PATCH;       // [2][S_WSTAGE]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave_c = wave & 1, wave_p = wave >> 1;      // 2 x 64 couts, 4 x 64 pixels (4 tile rows)
    const int g = lane >> 5, l31 = lane & 31;

    const size_t img_bytes = (size_t)p.H * p.W * p.Cin * 2;
    const unsigned out_bytes = (unsigned)((size_t)p.N * p.Ho * p.Wo * p.Cout * 2);
    const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, out_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_r = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(p.res ? p.res : p.y), 0, p.res ? out_bytes : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc(p.bias ? (void*)p.bias : (void*)p.y, 0, p.bias ? (unsigned)(p.Cout * 4) : 0u, 0x00020000);

    // ---- tiles: persistent work-group, static stride -------------------------------------------------------------
    const int total_tiles = p.N * p.tiles_h * p.tiles_w * p.n_ct;
    struct Tile { int n, h0, w0, c0; };
    auto decode = [&](int t) {
        Tile tc;
        const int ct = t % p.n_ct; t /= p.n_ct;
        const int tw_i = t % p.tiles_w; t /= p.tiles_w;
        const int th_i = t % p.tiles_h; tc.n = t / p.tiles_h;
        tc.c0 = ct * 128; tc.h0 = th_i * TH; tc.w0 = tw_i * 16;
        return tc;
    };

    // ---- patch staging plan ------------------------------------------------------------------------------------
    // slot i of this thread: DMA piece pc_i = wave + 8 i (waves 1..7 repeat piece wave+32 as their 6th: every wave issues the
    // same number of VMEM operations), patch pixel q = 8 piece + (lane >> 3), physical 16-byte slot lane & 7 (DMA: LDS image is
    // lane-linear, the channel-slot swizzle goes on the SOURCE address) or logical slot lane & 7 (register path: swizzle on the
    // ds_write address, so one thread needs one set of 8 scale/shift pairs).
    const int lrow = lane >> 3;
    auto slot_pix = [&](int i, int& pr, int& pc) -> bool {   // patch pixel of slot i (recomputed where needed: no register arrays)
        const int piece = (wave + 8 * i < G::NPIECE) ? wave + 8 * i : wave + 8 * (i - 1);
        const int q = piece * 8 + lrow;
        pr = (q * 3641) >> 16;                  // q / 18 for q < 3641
        pc = q - pr * S_PWL;
        return q < G::NPIX;                       // false: dead pixels of the last piece -- always out of range
    };
    auto make_plan = [&](const Tile& tc, int* vo, unsigned& inb_mask) {
        inb_mask = 0;
#pragma unroll
        for (int i = 0; i < G::NSLOT; ++i) {
            int pr, pc;
            const bool live = slot_pix(i, pr, pc);
            int ih = tc.h0 + pr - p.pad_top, iw = tc.w0 + pc - p.pad_left;
            const bool inb = live && (ih >= 0) && (ih < p.Hl) && (iw >= 0) && (iw < p.Wl);
            if (p.upsample) { ih >>= 1; iw >>= 1; }
            const int sl = (lane & 7) ^ ((pc >> 1) & 7);
            vo[i] = inb ? ((ih * p.W + iw) * p.Cin + sl * 8) * 2 : S_OOB;
            inb_mask |= inb ? (1u << i) : 0u;
        }
    };

    // ---- weight stage: 2 tap-step tiles (32 KiB), 4 x 1-KiB LDS-DMA pieces per wave ------------------------------
    // (MUBUF form on purpose: hipcc counts a FLAT-encoded `global_load_lds` as a possible LDS access, marks the wave
    //  "pending flat", and from then on every `s_waitcnt lgkmcnt` in front of an MFMA becomes lgkmcnt(0) -- the
    //  software-pipelined fragment reads stop overlapping.  `buffer_load ... lds` only touches vmcnt.)
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(p.w), 0,
                                                                           (unsigned)(9 * p.n_chunks * p.Cout_pad * 128), 0x00020000);
    const int wlane = lane * 16;
    const int wstride = p.Cout_pad * 128;                  // one tap-step of the packed image ([chunk][tap][Cout_pad][128 B])
    const int wpiece = (wave >> 2) * wstride + (wave & 3) * 4096;   // this wave's 4 pieces: step t0 + (wave >> 2), rows (wave & 3)*32..+31
    auto w_issue = [&](int t0, int c0, int sel) {          // steps t0, t0+1 of a tile (t = chunk * 9 + tap): consecutive in memory
        const int soff = t0 * wstride + c0 * 128 + wpiece;
        unsigned char* dst = wbuf + sel * S_WSTAGE + wave * 4096;
        // the instruction's immediate offset advances BOTH the memory address and the LDS address (LDS = M0 + imm + 16*lane)
#ifndef S_ABL_NOW
#define S_WDMA(K) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (__attribute__((address_space(3))) void*)dst, 16, wlane, soff, (K) * 1024, 0)
        S_WDMA(0); S_WDMA(1); S_WDMA(2); S_WDMA(3);
#undef S_WDMA
#else
        if (p.N == -12345) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (__attribute__((address_space(3))) void*)dst, 16, wlane, soff, 0, 0);
#endif
    };

    // ---- patch operations ---------------------------------------------------------------------------------------
    float sc[8], sh[8];
    auto p_dma = [&](__amdgpu_buffer_rsrc_t rs, const int* vo, int soff, int buf, int i0, int cnt) {
#pragma unroll
        for (int k = 0; k < G::NSLOT; ++k) {
            if (k < i0 || k >= i0 + cnt) continue;
            const int piece = (wave + 8 * k < G::NPIECE) ? wave + 8 * k : wave + 8 * (k - 1);
#ifdef S_ABL_NOPATCH
            if (p.N != -12345) continue;
#endif
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(patch + buf * G::PATCH + piece * 1024),
                                                     16, vo[k], soff, 0, 0);
        }
    };
    auto ss_load = [&](int n, int ci0) {                  // 8 (scale, shift) pairs of this thread's logical channel slot: 4 x 16 B
        const f32x4* sp = reinterpret_cast<const f32x4*>(p.ss + ((size_t)n * p.Cin + ci0 + (lane & 7) * 8) * 2);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 v = sp[q];
            sc[2 * q] = v[0]; sh[2 * q] = v[1]; sc[2 * q + 1] = v[2]; sh[2 * q + 1] = v[3];
        }
    };
    // GroupNorm(+SiLU) prologue, applied IN PLACE to a patch buffer the DMA has filled with raw activations: this thread takes
    // logical channel slot lane & 7 (one set of 8 scale/shift pairs) of the pixels of slots i0..i0+cnt-1.  Padding pixels were
    // written as zeros by the DMA and must stay zero (the padding is applied AFTER the activation).
    auto p_activate = [&](unsigned inb_mask, int buf, int i0, int cnt) {
#pragma unroll
        for (int k = 0; k < G::NSLOT; ++k) {
            if (k < i0 || k >= i0 + cnt) continue;
            int pr, pc;
            const bool live = slot_pix(k, pr, pc) && (k < 5 || wave == 0) && ((inb_mask >> k) & 1u);
            if (!live) continue;
            unsigned char* dst = patch + buf * G::PATCH + (pr * S_PWL + pc) * 128 + (((lane & 7) ^ ((pc >> 1) & 7)) << 4);
            u32x4 v = *reinterpret_cast<const u32x4*>(dst);
            bf16_t* tv = reinterpret_cast<bf16_t*>(&v);
            if (p.act == MAS_ACT_AFFINE_SILU) {
#pragma unroll
                for (int e = 0; e < 8; ++e) tv[e] = (bf16_t)silu_f((float)tv[e] * sc[e] + sh[e]);
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) tv[e] = (bf16_t)((float)tv[e] * sc[e] + sh[e]);
            }
            *reinterpret_cast<u32x4*>(dst) = v;
        }
    };

    // ---- per-lane fragment addressing -----------------------------------------------------------------------------
    int bq[G::NJ];                                 // byte offset (inside a patch buffer) of this lane's pixel at tap (0,0)
#pragma unroll
    for (int j = 0; j < G::NJ; ++j) {
        const int pix = (wave_p * G::NJ + j) * 32 + l31;
        bq[j] = ((pix >> 4) * S_PWL + (pix & 15)) * 128;
    }
    const int bcol = l31 & 15;
    int bxs[3];                                 // (g ^ column swizzle) << 4 for kw = 0, 1, 2
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) bxs[kw] = (g ^ (((bcol + kw) >> 1) & 7)) << 4;
    int aoff[2], axs[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = (wave_c * 2 + i) * 32 + l31;
        aoff[i] = row * 128;
        axs[i] = (g ^ ((row >> 1) & 7)) << 4;
    }

    // ---- deferred output of the previous tile ----------------------------------------------------------------------
    u32x4 outp[G::NST];
    int ooff[G::NJ];
#pragma unroll
    for (int j = 0; j < G::NJ; ++j) ooff[j] = S_OOB;
    bool pending_out = false;
    bool imm_stores = false;                    // the previous tile's epilogue issued its 8 stores directly (they may stay in flight)
#pragma unroll
    for (int k = 0; k < G::NST; ++k) outp[k] = u32x4{0u, 0u, 0u, 0u};
    auto store_one = [&](int k) {               // k = (j*2 + i)*2 + qp
        const int j = k >> 2, i = (k >> 1) & 1, qp = k & 1;
        const int off = ooff[j] + (i * 32 + qp * 16) * 2;
#ifdef S_ABL_NOEPI
        if (p.N != -12345) { asm volatile("" :: "v"(outp[k])); return; }
#endif
        __builtin_amdgcn_raw_buffer_store_b128(outp[k], rs_y, off, 0, 0);
    };

    // ---- prologue -----------------------------------------------------------------------------------------------------
    int tile = blockIdx.x;                      // grid <= total_tiles
    int step = (int)gridDim.x, tile_end = total_tiles;
    if (p.xcd_bands && (gridDim.x & 7) == 0) {  // work-group b runs on XCD b % 8: the cout tiles that share a patch (consecutive tile ids) on ONE XCD
        const int xcd = blockIdx.x & 7;
        step = (int)(gridDim.x >> 3);
        tile = (int)(((long long)total_tiles * xcd) >> 3) + (int)(blockIdx.x >> 3);
        tile_end = (int)(((long long)total_tiles * (xcd + 1)) >> 3);
    }
    Tile cur = decode(tile);
    int vo_cur[G::NSLOT], vo_nxt[G::NSLOT];
    unsigned inb_cur, inb_nxt;
    make_plan(cur, vo_cur, inb_cur);
#pragma unroll
    for (int i = 0; i < G::NSLOT; ++i) vo_nxt[i] = vo_cur[i];
    inb_nxt = inb_cur;
    __amdgpu_buffer_rsrc_t rs_cur = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(p.x) + (size_t)cur.n * img_bytes, 0,
                                                                      (unsigned)img_bytes, 0x00020000);
    __amdgpu_buffer_rsrc_t rs_nxt = rs_cur;
    int wsel = 0;
    w_issue(0, cur.c0, 0);
    p_dma(rs_cur, vo_cur, 0, 0, 0, G::NSLOT);
    if constexpr (ACT) {
        ss_load(cur.n, 0);
        S_WAIT_BARRIER(0);                       // the raw patch of chunk 0 has landed for every wave
        p_activate(inb_cur, 0, 0, 3);
        p_activate(inb_cur, 0, 3, 3);
    }
    const int n_pairs = p.n_chunks >> 1;
#ifdef S_PRIO_HALF   // MI355X_MICROARCH.md "Two waves per SIMD" item 4: static priority for the second-dispatched half
    if (wave >= 4) __builtin_amdgcn_s_setprio(1);
#endif
#ifdef S_DEBUG_DUMP      // debug build: dump weight stage 0 (32 KiB) + patch buffer 0 (41 KiB) of work-group 0's first tile into y
    S_WAIT_BARRIER(0);
    if (blockIdx.x == 0) {
        for (int i = tid; i < S_WSTAGE / 16; i += 512) reinterpret_cast<u32x4*>(p.y)[i] = *reinterpret_cast<const u32x4*>(wbuf + i * 16);
        for (int i = tid; i < G::PATCH / 16; i += 512) reinterpret_cast<u32x4*>(p.y + S_WSTAGE)[i] = *reinterpret_cast<const u32x4*>(patch + i * 16);
    }
    return;
#endif

#ifdef S_TIMELINE
    int tl_iter = 0;
#endif
    for (;;) {
        const int next_tile = tile + step;
        const bool has_next = next_tile < tile_end;
        const Tile nxt = has_next ? decode(next_tile) : cur;
        STS(0);

        f32x16 acc[2][G::NJ];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < G::NJ; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

        for (int pair = 0; pair < n_pairs; ++pair) {
            const bool last_pair = pair + 1 == n_pairs;
            const bool do_store = DEFER && pending_out && pair == 0;
            const int ciA = pair * 128;                       // first channel of chunk A (even) of this pair; B = +64
            if (last_pair) {                                  // the next tile's staging plan (used from stage 5 on)
                make_plan(nxt, vo_nxt, inb_nxt);
                rs_nxt = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(p.x) + (size_t)nxt.n * img_bytes, 0,
                                                           (unsigned)img_bytes, 0x00020000);
            }
            s_static_for(std::make_integer_sequence<int, 9>{}, [&](auto s_c) {
                constexpr int s = decltype(s_c)::value;
                // ---- barrier(s): stage s's weights, and every patch chunk it reads, are visible; stage s-1's buffers are free.
                //      vmcnt allowance = the VMEM operations issued AFTER the weight DMA in stage s-1 (they may stay in flight)
                if (pair == 0) STS(1 + 3 * s);
                {
                    constexpr int sp = (s + 8) % 9;           // previous stage (of this or the previous pair)
                    constexpr int npatch = ACT ? ((sp == 0 || sp == 5) ? 7 : ((sp == 1 || sp == 6) ? 3 : 0))
                                               : ((sp <= 2 || (sp >= 5 && sp <= 7)) ? G::PCNT : 0);
                    if (s > 0 && sp < G::NST && do_store) s_wait_barrier(npatch + 1);     // stage s-1 also issued one deferred store
                    else if (s == 0 && imm_stores) { s_wait_barrier(G::NST); imm_stores = false; }   // younger than stage 0's weight DMA
                    else s_wait_barrier(npatch);
                }
                if (pair == 0) STS(2 + 3 * s);
#ifdef S_DEBUG_DUMP_STAGE   // debug build: the weight stage + both patch buffers as stage S_DEBUG_DUMP_STAGE of work-group 0 sees them -> p.res
                if (s == S_DEBUG_DUMP_STAGE && blockIdx.x == 0 && tile == 0 && pair == 0) {
                    unsigned char* dd = const_cast<unsigned char*>(p.res);
                    for (int i = tid; i < S_WSTAGE / 16; i += 512) reinterpret_cast<u32x4*>(dd)[i] = *reinterpret_cast<const u32x4*>(wbuf + wsel * S_WSTAGE + i * 16);
                    for (int i = tid; i < 2 * G::PATCH / 16; i += 512) reinterpret_cast<u32x4*>(dd + S_WSTAGE)[i] = *reinterpret_cast<const u32x4*>(patch + i * 16);
                }
#endif
                // ---- register path: commit what was loaded two stages ago (guaranteed landed by the wait above)
                if constexpr (ACT) {
                    if (s == 2) p_activate(inb_cur, 1, 0, 3);
                    if (s == 3) p_activate(inb_cur, 1, 3, 3);
                    if (s == 7) p_activate(last_pair ? inb_nxt : inb_cur, 0, 0, 3);
                    if (s == 8) p_activate(last_pair ? inb_nxt : inb_cur, 0, 3, 3);
                    asm volatile("" ::: "memory");
                }
                // ---- weight DMA for stage s+1 (steps 2s+2, 2s+3 of this pair; stage 9 = stage 0 of the next pair / tile)
                {
                    const int tn = (s < 8) ? pair * 18 + 2 * s + 2 : (last_pair ? 0 : (pair + 1) * 18);
                    const int c0n = (s == 8 && last_pair) ? nxt.c0 : cur.c0;
                    w_issue(tn, c0n, wsel ^ 1);
                }
                asm volatile("" ::: "memory");                // VMEM order = source order: the counted waits depend on it
                auto late_ops = [&]() {
                // ---- patch operations for later chunks
                // (ACT: 3 pieces in each of two stages + the chunk's scale/shift, then the in-place activation two stages later;
                //  plain: 2 pieces in each of three stages)
                if (s <= 2) {                                 // chunk B of this pair -> buffer 1
                    if constexpr (ACT) {
                        if (s == 0) { p_dma(rs_cur, vo_cur, (ciA + 64) * 2, 1, 0, 3); ss_load(cur.n, ciA + 64); }
                        if (s == 1) p_dma(rs_cur, vo_cur, (ciA + 64) * 2, 1, 3, 3);
                    } else {
                        p_dma(rs_cur, vo_cur, (ciA + 64) * 2, 1, G::PCNT * s, G::PCNT);
                    }
                } else if (s >= 5 && s <= 7) {                // chunk A of the next pair, or chunk 0 of the next tile -> buffer 0
                    const int soff = last_pair ? 0 : (ciA + 128) * 2;
                    if constexpr (ACT) {
                        if (s == 5) {
                            if (last_pair) p_dma(rs_nxt, vo_nxt, soff, 0, 0, 3); else p_dma(rs_cur, vo_cur, soff, 0, 0, 3);
                            ss_load(last_pair ? nxt.n : cur.n, last_pair ? 0 : ciA + 128);
                        }
                        if (s == 6) { if (last_pair) p_dma(rs_nxt, vo_nxt, soff, 0, 3, 3); else p_dma(rs_cur, vo_cur, soff, 0, 3, 3); }
                    } else {
                        if (last_pair) p_dma(rs_nxt, vo_nxt, soff, 0, G::PCNT * (s - 5), G::PCNT); else p_dma(rs_cur, vo_cur, soff, 0, G::PCNT * (s - 5), G::PCNT);
                    }
                }
                // ---- one deferred store of the previous tile's output
                if constexpr (DEFER) {
                    if (s < G::NST && do_store) store_one(s);
                }
                asm volatile("" ::: "memory");
                };
#ifndef S_LATE_ISSUE
                late_ops();
#endif
                if (pair == 0) STS(3 + 3 * s);
                // ---- 2 tap-steps = 8 k-steps of 4 MFMAs; fragment reads software-pipelined one k-step ahead
                {
                    const unsigned char* wb = wbuf + wsel * S_WSTAGE;
                    bf16x8 bfr[2][G::NJ], afr[2][2];
                    auto ld_k = [&](int n, int b) {           // n = 0..7: step n >> 2, kk = n & 3
                        const int t = 2 * s + (n >> 2);       // step inside the pair
                        const int cb = t / 9, tap = t - cb * 9;
                        const int kh = tap / 3, kw = tap - kh * 3, kk = n & 3;
                        const unsigned char* pb = patch + cb * G::PATCH + (kh * S_PWL + kw) * 128;
#pragma unroll
                        for (int j = 0; j < G::NJ; ++j) bfr[b][j] = s_ld_frag(pb + bq[j], bxs[kw], kk);
#pragma unroll
                        for (int i = 0; i < 2; ++i) afr[b][i] = s_ld_frag(wb + (n >> 2) * S_WT + aoff[i], axs[i], kk);
                    };
                    ld_k(0, 0);
#pragma unroll
                    for (int n = 0; n < 8; ++n) {
                        if (n + 1 < 8) ld_k(n + 1, (n + 1) & 1);
#pragma unroll
                        for (int i = 0; i < 2; ++i)
#pragma unroll
                            for (int j = 0; j < G::NJ; ++j) mma16(acc[i][j], afr[n & 1][i], bfr[n & 1][j]);
                    }
                    __builtin_amdgcn_sched_group_barrier(0x100, 2 + G::NJ, 0);     // DS reads of k-step 0
#pragma unroll
                    for (int n = 0; n + 1 < 8; ++n) {
                        __builtin_amdgcn_sched_group_barrier(0x100, 2 + G::NJ, 0);
                        __builtin_amdgcn_sched_group_barrier(0x008, 2 * G::NJ, 0);
                    }
                    __builtin_amdgcn_sched_group_barrier(0x008, 2 * G::NJ, 0);
                }
#ifdef S_LATE_ISSUE   // HBM-facing operations of this stage issued at its END, where the early half of the waves would wait at the barrier anyway
                asm volatile("" ::: "memory");
                late_ops();
#endif
                wsel ^= 1;
            });
        }

        STS(40);
#ifdef S_ABL_NOEPIALL   // timing experiment only: no epilogue at all (the accumulators are consumed by an impossible store)
        if (p.N != -12345) {
            float t = 0.0f;
            for (int i = 0; i < 2; ++i) for (int j = 0; j < G::NJ; ++j) for (int r = 0; r < 16; ++r) t += acc[i][j][r];
            if (t == 123.456f) reinterpret_cast<float*>(p.y)[0] = t;
        } else
#endif
        // ---- epilogue: lanes l / l+32 exchange accumulator quads (fp32) so each lane owns 8 consecutive couts of its pixel;
        //      bias and residual arrive by UNCONDITIONAL buffer loads (a null pointer is a zero-length descriptor that returns
        //      zeros): no per-load branches, so hipcc batches the 24 loads instead of waiting for each one
        {
            f32x4 bv[2][2][2];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int qp = 0; qp < 2; ++qp) {
                    const int cb = (cur.c0 + (wave_c * 2 + i) * 32 + 16 * qp + 8 * g) * 4;
                    bv[i][qp][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_b, cb, 0, 0));
                    bv[i][qp][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_b, cb + 16, 0, 0));
                }
#pragma unroll
            for (int j = 0; j < G::NJ; ++j) {
                const int pix = (wave_p * G::NJ + j) * 32 + l31;
                const int ho = cur.h0 + (pix >> 4), wo = cur.w0 + (pix & 15);
                const bool pix_ok = (ho < p.Ho) && (wo < p.Wo);
                // byte offset of (n, ho, wo, c0 + wave_c*64 + 8*g) -- the stores add (i*32 + qp*16)*2
                const int obase = pix_ok ? (int)((((size_t)(cur.n * p.Ho + ho) * p.Wo + wo) * p.Cout + cur.c0 + wave_c * 64 + 8 * g) * 2) : S_OOB;
                u32x4 rv[2][2];
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int qp = 0; qp < 2; ++qp) rv[i][qp] = __builtin_amdgcn_raw_buffer_load_b128(rs_r, obase + (i * 32 + qp * 16) * 2, 0, 0);
#pragma unroll
                for (int i = 0; i < 2; ++i) {
#pragma unroll
                    for (int qp = 0; qp < 2; ++qp) {
                        float v[8];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            // (copy the vector elements to scalars first: __builtin_bit_cast applied directly to an ext-vector
                            //  element lvalue reads element 0 of the vector, whatever the index)
                            const float qa = acc[i][j][(2 * qp) * 4 + e], qb = acc[i][j][(2 * qp + 1) * 4 + e];
                            const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(qa), __float_as_uint(qb), false, false);
                            v[e] = __uint_as_float(r[0]); v[4 + e] = __uint_as_float(r[1]);
                        }
                        const bf16_t* rb = reinterpret_cast<const bf16_t*>(&rv[i][qp]);
                        u32x4 o;
                        bf16_t* ob = reinterpret_cast<bf16_t*>(&o);
#pragma unroll
                        for (int e = 0; e < 8; ++e) ob[e] = (bf16_t)(v[e] + bv[i][qp][e >> 2][e & 3] + (float)rb[e]);
                        if constexpr (DEFER) outp[(j * 2 + i) * 2 + qp] = o;
                        else __builtin_amdgcn_raw_buffer_store_b128(o, rs_y, obase + (i * 32 + qp * 16) * 2, 0, 0);
                    }
                }
                if constexpr (DEFER) ooff[j] = obase;
            }
            pending_out = DEFER;
            imm_stores = !DEFER;
        }
        STS(41);
#ifdef S_TIMELINE
        ++tl_iter;
#endif
        if (!has_next) break;
        tile = next_tile; cur = nxt; rs_cur = rs_nxt; inb_cur = inb_nxt;
#pragma unroll
        for (int i = 0; i < G::NSLOT; ++i) vo_cur[i] = vo_nxt[i];
    }
    if constexpr (DEFER) {
        if (pending_out) {
#pragma unroll
            for (int k = 0; k < G::NST; ++k) store_one(k);
        }
    }
}

template <bool ACT, bool DEFER, int TH>
int launch_stream(const StreamParams& p, hipStream_t s) {
    auto kern = conv3x3_stream_kernel<ACT, DEFER, TH>;
    constexpr int S_LDS = 2 * SGeo<TH>::PATCH + 2 * S_WSTAGE;
    static mas_devmask_t attr_mask{0};
    unsigned long long attr_bit;
    if (mas_attr_needed(attr_mask, &attr_bit)) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, S_LDS) != hipSuccess)
            MAS_FAIL(MAS_ELAUNCH, "conv3x3_stream: cannot set dynamic LDS size %d", S_LDS);
        mas_attr_done(attr_mask, attr_bit);
    }
    const long long tiles = (long long)p.N * p.tiles_h * p.tiles_w * p.n_ct;
    // 4x the resident work-groups (one per CU): see conv_fwd.hip launch_v -- a co-running RCCL kernel then costs a quarter round
    long long resident = 4LL * mas_num_cus();
    static const int wgs_per_cu = mas_env_int("MAS_CONV_WGS_PER_CU", 0);
    if (wgs_per_cu > 0) resident = (long long)wgs_per_cu * mas_num_cus();
    const unsigned blocks = (unsigned)(tiles < resident ? tiles : resident);
    static const int bands = mas_env_int("MAS_CONV_XCD_BANDS", 1);
    StreamParams pb = p;
    pb.xcd_bands = (bands && blocks % 8 == 0 && tiles / 8 >= blocks / 8) ? 1 : 0;
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(512), S_LDS, s, pb);
    MAS_CHECK_LAUNCH("conv3x3_stream");
    return MAS_OK;
}

}  // namespace

// Returns 1 if the shape qualifies and the launch was made, 0 if the caller should use the general kernel, < 0 on error.
int mas_conv3x3_stream_try(const MasConvDesc* d, const void* x, const float* scale_shift, const void* w_packed, const float* bias,
                           const void* residual, void* y, hipStream_t s) {
    static const int mode = mas_env_int("MAS_CONV_STREAM", 3);   // 0 off | 1 on, immediate stores | 3 on, deferred stores (default)
    if (!(mode & 1)) return 0;
    if (d->ks != 3 || d->stride != 1 || d->in_dtype != MAS_BF16 || d->out_dtype != MAS_BF16) return 0;
    if (d->Cin % 128 != 0 || d->Cout % 128 != 0) return 0;
    const long long img_bytes = (long long)d->H * d->W * d->Cin * 2;
    const long long out_bytes = (long long)d->N * d->Ho * d->Wo * d->Cout * 2;
    if (img_bytes >= 0x7fffffffLL || out_bytes >= 0x7fffffffLL) return 0;
    StreamParams p;
    p.dbg = nullptr;
#ifdef S_TIMELINE
    if (const char* e = getenv("MAS_DBG_PTR")) p.dbg = reinterpret_cast<unsigned long long*>(strtoull(e, nullptr, 0));
#endif
    p.x = (const unsigned char*)x; p.ss = scale_shift; p.w = (const unsigned char*)w_packed; p.bias = bias;
    p.res = (const unsigned char*)residual; p.y = (unsigned char*)y;
    p.N = d->N; p.H = d->H; p.W = d->W; p.Cin = d->Cin; p.Ho = d->Ho; p.Wo = d->Wo; p.Cout = d->Cout;
    p.Hl = d->upsample ? 2 * d->H : d->H; p.Wl = d->upsample ? 2 * d->W : d->W;
    p.pad_top = d->pad_top; p.pad_left = d->pad_left; p.upsample = d->upsample; p.act = d->act;
    p.n_chunks = d->Cin / 64; p.Cout_pad = mas_roundup(d->Cout, 128);
    p.tiles_h = mas_cdiv(d->Ho, 16); p.tiles_w = mas_cdiv(d->Wo, 16); p.n_ct = d->Cout / 128;
    long long tiles = (long long)p.N * p.tiles_h * p.tiles_w * p.n_ct;
    // fewer 16-row tiles than CUs (the 16x16 level at batch 32: 128): 8-row tiles, twice as many (prologue-free launches only)
    static const int th8 = mas_env_int("MAS_CONV_STREAM_TH8", 1);
    const bool half = th8 && d->act == MAS_ACT_NONE && tiles < mas_num_cus();
    if (half) { p.tiles_h = mas_cdiv(d->Ho, 8); tiles = (long long)p.N * p.tiles_h * p.tiles_w * p.n_ct; }
    // Small maps (fewer tiles than CUs): round 2 sent them to the 8x16-tile general kernel, which fills the chip; since the 16x16 level's
    // layers are (Cin, Cout) = (512, 512) -- 72 tap-steps per tile -- one stream tile on half the CUs is as fast per launch (52 vs 51 us)
    // and the step is 0.3 ms faster with it (profiles/r03_ab_stream_small.txt).  MAS_CONV_STREAM_MIN_TILES_PER_CU=2 restores the old rule.
    static const int min_per_cu = mas_env_int("MAS_CONV_STREAM_MIN_TILES_PER_CU", 0);
    if (tiles < (long long)min_per_cu * mas_num_cus() || tiles > 0x7fffffffLL) return 0;
    const bool defer = (mode & 2) != 0;
    int rc;
    if (d->act != MAS_ACT_NONE) rc = defer ? launch_stream<true, true, 16>(p, s) : launch_stream<true, false, 16>(p, s);
    else if (half) rc = defer ? launch_stream<false, true, 8>(p, s) : launch_stream<false, false, 8>(p, s);
    else rc = defer ? launch_stream<false, true, 16>(p, s) : launch_stream<false, false, 16>(p, s);
    return rc == MAS_OK ? 1 : rc;
}
