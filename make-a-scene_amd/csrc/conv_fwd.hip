// Implicit-GEMM convolution forward for gfx950 (also the data gradient of every conv).
//
// Replaces F.conv2d at reference models/modules.py:49,68,93,100,113,145-160,219,236,345,364
// and models/vqvae.py:15,18, with the GroupNorm-apply + SiLU of modules.py:121-128 fused into
// the input loader and bias / residual add (modules.py:136,191) into the epilogue.
//
// GEMM view:  Y^T[cout][pixel] = sum_{tap,ci} Wp[tap][cout][ci] * A[pixel (+) tap][ci]
//   M = cout  (MFMA A operand = packed weights), N = pixel (MFMA B operand = activations; a lane owns
//   one pixel, so an accumulator quad is 4 consecutive couts -> 8/16-byte NHWC stores), K = (tap, ci).
// Work-group: 256 threads = 4 waves; output tile = 8x16 pixels x BC couts; 2 work-groups per CU.
//
// LDS im2col: the (8s+ks-1) x (16s+ks-1) input halo patch of a 128-byte channel chunk is staged ONCE
// (coalesced 16-byte NHWC loads; prologue applied once per element; zero padding applied after the
// activation) and all ks*ks taps read their shifted B fragments out of it.  The loads of the NEXT chunk
// are issued into registers before the MFMA phase of the current one, so their HBM latency hides
// under 9 taps of MFMAs.  Pixel rows are exactly 128 B; bank conflicts are removed by XOR-swizzling the
// 16-byte slot index with (patch_column>>1)&7 instead of padding.
// Weights: mas_pack_conv_weight emits the swizzled LDS image of every [chunk][tap][cout] row, so a
// weight tile is a linear, perfectly coalesced 16-byte copy (global -> registers one tap ahead ->
// ds_write_b128) into a double buffer; one barrier per tap.  All global loads are ordinary loads, so
// hipcc's counted s_waitcnt vmcnt(N) keeps the patch prefetch (issued one slot per tap, after that
// tap's weight loads) in flight across two taps and barriers.
#include "mas_common.h"
#include <stdlib.h>

namespace {

struct ConvParams {
    unsigned long long* dbg;
    const void* x; const float* ss; const void* w; const float* bias; const void* res; void* y;
    int N, H, W, Cin, Ho, Wo, Cout;
    int Hl, Wl;            // logical input size (2H,2W when upsample)
    int pad_top, pad_left, act, upsample;
    int n_chunks, Cout_pad; // packed weight dims
    int tiles_h, tiles_w, n_ct;
};

constexpr int TW = 16;
// BIG = 0: 256 threads / 8x16-pixel tile, 2 work-groups per CU; 1: 512 threads / 16x16 (weights fetched once per 256
// pixels), 1 per CU; 2: 512 threads / 32x16, each wave a 64(cout) x 128(pixel) block (0.75 LDS reads per MFMA)

template <typename T, int KS, int STRIDE, int BIG>
struct Geo {
    static constexpr int TH = BIG == 2 ? 32 : (BIG ? 16 : 8), NT = BIG ? 512 : 256;
    static constexpr int EPU = 16 / (int)sizeof(T);          // elements per 16-byte slot
    static constexpr int CK = 128 / (int)sizeof(T);          // channels per chunk (one 128-byte pixel row)
    static constexpr int PH = (TH - 1) * STRIDE + KS, PW = (TW - 1) * STRIDE + KS;
    static constexpr int PWL = (PW + 1) & ~1;                // even LDS pitch: slot parity == column parity
    static constexpr int PATCH_BYTES = PH * PWL * 128;
    static constexpr int P_UNITS = PH * PWL * 8;             // 16-byte slots in the patch
    static constexpr int NPU = (P_UNITS + NT - 1) / NT;      // slots per thread
    static constexpr int PB = (BIG == 2 && NPU <= 10) ? NPU : (NPU < 6 ? NPU : 6);   // slots per staging batch (registers)
};

// 8 consecutive K elements of one 128-byte row whose 16-byte slots are XOR-swizzled with h
template <typename T>
__device__ __forceinline__ typename Vec8<T>::type ld_frag(const unsigned char* row, int h, int kk, int g);
template <>
__device__ __forceinline__ bf16x8 ld_frag<bf16_t>(const unsigned char* row, int h, int kk, int g) {
    return *reinterpret_cast<const bf16x8*>(row + (((kk * 2 + g) ^ h) << 4));
}
template <>
__device__ __forceinline__ f32x8 ld_frag<float>(const unsigned char* row, int h, int kk, int g) {
    const int s = (kk * 4 + g * 2) ^ h;
    const f32x4 lo = *reinterpret_cast<const f32x4*>(row + (s << 4));
    const f32x4 hi = *reinterpret_cast<const f32x4*>(row + ((s ^ 1) << 4));
    f32x8 r = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return r;
}

template <typename T, typename TO, int KS, int STRIDE, int BC, int WC, bool VEC, int BIG>
__global__ __launch_bounds__(BIG ? 512 : 256, 2) void conv_fwd_kernel(ConvParams p) {
    using G = Geo<T, KS, STRIDE, BIG>;
    constexpr int TH = G::TH, NT = G::NT, NWAVE = NT / 64;
    using V8 = typename Vec8<T>::type;
    constexpr int EPU = G::EPU, CK = G::CK, PW = G::PW, PWL = G::PWL, NPU = G::NPU, PB = G::PB;
    constexpr int WP = NWAVE / WC;             // waves along pixels
    constexpr int MI = BC / WC / 32;           // 32-cout tiles per wave
    constexpr int NI = (TH * TW) / WP / 32;    // 32-pixel tiles per wave
    constexpr int WT_BYTES = BC * 128;         // one weight tile (BC rows x 128 B)
    constexpr int NTAP = KS * KS;
    // weight stage = TPS consecutive taps staged (and barrier-synchronised) together: the measured cost of a
    // work-group barrier is ~650 cycles of wave skew, so the big tile amortises it over 48 MFMAs instead of 16
    constexpr int TPS = (BIG == 1 && KS == 3) ? 3 : 1;
    constexpr int NSTAGE = NTAP / TPS;
    constexpr bool PREFETCH = VEC && (NPU <= PB);   // whole chunk fits one register batch -> issue early / write late
    constexpr int PPT = (NPU + NSTAGE - 1) / NSTAGE;   // prefetch slots issued per stage
    constexpr bool vec_in = VEC;               // Cin % EPU == 0: 16-byte aligned channel slots

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* patch = smem;
    unsigned char* wbuf = smem + G::PATCH_BYTES;   // 2 buffers

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wave_c = wave % WC, wave_p = wave / WC;
    const int g = lane >> 5, l31 = lane & 31;
    const int sl = tid & 7;                    // this thread's 16-byte channel slot inside a 128-byte pixel row

    const T* __restrict__ Xall = reinterpret_cast<const T*>(p.x);
    const unsigned char* __restrict__ Wimg = reinterpret_cast<const unsigned char*>(p.w);
    const size_t img_elems = (size_t)p.H * p.W * p.Cin;

    // ---- tile bookkeeping: the work-group is PERSISTENT and walks tiles t = blockIdx.x + k*gridDim.x -----
    // (so the patch loads of tile k+1 and the output stores of tile k-1 overlap the MFMAs of tile k instead of
    //  every work-group on the chip loading, computing and storing in lock-step)
    const int total_tiles = p.N * p.tiles_h * p.tiles_w * p.n_ct;
    struct Tile { int n, h0, w0, c0; };
    auto decode = [&](int t) {
        Tile tc;
        const int ct = t % p.n_ct; t /= p.n_ct;
        const int tw_i = t % p.tiles_w; t /= p.tiles_w;
        const int th_i = t % p.tiles_h; tc.n = t / p.tiles_h;
        tc.c0 = ct * BC; tc.h0 = th_i * TH; tc.w0 = tw_i * TW;
        return tc;
    };
    // staging plan: thread owns slot `sl` of patch pixels q = (tid>>3) + 32*i
    int dstoff[NPU];                           // byte offset in the LDS patch (swizzled), -1 = skip (tile independent)
#pragma unroll
    for (int i = 0; i < NPU; ++i) {
        const int q = (tid >> 3) + i * (NT / 8);
        const int pr = q / PWL, pc = q - pr * PWL;
        const bool live = (q < G::PH * PWL) && (pc < PW);
        dstoff[i] = live ? q * 128 + ((sl ^ ((pc >> 1) & 7)) << 4) : -1;
    }
    auto make_plan = [&](const Tile& tc, int (&so)[NPU]) {   // element offset of each pixel in its image, -1 = zero padding
#pragma unroll
        for (int i = 0; i < NPU; ++i) {
            const int q = (tid >> 3) + i * (NT / 8);
            const int pr = q / PWL, pc = q - pr * PWL;
            int ih = tc.h0 * STRIDE + pr - p.pad_top, iw = tc.w0 * STRIDE + pc - p.pad_left;
            const bool inb = (dstoff[i] >= 0) && (ih >= 0) && (ih < p.Hl) && (iw >= 0) && (iw < p.Wl);
            if (p.upsample) { ih >>= 1; iw >>= 1; }
            so[i] = inb ? (ih * p.W + iw) * p.Cin : -1;
        }
    };

    u32x4 preg[PB];
    auto p_issue_one = [&](const T* X, const int (&so)[NPU], int ci0, int b0, int k) {   // global -> register k
        const int cb = ci0 + sl * EPU;
        const int i = b0 + k;
        u32x4 v = {0u, 0u, 0u, 0u};
        if constexpr (vec_in) {
#ifdef MAS_ABL_NOPLOAD
            if (p.N != -12345) { preg[k] = v; return; }
#endif
            // unconditional load (clamped address; commit discards it for padding) keeps the per-wave
            // VMEM instruction count static, which hipcc's counted waits rely on
            const int o = (i < NPU && so[i < NPU ? i : 0] >= 0 && cb < p.Cin) ? so[i < NPU ? i : 0] + cb : 0;
            v = *reinterpret_cast<const u32x4*>(X + o);
        } else if (i < NPU && so[i < NPU ? i : 0] >= 0 && cb < p.Cin) {
            const T* src = X + so[i] + cb;
            T* tv = reinterpret_cast<T*>(&v);
#pragma unroll
            for (int e = 0; e < EPU; ++e) if (cb + e < p.Cin) tv[e] = src[e];
        }
        preg[k] = v;
    };
    auto p_issue = [&](const T* X, const int (&so)[NPU], int ci0, int b0) {
#pragma unroll
        for (int k = 0; k < PB; ++k) p_issue_one(X, so, ci0, b0, k);
    };
    auto p_commit = [&](const int (&so)[NPU], int n, int ci0, int b0) {    // registers -> (prologue) -> LDS
        const int cb = ci0 + sl * EPU;
        float sc[EPU], sh[EPU];
        if (p.act != MAS_ACT_NONE) {
            if constexpr (vec_in) {            // 16-byte loads of the [c][2] pairs (channels past Cin are masked below)
                const int cc = (cb + EPU <= p.Cin) ? cb : (p.Cin - EPU);
                const f32x4* sp = reinterpret_cast<const f32x4*>(p.ss + ((size_t)n * p.Cin + cc) * 2);
#pragma unroll
                for (int q = 0; q < EPU / 2; ++q) {
                    const f32x4 v = sp[q];
                    sc[2 * q] = v[0]; sh[2 * q] = v[1]; sc[2 * q + 1] = v[2]; sh[2 * q + 1] = v[3];
                }
            } else {
#pragma unroll
                for (int e = 0; e < EPU; ++e) {
                    const int c = cb + e;
                    sc[e] = (c < p.Cin) ? p.ss[((size_t)n * p.Cin + c) * 2 + 0] : 0.0f;
                    sh[e] = (c < p.Cin) ? p.ss[((size_t)n * p.Cin + c) * 2 + 1] : 0.0f;
                }
            }
        }
#pragma unroll
        for (int k = 0; k < PB; ++k) {
            const int i = b0 + k;
            if (i >= NPU || dstoff[i < NPU ? i : 0] < 0) continue;
            u32x4 v = preg[k];
            if (so[i] < 0 || cb >= p.Cin) v = u32x4{0u, 0u, 0u, 0u};   // zero padding / channels past Cin
            if (p.act != MAS_ACT_NONE && so[i] >= 0) {          // padding stays exactly zero
                T* tv = reinterpret_cast<T*>(&v);
                // vec_in: Cin is a multiple of the slot, so a slot is entirely inside or outside the tensor (handled above);
                // the SiLU / plain-affine choice is wave-uniform: two straight-line bodies instead of per-element selects
                if (p.act == MAS_ACT_AFFINE_SILU) {
#pragma unroll
                    for (int e = 0; e < EPU; ++e) {
                        const float a = silu_f((float)tv[e] * sc[e] + sh[e]);
                        tv[e] = (T)((vec_in || cb + e < p.Cin) ? a : 0.0f);
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < EPU; ++e) {
                        const float a = (float)tv[e] * sc[e] + sh[e];
                        tv[e] = (T)((vec_in || cb + e < p.Cin) ? a : 0.0f);
                    }
                }
            }
            *reinterpret_cast<u32x4*>(patch + dstoff[i]) = v;
        }
    };
    // weight tile (tap, chunk, cout tile): a linear 16-byte-per-thread copy of its pre-swizzled image
#ifndef MAS_CONV_W_REGSTAGE
    // weight stage -> LDS by LDS-DMA (global_load_lds_dwordx4: no VGPRs, no VALU, no ds_write).  The packed image is
    // byte-for-byte the LDS tile, so each wave copies W_DMA contiguous 1-KiB pieces (lane l moves bytes [16l,16l+16)).
    // Double buffer: the DMA of stage s+1 is issued right after the barrier of stage s and is drained by the
    // vmcnt(0) hipcc puts in front of the next __syncthreads() -- it has the whole MFMA phase of stage s to land.
    constexpr int W_DMA = TPS * BC * 128 / 1024 / NWAVE;
    // MUBUF form (`buffer_load ... lds`), not `global_load_lds`: hipcc counts the FLAT-encoded instruction as a possible LDS
    // access ("pending flat") and then turns every lgkmcnt wait in front of an MFMA into lgkmcnt(0), which serialises the
    // software-pipelined fragment reads.
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(Wimg), 0,
                                                                           (unsigned)(NTAP * p.n_chunks * p.Cout_pad * 128), 0x00020000);
    auto w_issue = [&](int stage, int ch, int c0, int buf) {
#pragma unroll
        for (int k = 0; k < W_DMA; ++k) {
            const int piece = __builtin_amdgcn_readfirstlane(wave) * W_DMA + k;   // 1-KiB piece of the TPS-tile stage
            const int tt = piece / (BC / 8), row8 = piece % (BC / 8);
            const int soff = ((ch * NTAP + stage * TPS + tt) * p.Cout_pad + c0) * 128 + row8 * 1024;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (__attribute__((address_space(3))) void*)(wbuf + buf * (TPS * WT_BYTES) + piece * 1024),
                                                     16, lane * 16, soff, 0, 0);
        }
    };
    auto w_commit = [&](int) {};
    constexpr int NWLOAD = W_DMA;
#else
    constexpr int NWLOAD = TPS * W_PER_T;
    u32x4 wreg[TPS * W_PER_T];
    auto w_issue = [&](int stage, int ch, int c0, int) {
#pragma unroll
        for (int tt = 0; tt < TPS; ++tt) {
            const unsigned char* src = Wimg + ((size_t)(ch * NTAP + stage * TPS + tt) * p.Cout_pad + c0) * 128 + tid * 16;
#pragma unroll
            for (int k = 0; k < W_PER_T; ++k) wreg[tt * W_PER_T + k] = *reinterpret_cast<const u32x4*>(src + k * NT * 16);
        }
    };
    auto w_commit = [&](int buf) {
        unsigned char* dst = wbuf + buf * (TPS * WT_BYTES) + tid * 16;
#pragma unroll
        for (int k = 0; k < TPS * W_PER_T; ++k) *reinterpret_cast<u32x4*>(dst + k * NT * 16) = wreg[k];
    };
#endif

    // ---- per-lane fragment addressing ------------------------------------------------------------
    int bq[NI];                                // patch pixel index of this lane's pixel (tap (0,0))
#pragma unroll
    for (int j = 0; j < NI; ++j) {
        const int pix = (wave_p * NI + j) * 32 + l31;
        bq[j] = (pix >> 4) * STRIDE * PWL + (pix & 15) * STRIDE;
    }
    const int bcol = (l31 & 15) * STRIDE;      // patch column of the pixel at tap (.,0)
    int arow[MI];
#pragma unroll
    for (int i = 0; i < MI; ++i) arow[i] = (wave_c * MI + i) * 32 + l31;

#ifdef MAS_TIMELINE
    int tl_iter = 0;
#define TS(id) do { if (lane == 0 && blockIdx.x == 100 && tl_iter < 2 && p.dbg) p.dbg[(tl_iter * NWAVE + wave) * 64 + (id)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define TS(id) do {} while (0)
#endif
    int tile = blockIdx.x;                     // grid <= total_tiles
    Tile cur = decode(tile);
    int so_cur[NPU], so_nxt[NPU];
    make_plan(cur, so_cur);
#pragma unroll
    for (int i = 0; i < NPU; ++i) so_nxt[i] = so_cur[i];
    const T* Xcur = Xall + (size_t)cur.n * img_elems;
    w_issue(0, 0, cur.c0, 0);
    if (PREFETCH) p_issue(Xcur, so_cur, 0, 0);
    int wsel = 0;
    bool first = true;
    for (;;) {
        const int next_tile = tile + (int)gridDim.x;
        const bool has_next = next_tile < total_tiles;
        const Tile nxt = has_next ? decode(next_tile) : cur;
        const T* Xnxt = Xall + (size_t)nxt.n * img_elems;

        f32x16 acc[MI][NI];
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

        TS(0);
        for (int ch = 0; ch < p.n_chunks; ++ch) {
            const int ci0 = ch * CK;
            if (!first) __syncthreads();       // every wave is done reading the previous patch
            first = false;
            if (ch == 0) TS(1);
            if (PREFETCH) p_commit(so_cur, cur.n, ci0, 0);
            else for (int b0 = 0; b0 < NPU; b0 += PB) { p_issue(Xcur, so_cur, ci0, b0); p_commit(so_cur, cur.n, ci0, b0); }
            // what the taps of this chunk prefetch: the next chunk of this tile, or chunk 0 of the next tile
            const bool more = ch + 1 < p.n_chunks;
            if (!more && has_next) make_plan(nxt, so_nxt);
            const T* Xpf = more ? Xcur : Xnxt;
            const int pci = more ? ci0 + CK : 0;
#pragma unroll
            for (int stage = 0; stage < NSTAGE; ++stage) {
                TS(2 + ch * 28 + stage * 3);
                w_commit(wsel);
                TS(3 + ch * 28 + stage * 3);
#if !defined(MAS_CONV_W_REGSTAGE) && !defined(MAS_CONV_PLAIN_BARRIER)
                // With an LDS-DMA in flight hipcc drains vmcnt(0) in front of every __syncthreads(), which would also wait
                // for the patch slots prefetched one stage ago (HBM latency >= a stage).  In-order VMEM retirement lets a
                // COUNTED wait finish the weight DMA (older) and leave the NPF patch loads issued after it in flight:
                //   stage 0: everything (p_commit just consumed the registers);  stage s: N = slots issued in stage s-1.
                if (stage == 0 || !PREFETCH) {
                    // EXPLICIT vmcnt(0): hipcc's own __syncthreads() here is `s_waitcnt lgkmcnt(0); s_barrier` -- it counts on the
                    // vmcnt(0) that p_commit's last patch slot waits with, and that wait sits inside the slot's exec-masked region:
                    // a wave none of whose lanes owns the last slot (wave 3 of the 8x16 tile: slots 184..191 of 180) skips it and reaches
                    // the barrier with its newest weight DMA still in flight -- about one launch in 700 then read a stale weight stage
                    // (found in round 6 as a run-to-run difference of the encoder's last convolution; profiles/r06_determinism.txt)
                    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                    asm volatile("" ::: "memory");
                } else {
                    constexpr int lo = 0;
                    const int npf_prev = ((stage - 1) * PPT < NPU) ? (((stage) * PPT <= NPU) ? PPT : NPU - (stage - 1) * PPT) : lo;
                    if (npf_prev >= 2) asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");
                    else if (npf_prev == 1) asm volatile("s_waitcnt vmcnt(1) lgkmcnt(0)" ::: "memory");
                    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                    asm volatile("" ::: "memory");
                }
#elif defined(MAS_CONV_W_REGSTAGE)
                __syncthreads();               // weight stage (and, at stage 0, the patch) visible
#else
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // (see above: never leave the DMA to hipcc's barrier)
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
#endif
                TS(4 + ch * 28 + stage * 3);
                {
                    int nst = stage + 1, nch = ch, nc0 = cur.c0;
                    if (nst == NSTAGE) { nst = 0; nch = ch + 1; }
                    if (nch >= p.n_chunks) { nch = 0; nc0 = nxt.c0; }   // next tile (or, at the very end, a harmless re-read)
#ifndef MAS_ABL_NOWLOAD
                    w_issue(nst, nch, nc0, wsel ^ 1);
#endif
                }
                if (PREFETCH) {                // PPT slots per stage, issued after (= newer than) this stage's weight loads,
                                               // unconditionally: a branch-free VMEM stream lets hipcc keep them in flight
#pragma unroll
                    for (int k = stage * PPT; k < (stage + 1) * PPT && k < NPU; ++k) {
                        if (more) p_issue_one(Xpf, so_cur, pci, 0, k);
                        else p_issue_one(Xpf, so_nxt, pci, 0, k);
                    }
                }
                if constexpr (sizeof(T) == 2) __builtin_amdgcn_sched_group_barrier(0x020, NWLOAD + (PREFETCH ? PPT : 0), 0);   // VMEM reads first
#pragma unroll
              for (int tt = 0; tt < TPS; ++tt) {
                const int tap = stage * TPS + tt;
                const int kh = tap / KS, kw = tap - kh * KS;
                const int hb = ((bcol + kw) >> 1) & 7;
                const unsigned char* wb = wbuf + wsel * (TPS * WT_BYTES) + tt * WT_BYTES;
                // software-pipelined fragment loads: the ds_reads of k-step kk+1 are issued BEFORE the MFMAs of kk
                constexpr int NKK = CK / 16;   // channels past Cin are zero in both operands
                V8 bfr[2][NI], afr[2][MI];
                auto ld_k = [&](int kk, int b) {
#pragma unroll
                    for (int j = 0; j < NI; ++j) bfr[b][j] = ld_frag<T>(patch + (bq[j] + kh * PWL + kw) * 128, hb, kk, g);
#pragma unroll
                    for (int i = 0; i < MI; ++i) afr[b][i] = ld_frag<T>(wb + arow[i] * 128, (arow[i] >> 1) & 7, kk, g);
                };
                ld_k(0, 0);
#pragma unroll
                for (int kk = 0; kk < NKK; ++kk) {
                    if (kk + 1 < NKK) ld_k(kk + 1, (kk + 1) & 1);
#pragma unroll
                    for (int i = 0; i < MI; ++i)
#pragma unroll
                        for (int j = 0; j < NI; ++j) mma16(acc[i][j], afr[kk & 1][i], bfr[kk & 1][j]);
                }
                if constexpr (sizeof(T) == 2) {    // pin the interleave (masks: 0x100 DS read, 0x008 MFMA)
                    __builtin_amdgcn_sched_group_barrier(0x100, NI + MI, 0);
#pragma unroll
                    for (int kk = 0; kk + 1 < NKK; ++kk) {
                        __builtin_amdgcn_sched_group_barrier(0x100, NI + MI, 0);
                        __builtin_amdgcn_sched_group_barrier(0x008, MI * NI, 0);
                    }
                    __builtin_amdgcn_sched_group_barrier(0x008, MI * NI, 0);
                }
              }
                wsel ^= 1;
            }
        }

        TS(60);
#ifdef MAS_ABL_NOEPI
        if (p.N != -12345) {
            float t = 0.0f;
            for (int i = 0; i < MI; ++i) for (int j = 0; j < NI; ++j) for (int r = 0; r < 16; ++r) t += acc[i][j][r];
            if (t == 123.456f) reinterpret_cast<float*>(p.y)[0] = t;
        } else
#endif
        {
            const int n = cur.n, h0 = cur.h0, w0 = cur.w0, c0 = cur.c0;
            [&]() {
    // ---- epilogue: bias + residual, NHWC store ------------------------------------------------------
    TO* __restrict__ Y = reinterpret_cast<TO*>(p.y);
    const T* __restrict__ R = reinterpret_cast<const T*>(p.res);
    if constexpr (sizeof(TO) == 2 && sizeof(T) == 2) {
        if ((p.Cout & 7) == 0) {
            // bf16 fast path.  A lane owns 4 consecutive couts per accumulator quad and its partner lane (+32) the
            // next 4 of the same pixel: one v_permlane32_swap per dword turns two 8-byte pieces into one 16-byte
            // store per lane (half the store instructions; the tail is store-issue bound).
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                const int pix = (wave_p * NI + j) * 32 + l31;
                const int ho = h0 + (pix >> 4), wo = w0 + (pix & 15);
                const bool pix_ok = (ho < p.Ho) && (wo < p.Wo);
                const size_t obase = ((size_t)(n * p.Ho + ho) * p.Wo + wo) * p.Cout;
#pragma unroll
                for (int i = 0; i < MI; ++i) {
#pragma unroll
                    for (int qp = 0; qp < 2; ++qp) {
                        unsigned pk[2][2];
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            const int q = qp * 2 + h;
                            const int co = c0 + (wave_c * MI + i) * 32 + 8 * q + 4 * g;
                            float v[4];
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = acc[i][j][q * 4 + e];
                            if (pix_ok && co < p.Cout) {
                                if (p.bias) {
                                    const f32x4 b = *reinterpret_cast<const f32x4*>(p.bias + co);
#pragma unroll
                                    for (int e = 0; e < 4; ++e) v[e] += b[e];
                                }
                                if (R) {
                                    const bf16x4 rv = *reinterpret_cast<const bf16x4*>(R + obase + co);
#pragma unroll
                                    for (int e = 0; e < 4; ++e) v[e] += (float)rv[e];
                                }
                            }
                            bf16x4 o = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
                            const u32x2 od = *reinterpret_cast<const u32x2*>(&o);
                            pk[h][0] = od[0]; pk[h][1] = od[1];
                        }
                        // after the swaps: lanes 0-31 hold couts 8q..8q+7 of quad q=2qp, lanes 32-63 those of quad 2qp+1
#pragma unroll
                        for (int d = 0; d < 2; ++d) {
                            const auto r = __builtin_amdgcn_permlane32_swap(pk[0][d], pk[1][d], false, false);
                            pk[0][d] = r[0]; pk[1][d] = r[1];
                        }
                        const int co8 = c0 + (wave_c * MI + i) * 32 + 8 * (qp * 2 + g);
                        if (pix_ok && co8 < p.Cout) {
                            const u32x4 o = {pk[0][0], pk[0][1], pk[1][0], pk[1][1]};
                            *reinterpret_cast<u32x4*>(Y + obase + co8) = o;
                        }
                    }
                }
            }
            return;
        }
    }
    const bool vec_out = (p.Cout & 3) == 0;
#pragma unroll
    for (int j = 0; j < NI; ++j) {
        const int pix = (wave_p * NI + j) * 32 + l31;
        const int ho = h0 + (pix >> 4), wo = w0 + (pix & 15);
        if (ho >= p.Ho || wo >= p.Wo) continue;
        const size_t obase = ((size_t)(n * p.Ho + ho) * p.Wo + wo) * p.Cout;
#pragma unroll
        for (int i = 0; i < MI; ++i) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int co = c0 + (wave_c * MI + i) * 32 + 8 * q + 4 * g;
                if (co >= p.Cout) continue;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[i][j][q * 4 + e];
                if (vec_out) {
                    if (p.bias) {
                        const f32x4 b = *reinterpret_cast<const f32x4*>(p.bias + co);
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] += b[e];
                    }
                    if (R) {
                        if constexpr (sizeof(T) == 2) {
                            const bf16x4 rv = *reinterpret_cast<const bf16x4*>(R + obase + co);
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] += (float)rv[e];
                        } else {
                            const f32x4 rv = *reinterpret_cast<const f32x4*>(R + obase + co);
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] += rv[e];
                        }
                    }
                    if constexpr (sizeof(TO) == 4) {
                        f32x4 o = {v[0], v[1], v[2], v[3]};
                        *reinterpret_cast<f32x4*>(Y + obase + co) = o;
                    } else {
                        bf16x4 o = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
                        *reinterpret_cast<bf16x4*>(Y + obase + co) = o;
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if (co + e < p.Cout) {
                            float o = v[e] + (p.bias ? p.bias[co + e] : 0.0f);
                            if (R) o += (float)R[obase + co + e];
                            Y[obase + co + e] = (TO)o;
                        }
                    }
                }
            }
        }
    }
            }();
        }
        TS(61);
#ifdef MAS_TIMELINE
        ++tl_iter;
#endif
        if (!has_next) break;
        tile = next_tile; cur = nxt; Xcur = Xnxt;
#pragma unroll
        for (int i = 0; i < NPU; ++i) so_cur[i] = so_nxt[i];
    }
}

template <typename T, typename TO, int KS, int STRIDE, int BC, int WC, bool VEC, int BIG>
int launch_v(const ConvParams& p0, hipStream_t s) {
    using G = Geo<T, KS, STRIDE, BIG>;
    constexpr int NT = G::NT;
    ConvParams p = p0;
    p.tiles_h = mas_cdiv(p.Ho, G::TH); p.tiles_w = mas_cdiv(p.Wo, TW);
    constexpr int TPS = (BIG == 1 && KS == 3) ? 3 : 1;
    size_t lds = (size_t)G::PATCH_BYTES + 2 * TPS * BC * 128;
    auto kern = conv_fwd_kernel<T, TO, KS, STRIDE, BC, WC, VEC, BIG>;
    static mas_devmask_t attr_mask{0};
    unsigned long long attr_bit;
    if (mas_attr_needed(attr_mask, &attr_bit)) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            MAS_FAIL(MAS_ELAUNCH, "conv_fwd: cannot set dynamic LDS size %zu", lds);
        mas_attr_done(attr_mask, attr_bit);
    }
    ConvParams q = p;
    q.n_ct = mas_cdiv(p.Cout, BC);
    const long long tiles = (long long)p.N * p.tiles_h * p.tiles_w * q.n_ct;
    if (tiles <= 0 || tiles > 0x7fffffffLL) MAS_FAIL(MAS_EINVAL, "conv_fwd: bad tile count %lld", tiles);
    // Persistent work-groups walk a static stride of tiles.  The grid is 4x what fits on the chip at once (1 or 2 work-groups
    // per CU by LDS / VGPRs) so that the hardware dispatcher still balances at work-group granularity: when another kernel
    // (an RCCL collective overlapping backward) holds some CUs, a grid of exactly one work-group per CU would run the
    // displaced work-groups as a second full round (2x), this runs them as a fifth quarter-round (1.25x).  Measured cost
    // of the 4x on an idle GPU: < 0.5 % (kbench / bench.py).
    long long resident = 4LL * (BIG ? 1LL : 2LL) * mas_num_cus();
    static const int wgs_per_cu = mas_env_int("MAS_CONV_WGS_PER_CU", 0);      // tests: one work-group per CU -> several tiles per work-group
    if (wgs_per_cu > 0) resident = (long long)wgs_per_cu * mas_num_cus();
    const unsigned blocks = (unsigned)(tiles < resident ? tiles : resident);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(NT), lds, s, q);
    MAS_CHECK_LAUNCH("conv_fwd");
    return MAS_OK;
}

template <typename T, typename TO, int KS, int STRIDE, int BC, int WC>
int launch(const ConvParams& p, hipStream_t s) {
    if (p.Cin % (16 / (int)sizeof(T)) == 0) {
        if constexpr (KS == 3 && STRIDE == 1 && BC == 128 && sizeof(T) == 2) {
            // 16x16 tiles when they still fill the chip at one (8-wave) work-group per CU
            const long long big_tiles = (long long)p.N * mas_cdiv(p.Ho, 16) * mas_cdiv(p.Wo, TW) * mas_cdiv(p.Cout, BC);
            const long long huge_tiles = (long long)p.N * mas_cdiv(p.Ho, 32) * mas_cdiv(p.Wo, TW) * mas_cdiv(p.Cout, BC);
            (void)huge_tiles;   // (a 32x16 level spills: acc 128 + prefetch 40 + fragments 48 VGPRs; never dispatched)
            if (big_tiles >= 2LL * mas_num_cus()) return launch_v<T, TO, KS, STRIDE, BC, WC, true, 1>(p, s);
        }
        return launch_v<T, TO, KS, STRIDE, BC, WC, true, 0>(p, s);
    }
    return launch_v<T, TO, KS, STRIDE, BC, WC, false, 0>(p, s);
}

template <typename T, typename TO, int KS, int STRIDE>
int launch_bc(const ConvParams& p, hipStream_t s) {
    if (p.Cout <= 32) return launch<T, TO, KS, STRIDE, 32, 1>(p, s);
    if (p.Cout <= 64) return launch<T, TO, KS, STRIDE, 64, 1>(p, s);
    // Small maps (the 16x16 level of the encoder / decoder: 512 -> 512 at batch 32 is 256 tiles of 8x16 pixels x 128 couts) leave half
    // of the chip's 2 x 256 work-group slots empty -- one wave per SIMD, nothing to overlap with; 64-cout tiles double the work-groups
    // (the packed weight image is row-addressed: any 64-row window of it is a valid tile): -0.5 ms per VQ-IMG step (DESIGN history R3).
    if (STRIDE == 1) {                  // (stride 2 measured slower with 64-cout tiles: 0.114 vs 0.083 ms at 512 -> 512 @32^2)
        const long long tiles128 = (long long)p.N * mas_cdiv(p.Ho, 8) * mas_cdiv(p.Wo, TW) * mas_cdiv(p.Cout, 128);
        if (tiles128 < 2LL * mas_num_cus()) return launch<T, TO, KS, STRIDE, 64, 1>(p, s);
    }
    return launch<T, TO, KS, STRIDE, 128, 2>(p, s);
}

template <typename T, typename TO>
int launch_ks(const ConvParams& p, int ks, int stride, hipStream_t s) {
    if (ks == 1 && stride == 1) return launch_bc<T, TO, 1, 1>(p, s);
    if (ks == 3 && stride == 1) return launch_bc<T, TO, 3, 1>(p, s);
    if (ks == 3 && stride == 2) return launch_bc<T, TO, 3, 2>(p, s);
    if constexpr (sizeof(T) == 2) {
        // the 4x4 / pad-1 geometries of the PatchGAN discriminator (reference losses/discriminator.py:20-36: stride 2 and stride 1),
        // and the 4x4 stride-1 / pad-2 convolutions that are their data gradients -- bf16 input only, bf16 or fp32 output
        if (ks == 4 && stride == 1) return launch_bc<T, TO, 4, 1>(p, s);
        if (ks == 4 && stride == 2) return launch_bc<T, TO, 4, 2>(p, s);
    }
    MAS_FAIL(MAS_EUNSUPPORTED, "conv_fwd: unsupported ks=%d stride=%d", ks, stride);
}

}  // namespace

int mas_conv3x3_stream_try(const MasConvDesc* d, const void* x, const float* scale_shift, const void* w_packed, const float* bias,
                           const void* residual, void* y, hipStream_t s);   // conv3x3_stream.hip
bool mas_conv3x3_wide_eligible(const MasConvDesc* d);                        // conv3x3_wide.hip
int mas_conv3x3_wide_launch(const MasConvDesc* d, const void* x, const float* scale_shift, const void* w_packed, const float* bias,
                            const void* residual, void* y, float* stats, hipStream_t s);
int mas_conv3x3_wide_stat_rows(const MasConvDesc* d);
int mas_conv1x1_try(const MasConvDesc* d, const void* x, const void* w_packed, const float* bias, const void* residual, void* y, hipStream_t s);
int mas_conv_thin_fwd_try(const MasConvDesc* d, const void* x, const void* w_packed, const float* bias, const void* residual, void* y, hipStream_t s);
int mas_conv_thin_out_try(const MasConvDesc* d, const void* x, const void* w_packed, const float* bias, const void* residual, void* y, hipStream_t s);
int mas_conv_s2_fwd_try(const MasConvDesc* d, const void* x, const void* w_packed, const float* bias, const void* residual, void* y, hipStream_t s);
bool mas_conv_up2_fwd_eligible(const MasConvDesc* d);                        // conv_up2.hip
int mas_conv_up2_stat_rows(const MasConvDesc* d);
int mas_conv_up2_fwd_launch(const MasConvDesc* d, const void* x, const void* w_packed, const float* bias, void* y, float* stats, hipStream_t s);

// Can mas_conv_fwd_stats fill the consumer's GroupNorm statistics for this convolution, and with how many table rows per image?
extern "C" int mas_conv_stat_rows(const MasConvDesc* d) {
    if (d && d->w_layout == MAS_WLAYOUT_UP2) return mas_conv_up2_fwd_eligible(d) ? mas_conv_up2_stat_rows(d) : 0;
    if (!d || d->w_layout != MAS_WLAYOUT_K32 || !mas_conv3x3_wide_eligible(d)) return 0;
    return mas_conv3x3_wide_stat_rows(d);
}

// Upsample + 3x3 in its sub-pixel form (conv_up2.hip): a separate question from mas_conv_weight_layout, because the caller must also
// know that the call carries no residual and no prologue
extern "C" int mas_conv_up2_supported(const MasConvDesc* d) { return (d && mas_conv_up2_fwd_eligible(d)) ? 1 : 0; }

extern "C" int mas_conv_weight_layout(const MasConvDesc* d) {
    if (!d) return MAS_WLAYOUT_K64;
    return mas_conv3x3_wide_eligible(d) ? MAS_WLAYOUT_K32 : MAS_WLAYOUT_K64;
}

static int conv_fwd_impl(const MasConvDesc* d, const void* x, const float* scale_shift, const void* w_packed,
                         const float* bias, const void* residual, void* y, float* stats, void* stream);

extern "C" int mas_conv_fwd(const MasConvDesc* d, const void* x, const float* scale_shift, const void* w_packed,
                            const float* bias, const void* residual, void* y, void* stream) {
    return conv_fwd_impl(d, x, scale_shift, w_packed, bias, residual, y, nullptr, stream);
}

extern "C" int mas_conv_fwd_stats(const MasConvDesc* d, const void* x, const float* scale_shift, const void* w_packed,
                                  const float* bias, const void* residual, void* y, float* stats_partial, void* stream) {
    if (stats_partial && mas_conv_stat_rows(d) == 0) {
        MAS_ENTER();
        MAS_FAIL(MAS_EUNSUPPORTED, "conv_fwd_stats: this convolution does not take a kernel with fused statistics (mas_conv_stat_rows == 0)");
    }
    return conv_fwd_impl(d, x, scale_shift, w_packed, bias, residual, y, stats_partial, stream);
}

static int conv_fwd_impl(const MasConvDesc* d, const void* x, const float* scale_shift, const void* w_packed,
                         const float* bias, const void* residual, void* y, float* stats, void* stream) {
    MAS_ENTER();
    if (!d || !x || !w_packed || !y) MAS_FAIL(MAS_EINVAL, "conv_fwd: null argument");
    if (d->act != MAS_ACT_NONE && !scale_shift) MAS_FAIL(MAS_EINVAL, "conv_fwd: act prologue needs scale_shift");
    if (d->N <= 0 || d->H <= 0 || d->W <= 0 || d->Cin <= 0 || d->Cout <= 0 || d->Ho <= 0 || d->Wo <= 0)
        MAS_FAIL(MAS_EINVAL, "conv_fwd: non-positive dimension");
    if (d->upsample && d->stride != 1) MAS_FAIL(MAS_EUNSUPPORTED, "conv_fwd: upsample fold needs stride 1");
    if ((long long)d->H * d->W * d->Cin > 0x7fffffffLL) MAS_FAIL(MAS_EUNSUPPORTED, "conv_fwd: one image exceeds 2^31 elements");
    if (d->w_layout == MAS_WLAYOUT_UP2) {   // Upsample + conv, sub-pixel form: the caller asked mas_conv_up2_supported and packed for it
        if (!mas_conv_up2_fwd_eligible(d)) MAS_FAIL(MAS_EINVAL, "conv_fwd: w_layout UP2 but this convolution does not take the sub-pixel kernel");
        if (residual || d->act != MAS_ACT_NONE) MAS_FAIL(MAS_EUNSUPPORTED, "conv_fwd: the sub-pixel kernel takes no residual and no prologue");
        return mas_conv_up2_fwd_launch(d, x, w_packed, bias, y, stats, reinterpret_cast<hipStream_t>(stream));
    }
    if (d->w_layout == MAS_WLAYOUT_K32) {   // the caller packed for the wide 3x3 kernel (mas_conv_weight_layout said so)
        if (!mas_conv3x3_wide_eligible(d)) MAS_FAIL(MAS_EINVAL, "conv_fwd: w_layout K32 but this convolution does not take the wide kernel");
        return mas_conv3x3_wide_launch(d, x, scale_shift, w_packed, bias, residual, y, stats, reinterpret_cast<hipStream_t>(stream));
    }
    if (d->w_layout != MAS_WLAYOUT_K64) MAS_FAIL(MAS_EINVAL, "conv_fwd: bad w_layout %d", d->w_layout);
    {   // the FLOP-carrying shapes (3x3, stride 1, bf16, Cin/Cout multiples of 128) take the stream-scheduled kernel
        const int rc = mas_conv3x3_stream_try(d, x, scale_shift, w_packed, bias, residual, y, reinterpret_cast<hipStream_t>(stream));
        if (rc != 0) return rc < 0 ? rc : MAS_OK;
    }
    {   // plain 1x1 GEMMs (nin_shortcut, AttnBlock q / k / v / proj_out and their data gradients): conv1x1.hip
        int rc = mas_conv1x1_try(d, x, w_packed, bias, residual, y, reinterpret_cast<hipStream_t>(stream));
        if (rc != 0) return rc < 0 ? rc : MAS_OK;
        // 8 -> 128 channels (conv_in's forward, conv_out's data gradient): conv_thin.hip
        rc = mas_conv_thin_fwd_try(d, x, w_packed, bias, residual, y, reinterpret_cast<hipStream_t>(stream));
        if (rc != 0) return rc < 0 ? rc : MAS_OK;
        // 128 -> 8 channels (conv_out's forward): conv_thin.hip
        rc = mas_conv_thin_out_try(d, x, w_packed, bias, residual, y, reinterpret_cast<hipStream_t>(stream));
        if (rc != 0) return rc < 0 ? rc : MAS_OK;
        // Downsample (3x3 stride 2): conv_s2.hip
        rc = mas_conv_s2_fwd_try(d, x, w_packed, bias, residual, y, reinterpret_cast<hipStream_t>(stream));
        if (rc != 0) return rc < 0 ? rc : MAS_OK;
    }
    ConvParams p;
    p.dbg = nullptr;
#ifdef MAS_TIMELINE          // s_memtime timeline builds only (tools/build_variant.sh tl -DMAS_TIMELINE)
    if (const char* e = getenv("MAS_DBG_PTR")) p.dbg = reinterpret_cast<unsigned long long*>(strtoull(e, nullptr, 0));
#endif
    p.x = x; p.ss = scale_shift; p.w = w_packed; p.bias = bias; p.res = residual; p.y = y;
    p.N = d->N; p.H = d->H; p.W = d->W; p.Cin = d->Cin; p.Ho = d->Ho; p.Wo = d->Wo; p.Cout = d->Cout;
    p.Hl = d->upsample ? 2 * d->H : d->H; p.Wl = d->upsample ? 2 * d->W : d->W;
    p.pad_top = d->pad_top; p.pad_left = d->pad_left; p.act = d->act; p.upsample = d->upsample;
    const int ck = d->in_dtype == MAS_BF16 ? 64 : 32;
    p.n_chunks = mas_cdiv(d->Cin, ck); p.Cout_pad = mas_roundup(d->Cout, 128);
    p.tiles_h = p.tiles_w = 0; p.n_ct = 1;   // set per tile geometry at launch
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (d->in_dtype == MAS_BF16 && d->out_dtype == MAS_BF16) return launch_ks<bf16_t, bf16_t>(p, d->ks, d->stride, s);
    if (d->in_dtype == MAS_BF16 && d->out_dtype == MAS_F32) return launch_ks<bf16_t, float>(p, d->ks, d->stride, s);
    if (d->in_dtype == MAS_F32 && d->out_dtype == MAS_F32) return launch_ks<float, float>(p, d->ks, d->stride, s);
    MAS_FAIL(MAS_EUNSUPPORTED, "conv_fwd: unsupported dtype pair in=%d out=%d", d->in_dtype, d->out_dtype);
}
