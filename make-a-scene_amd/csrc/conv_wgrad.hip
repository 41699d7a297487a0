// Convolution weight gradient for gfx950 (autograd of the F.conv2d sites listed in conv_fwd.hip;
// reference models/modules.py:93-117,145-164,219,236,345,364, models/vqvae.py:15,18).
//
// GEMM view (per tap):  dW[tap][co][ci] = sum_pixels dY[pixel][co] * A[pixel (+) tap][ci]
//   M = co (MFMA A operand = dY^T), N = ci (MFMA B operand = activated input), K = pixels.
// Both operands need the contraction index (pixels) contiguous per lane, which NHWC does not give,
// so both are transposed while being staged into LDS ([channel][pixel] images).  The input halo
// patch is staged ONCE per pixel tile (prologue = GroupNorm apply + SiLU recomputed here, the
// activated tensor is never stored) and every tap reads a column-shifted fragment of it: an aligned
// 5-dword read + v_alignbit for the odd shift, plain register renaming for the even one.
// One work-group (8 waves) owns a 128(co) x 64(ci) x all-taps accumulator block in registers and
// walks a strided subset of the pixel tiles (split-K); each split STORES its partial sums into its own slab (WgradParams::slabs, round 6:
// mas_conv_wgrad_partial + the fixed-order mas_wgrad_reduce -- bitwise reproducible) or, through mas_conv_wgrad, commits with fp32 atomics.
#include "mas_common.h"
#include <stdlib.h>
#include <type_traits>

namespace {

struct WgradParams {
    unsigned long long* dbg;
    const void* x; const float* ss; const void* dy; float* dw; float* dbias;
    int N, H, W, Cin, Ho, Wo, Cout;
    int Hl, Wl, pad_top, pad_left, act, upsample;
    int tiles_h, tiles_w, n_pt, n_co_t, n_ci_t, nsplit;
    int slabs;             // 0: commit with fp32 atomics into dw / dbias (zeroed by the caller).  1 (round 6): split s STORES its partial
                           // sums into dw + s * Cout*KS*KS*Cin and dbias + s * Cout -- mas_wgrad_reduce adds the slabs in a fixed order
};

#ifndef MAS_WGRAD_BCI
#define MAS_WGRAD_BCI 32
#endif
// work-group = 128 co x BCI ci x all taps; 4 waves per 32 ci, so BCI = 32 -> 256 threads and two independent
// work-groups per CU (one stages while the other runs MFMAs), BCI = 64 -> 512 threads, one per CU
constexpr int BCI = MAS_WGRAD_BCI, TWW = 16, BCO = 128;
// 3x3: each 32x32 (co x ci) accumulator tile is shared by SPLIT = 2 waves that split the 9 taps 5 / 4, so a wave holds
// 80 accumulator VGPRs instead of 144 (which left no registers to batch the staging loads and forced spills) and the
// work-group is 16 waves = 4 per SIMD.  1x1: one wave per tile, 8 waves.
#ifndef MAS_WGRAD_SPLIT
#define MAS_WGRAD_SPLIT 2
#endif
template <int KS> struct WSplit { static constexpr int SPLIT = KS == 3 ? MAS_WGRAD_SPLIT : 1, NTW = 64 * 4 * (BCI / 32) * SPLIT; };

template <typename T, int KS, int STRIDE, int THW>
struct WGeo {
    static constexpr int EPU = 16 / (int)sizeof(T);
    static constexpr int NPIX = THW * TWW;
    // QUAD staging (bf16, stride 1; see the kernel): rows of channel unit cu are skewed by (cu&7)*16 bytes instead of
    // rotating elements, so the 16 channel units of a ds_write_b64 group land on different banks with zero VALU cost
    static constexpr bool QUAD = (STRIDE == 1) && (sizeof(T) == 2);
    static constexpr int DS = QUAD ? NPIX + 72 : NPIX + EPU;    // dY^T row stride (elements); 400 B = 25 slots (odd) in QUAD mode
    static constexpr int PH = (THW - 1) * STRIDE + KS, PW = (TWW - 1) * STRIDE + KS;
    static constexpr int PWA = 24;                              // padded plane width (>= 8*1+10)
    static constexpr int PLANES = STRIDE;                       // stride 2: even / odd input columns
    static constexpr int CS = QUAD ? PH * PLANES * PWA + 72 : PH * PLANES * PWA + EPU;   // per-channel stride; QUAD: odd number of 16-byte slots
    static constexpr size_t LDS_BYTES = (size_t)(BCO * DS + BCI * CS) * sizeof(T);
};

// 8 consecutive elements starting `shift` (0..2) elements after the 16-byte aligned pointer p
__device__ __forceinline__ bf16x8 shifted8(const bf16_t* p, int shift) {
    const u32x4 lo = *reinterpret_cast<const u32x4*>(p);
    const unsigned hi = *reinterpret_cast<const unsigned*>(p + 8);
    u32x4 r;
    if (shift == 0) r = lo;
    else if (shift == 1) {
        r[0] = __builtin_amdgcn_alignbit(lo[1], lo[0], 16); r[1] = __builtin_amdgcn_alignbit(lo[2], lo[1], 16);
        r[2] = __builtin_amdgcn_alignbit(lo[3], lo[2], 16); r[3] = __builtin_amdgcn_alignbit(hi, lo[3], 16);
    } else { r[0] = lo[1]; r[1] = lo[2]; r[2] = lo[3]; r[3] = hi; }
    return *reinterpret_cast<bf16x8*>(&r);
}
__device__ __forceinline__ f32x8 shifted8(const float* p, int shift) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(p);
    const f32x4 b = *reinterpret_cast<const f32x4*>(p + 4);
    const float c0 = p[8], c1 = p[9];
    const float v[10] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3], c0, c1};
    f32x8 r;
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = v[j + shift];
    return r;
}

// Rotates the EPU elements packed in a 16-byte register vector: out[e] = in[(e + r) % EPU] (select network on
// packed data, no dynamic register indexing).  Used so that at transposed-store step e lane `cu` writes channel
// (e + cu) % EPU: channel rows of one 16-byte unit are a multiple of 32 LDS banks apart, so without the rotation
// all channel units of a wave-instruction hit ONE bank (measured: ~27 cycles per ds_write_b16).
template <typename T> __device__ __forceinline__ u32x4 rot_elems(u32x4 v, int r);
template <> __device__ __forceinline__ u32x4 rot_elems<bf16_t>(u32x4 v, int r) {
    if (r & 1) {
        const u32x4 t = {__builtin_amdgcn_alignbit(v[1], v[0], 16), __builtin_amdgcn_alignbit(v[2], v[1], 16),
                         __builtin_amdgcn_alignbit(v[3], v[2], 16), __builtin_amdgcn_alignbit(v[0], v[3], 16)};
        v = t;
    }
    if (r & 2) { const u32x4 t = {v[1], v[2], v[3], v[0]}; v = t; }
    if (r & 4) { const u32x4 t = {v[2], v[3], v[0], v[1]}; v = t; }
    return v;
}
template <> __device__ __forceinline__ u32x4 rot_elems<float>(u32x4 v, int r) {
    if (r & 1) { const u32x4 t = {v[1], v[2], v[3], v[0]}; v = t; }
    if (r & 2) { const u32x4 t = {v[2], v[3], v[0], v[1]}; v = t; }
    return v;
}

// Transposed store of a 4-pixel x 8-channel bf16 block (o[j] = pixel j's 8 channels): for every channel the four pixels
// are packed with v_perm_b32 into one 8-byte ds_write_b64 at row cu*8+e, column `col` + the row skew.
__device__ __forceinline__ void store_quad_bf16(bf16_t* base, int row_stride, int cu, int col, const u32x4 (&o)[4]) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const unsigned lo01 = __builtin_amdgcn_perm(o[1][k], o[0][k], 0x05040100u), lo23 = __builtin_amdgcn_perm(o[3][k], o[2][k], 0x05040100u);
        const unsigned hi01 = __builtin_amdgcn_perm(o[1][k], o[0][k], 0x07060302u), hi23 = __builtin_amdgcn_perm(o[3][k], o[2][k], 0x07060302u);
        bf16_t* dst = base + (cu * 8 + 2 * k) * row_stride + (cu & 7) * 8 + col;    // rows skewed by (cu&7)*16 B
        *reinterpret_cast<u32x2*>(dst) = u32x2{lo01, lo23};
        *reinterpret_cast<u32x2*>(dst + row_stride) = u32x2{hi01, hi23};
    }
}

// Bias gradient of one work-group: every thread holds the column sums (EPU channels of unit `cu` = tid % UPP) of the dY elements it staged.
// Through LDS, added in THREAD ORDER (the first version used LDS float atomics: the order of the additions, and with it the last bits, changed
// run to run); then one store into this split's slab, or one fp32 atomic per channel into the caller's zeroed vector.
template <int NT, int EPU, int UPP>
__device__ __forceinline__ void commit_bias(unsigned char* smem, int tid, int cu, const float (&bsum)[EPU], float* dst, int n_live, int slabs) {
    static_assert(NT % UPP == 0, "threads of one channel unit are tid = cu, cu + UPP, ...");
    __syncthreads();                                     // the staging buffers are dead: every wave is past its last fragment read
    float* red = reinterpret_cast<float*>(smem);         // [NT / UPP][UPP * EPU]
#pragma unroll
    for (int e = 0; e < EPU; ++e) red[(tid / UPP) * (UPP * EPU) + cu * EPU + e] = bsum[e];
    __syncthreads();
    for (int i = tid; i < UPP * EPU; i += NT) {
        float v = 0.0f;
        for (int k = 0; k < NT / UPP; ++k) v += red[k * (UPP * EPU) + i];
        if (i < n_live) {
            if (slabs) dst[i] = v;
            else atomicAdd(dst + i, v);
        }
    }
}

template <typename T, int KS, int STRIDE, int THW>
__global__ __launch_bounds__(WSplit<KS>::NTW) void conv_wgrad_kernel(WgradParams p) {
    using G = WGeo<T, KS, STRIDE, THW>;
    constexpr int NTW = WSplit<KS>::NTW, SPLIT = WSplit<KS>::SPLIT;
    using V8 = typename Vec8<T>::type;
    constexpr int EPU = G::EPU, DS = G::DS, CS = G::CS, PW = G::PW, PH = G::PH, PWA = G::PWA;
    constexpr int NTAP = KS * KS;
    constexpr int DY_UPP = BCO / EPU;            // 16-byte units per pixel of the dY tile
    constexpr int DY_UNITS = G::NPIX * DY_UPP;
    constexpr int A_UPP = BCI / EPU;
    constexpr int A_UNITS = PH * PW * A_UPP;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    T* dyt = reinterpret_cast<T*>(smem);         // [BCO][DS]
    T* at = dyt + BCO * DS;                      // [BCI][CS]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 5, l31 = lane & 31;
    constexpr int NTILE = 4 * (BCI / 32);        // 32x32 accumulator tiles per work-group
    const int wtile = wave % NTILE, half = wave / NTILE;
    const int wco = (wtile & 3) * 32, wci = (wtile >> 2) * 32;
    constexpr int TPH = (KS * KS + SPLIT - 1) / SPLIT;   // taps per wave

    int bid = blockIdx.x;
    const int split = bid % p.nsplit; bid /= p.nsplit;
    const int ci_t = bid % p.n_ci_t; const int co_t = bid / p.n_ci_t;
    const int co0 = co_t * BCO, ci0 = ci_t * BCI;

    const T* __restrict__ X = reinterpret_cast<const T*>(p.x);
    const T* __restrict__ DY = reinterpret_cast<const T*>(p.dy);

    f32x16 acc[TPH];                             // taps [half*TPH, min(NTAP, (half+1)*TPH))
#pragma unroll
    for (int t = 0; t < TPH; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;

    const bool vec_dy = (p.Cout % EPU) == 0, vec_x = (p.Cin % EPU) == 0;
    const int dy_cu = tid % DY_UPP, a_cu = tid % A_UPP;
    float bsum[EPU];
#pragma unroll
    for (int e = 0; e < EPU; ++e) bsum[e] = 0.0f;
    const bool do_bias = (p.dbias != nullptr) && (ci_t == 0);

    const int skew_a = G::QUAD ? (((wco + l31) >> 3) & 7) * 8 : 0, skew_b = G::QUAD ? (((wci + l31) >> 3) & 7) * 8 : 0;
    const T* a_frag_base = dyt + (wco + l31) * DS + skew_a + g * 8;
    const T* b_frag_base = at + (wci + l31) * CS + skew_b + g * 8;

#ifdef MAS_TIMELINE
    int tl_iter = 0;
#define WTS(id) do { if (lane == 0 && blockIdx.x == 100 && tl_iter < 4 && p.dbg) p.dbg[(tl_iter * 8 + wave) * 8 + (id)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define WTS(id) do {} while (0)
#endif
    // ---- fast staging (16-byte aligned channel slots): the global loads of tile k+1 are issued into registers right
    // before the MFMA phase of tile k and converted / transposed into LDS after it, so HBM latency hides under the MFMAs
    const bool fast = vec_dy && vec_x;
    // QUAD staging (bf16, stride 1): a thread owns 4 ADJACENT pixels of one 16-byte channel unit, so the transposed
    // store is 8 x ds_write_b64 (4 pixels of one channel) instead of 32 x ds_write_b16 -- LDS store issue was the
    // measured cost of staging (~10 cycles per wave-level ds_write_b16)
    constexpr bool QUAD = G::QUAD;
    constexpr int NG = (PW + 3) / 4;             // 4-column groups per patch row
    constexpr int A_QUADS = (PH * NG * A_UPP + NTW - 1) / NTW;
    constexpr int DY_PER_T = (DY_UNITS + NTW - 1) / NTW, A_PER_T = QUAD ? 4 * A_QUADS : (A_UNITS + NTW - 1) / NTW;
    u32x4 rdy[DY_PER_T];                         // dY slots of the NEXT tile, in flight across the MFMA phase
    const int cbd = co0 + dy_cu * EPU, cba = ci0 + a_cu * EPU;
    auto coords = [&](int pt, int& n, int& h0, int& w0) {
        int t = pt;
        const int tw_i = t % p.tiles_w; t /= p.tiles_w;
        const int th_i = t % p.tiles_h; n = t / p.tiles_h;
        h0 = th_i * THW; w0 = tw_i * TWW;
    };
    auto dy_ok = [&](int i, int h0, int w0, int& pix) {
        const int u = tid + i * NTW;
        if constexpr (QUAD) pix = 4 * (tid / DY_UPP + (i >> 2) * (NTW / DY_UPP)) + (i & 3);
        else pix = u / DY_UPP;
        const int ho = h0 + (pix >> 4), wo = w0 + (pix & 15);
        return (QUAD || u < DY_UNITS) && (ho < p.Ho) && (wo < p.Wo) && (cbd < p.Cout);
    };
    auto a_ok = [&](int i, int h0, int w0, int& ih, int& iw, int& off) {
        const int u = tid + i * NTW;
        int pr, pc;
        bool live;
        if constexpr (QUAD) {
            const int grp = tid / A_UPP + (i >> 2) * (NTW / A_UPP);
            pr = grp / NG; pc = 4 * (grp - pr * NG) + (i & 3);
            live = (grp < PH * NG) && (pc < PW);
        } else {
            const int pp = u / A_UPP;
            pr = pp / PW; pc = pp - pr * PW;
            live = u < A_UNITS;
        }
        ih = h0 * STRIDE + pr - p.pad_top; iw = w0 * STRIDE + pc - p.pad_left;
        const bool ok = live && (ih >= 0) && (ih < p.Hl) && (iw >= 0) && (iw < p.Wl) && (cba < p.Cin);
        if (p.upsample) { ih >>= 1; iw >>= 1; }
        off = (STRIDE == 1) ? (pr * PWA + pc) : (pr * 2 * PWA + (pc & 1) * PWA + (pc >> 1));
        return ok;
    };
    auto issue = [&](int pt) {                   // unconditional (clamped) loads: a branch-free VMEM stream
        int n, h0, w0;
        coords(pt, n, h0, w0);
#pragma unroll
        for (int i = 0; i < DY_PER_T; ++i) {
            int pix;
            const bool ok = dy_ok(i, h0, w0, pix);
            const size_t off = ok ? ((size_t)(n * p.Ho + h0 + (pix >> 4)) * p.Wo + w0 + (pix & 15)) * p.Cout + cbd : 0;
            rdy[i] = *reinterpret_cast<const u32x4*>(DY + off);
        }
    };
    auto commit_tile = [&](int pt) {             // registers -> (prologue) -> transposed LDS images
        int n, h0, w0;
        coords(pt, n, h0, w0);
        u32x4 ra[A_PER_T];                       // input patch slots: issued here, land while the dY stores run
#pragma unroll
        for (int i = 0; i < A_PER_T; ++i) {
            int ih, iw, o;
            const bool ok = a_ok(i, h0, w0, ih, iw, o);
            const size_t off = ok ? ((size_t)(n * p.H + ih) * p.W + iw) * p.Cin + cba : 0;
            ra[i] = *reinterpret_cast<const u32x4*>(X + off);
        }
        f32x4 rss[EPU / 2];                      // scale/shift of this thread's channels (L2 hit; lands during the dY stores)
        if (p.act != MAS_ACT_NONE) {
            const int cc = (cba + EPU <= p.Cin) ? cba : (p.Cin - EPU);
            const f32x4* sp = reinterpret_cast<const f32x4*>(p.ss + ((size_t)n * p.Cin + cc) * 2);
#pragma unroll
            for (int q = 0; q < EPU / 2; ++q) rss[q] = sp[q];
        }
        if constexpr (QUAD) {
#pragma unroll
            for (int qi = 0; qi < DY_PER_T / 4; ++qi) {
                u32x4 o[4];
                int pix0 = 0;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    int pix;
                    const bool ok = dy_ok(qi * 4 + j, h0, w0, pix);
                    if (j == 0) pix0 = pix;
                    u32x4 raw = rdy[qi * 4 + j];
                    if (!ok) raw = u32x4{0u, 0u, 0u, 0u};
                    if (do_bias) {
                        const T* rv = reinterpret_cast<const T*>(&raw);
#pragma unroll
                        for (int e = 0; e < EPU; ++e) bsum[e] += (float)rv[e];
                    }
                    o[j] = raw;
                }
                store_quad_bf16(reinterpret_cast<bf16_t*>(dyt), DS, dy_cu, pix0, o);
            }
            WTS(2);
#pragma unroll
            for (int qi = 0; qi < A_PER_T / 4; ++qi) {
                const int grp = tid / A_UPP + qi * (NTW / A_UPP);
                if (grp >= PH * NG) continue;
                u32x4 o[4];
                int off0 = 0;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    int ih, iw, off;
                    const bool ok = a_ok(qi * 4 + j, h0, w0, ih, iw, off);
                    if (j == 0) off0 = off;
                    u32x4 raw = ra[qi * 4 + j];
                    if (!ok) raw = u32x4{0u, 0u, 0u, 0u};
                    else if (p.act != MAS_ACT_NONE) {
                        T* tv = reinterpret_cast<T*>(&raw);
#pragma unroll
                        for (int e = 0; e < EPU; ++e) {
                            const float sc = rss[e >> 1][(e & 1) * 2], sh = rss[e >> 1][(e & 1) * 2 + 1];
                            float a = (float)tv[e] * sc + sh;
                            if (p.act == MAS_ACT_AFFINE_SILU) a = silu_f(a);
                            tv[e] = (T)((cba + e < p.Cin) ? a : 0.0f);
                        }
                    }
                    o[j] = raw;
                }
                store_quad_bf16(reinterpret_cast<bf16_t*>(at), CS, a_cu, off0, o);
            }
        } else {
#pragma unroll
        for (int i = 0; i < DY_PER_T; ++i) {
            int pix;
            const bool ok = dy_ok(i, h0, w0, pix);
            if (tid + i * NTW >= DY_UNITS) continue;
            u32x4 raw = rdy[i];
            if (!ok) raw = u32x4{0u, 0u, 0u, 0u};
            if (do_bias) {
                const T* rv = reinterpret_cast<const T*>(&raw);
#pragma unroll
                for (int e = 0; e < EPU; ++e) bsum[e] += (float)rv[e];
            }
            const u32x4 rot = rot_elems<T>(raw, dy_cu);
            const T* ov = reinterpret_cast<const T*>(&rot);
#pragma unroll
            for (int e = 0; e < EPU; ++e) dyt[(dy_cu * EPU + ((e + dy_cu) & (EPU - 1))) * DS + pix] = ov[e];
        }
        WTS(2);
#pragma unroll
        for (int i = 0; i < A_PER_T; ++i) {
            int ih, iw, off;
            const bool ok = a_ok(i, h0, w0, ih, iw, off);
            if (tid + i * NTW >= A_UNITS) continue;
            u32x4 raw = ra[i];
            if (!ok) raw = u32x4{0u, 0u, 0u, 0u};
            else if (p.act != MAS_ACT_NONE) {
                T* tv = reinterpret_cast<T*>(&raw);
#pragma unroll
                for (int e = 0; e < EPU; ++e) {
                    const float sc = rss[e >> 1][(e & 1) * 2], sh = rss[e >> 1][(e & 1) * 2 + 1];
                    float a = (float)tv[e] * sc + sh;
                    if (p.act == MAS_ACT_AFFINE_SILU) a = silu_f(a);
                    tv[e] = (T)((cba + e < p.Cin) ? a : 0.0f);
                }
            }
            const u32x4 rot = rot_elems<T>(raw, a_cu);
            const T* ov = reinterpret_cast<const T*>(&rot);
#pragma unroll
            for (int e = 0; e < EPU; ++e) at[(a_cu * EPU + ((e + a_cu) & (EPU - 1))) * CS + off] = ov[e];
        }
        }
    };
    if (fast && split < p.n_pt) issue(split);

    for (int pt = split; pt < p.n_pt; pt += p.nsplit) {
        int t = pt;
        const int tw_i = t % p.tiles_w; t /= p.tiles_w;
        const int th_i = t % p.tiles_h; const int n = t / p.tiles_h;
        const int h0 = th_i * THW, w0 = tw_i * TWW;
        WTS(0);
        __syncthreads();                         // previous tile's fragment reads are done
        WTS(1);
        if (fast) {
            commit_tile(pt);
        } else {
        // ---- stage dY^T ------------------------------------------------------------
        for (int u = tid; u < DY_UNITS; u += NTW) {
            const int pix = u / DY_UPP;
            const int ho = h0 + (pix >> 4), wo = w0 + (pix & 15);
            const int cb = co0 + dy_cu * EPU;
            float v[EPU];
#pragma unroll
            for (int e = 0; e < EPU; ++e) v[e] = 0.0f;
            if (ho < p.Ho && wo < p.Wo && cb < p.Cout) {
                const T* src = DY + ((size_t)(n * p.Ho + ho) * p.Wo + wo) * p.Cout + cb;
                if (vec_dy) {
                    u32x4 raw = *reinterpret_cast<const u32x4*>(src);
                    const T* rv = reinterpret_cast<const T*>(&raw);
#pragma unroll
                    for (int e = 0; e < EPU; ++e) v[e] = (float)rv[e];
                } else {
#pragma unroll
                    for (int e = 0; e < EPU; ++e) if (cb + e < p.Cout) v[e] = (float)src[e];
                }
            }
            if (do_bias) {
#pragma unroll
                for (int e = 0; e < EPU; ++e) bsum[e] += v[e];
            }
#pragma unroll
            for (int e = 0; e < EPU; ++e) dyt[(dy_cu * EPU + e) * DS + (G::QUAD ? (dy_cu & 7) * 8 : 0) + pix] = (T)v[e];
        }
        // ---- stage A^T (activated input patch) ---------------------------------------
        {
            const int cb = ci0 + a_cu * EPU;
            float sc[EPU], sh[EPU];
            if (p.act != MAS_ACT_NONE) {
#pragma unroll
                for (int e = 0; e < EPU; ++e) {
                    const int c = cb + e;
                    sc[e] = (c < p.Cin) ? p.ss[((size_t)n * p.Cin + c) * 2 + 0] : 0.0f;
                    sh[e] = (c < p.Cin) ? p.ss[((size_t)n * p.Cin + c) * 2 + 1] : 0.0f;
                }
            }
            for (int u = tid; u < A_UNITS; u += NTW) {
                const int pp = u / A_UPP;
                const int pr = pp / PW, pc = pp - pr * PW;
                int ih = h0 * STRIDE + pr - p.pad_top, iw = w0 * STRIDE + pc - p.pad_left;
                const bool inb = (ih >= 0) && (ih < p.Hl) && (iw >= 0) && (iw < p.Wl);
                if (p.upsample) { ih >>= 1; iw >>= 1; }
                float v[EPU];
#pragma unroll
                for (int e = 0; e < EPU; ++e) v[e] = 0.0f;
                if (inb && cb < p.Cin) {
                    const T* src = X + ((size_t)(n * p.H + ih) * p.W + iw) * p.Cin + cb;
                    if (vec_x) {
                        u32x4 raw = *reinterpret_cast<const u32x4*>(src);
                        const T* rv = reinterpret_cast<const T*>(&raw);
#pragma unroll
                        for (int e = 0; e < EPU; ++e) v[e] = (float)rv[e];
                    } else {
#pragma unroll
                        for (int e = 0; e < EPU; ++e) if (cb + e < p.Cin) v[e] = (float)src[e];
                    }
                    if (p.act != MAS_ACT_NONE) {
#pragma unroll
                        for (int e = 0; e < EPU; ++e) {
                            float a = v[e] * sc[e] + sh[e];
                            if (p.act == MAS_ACT_AFFINE_SILU) a = silu_f(a);
                            v[e] = (cb + e < p.Cin) ? a : 0.0f;
                        }
                    }
                }
                const int off = (STRIDE == 1) ? (pr * PWA + pc) : (pr * 2 * PWA + (pc & 1) * PWA + (pc >> 1));
#pragma unroll
                for (int e = 0; e < EPU; ++e) at[(a_cu * EPU + e) * CS + (G::QUAD ? (a_cu & 7) * 8 : 0) + off] = (T)v[e];
            }
        }
        }
        WTS(3);
        __syncthreads();
        WTS(4);
        if (fast) {                              // next tile's loads fly during this tile's MFMAs (a harmless re-read at the end)
            const int npt = (pt + p.nsplit < p.n_pt) ? pt + p.nsplit : pt;
            issue(npt);
        }
        // ---- MFMA: for every patch row, every tap of this wave's range that touches it -----------------
        auto mfma_phase = [&](auto HALF) {
            constexpr int LO = decltype(HALF)::value * TPH, HI = (LO + TPH < NTAP) ? LO + TPH : NTAP;
#pragma unroll
            for (int pr = 0; pr < PH; ++pr) {
                V8 bf[KS];
#pragma unroll
                for (int kw = 0; kw < KS; ++kw) {
                    bool need = false;
#pragma unroll
                    for (int kh = 0; kh < KS; ++kh) {
                        const int rr = pr - kh, tap = kh * KS + kw;
                        if (rr >= 0 && (rr % STRIDE) == 0 && (rr / STRIDE) < THW && tap >= LO && tap < HI) need = true;
                    }
                    if (!need) continue;
                    const int plane = (STRIDE == 1) ? 0 : (kw & 1);
                    const int shift = (STRIDE == 1) ? kw : (kw >> 1);
                    bf[kw] = shifted8(b_frag_base + (pr * G::PLANES + plane) * PWA, shift);
                }
#pragma unroll
                for (int kh = 0; kh < KS; ++kh) {
                    const int rr = pr - kh;
                    if (rr < 0 || (rr % STRIDE) != 0 || (rr / STRIDE) >= THW) continue;
                    if (kh * KS + KS <= LO || kh * KS >= HI) continue;
                    const V8 af = ld8<T>(a_frag_base + (rr / STRIDE) * 16);
#pragma unroll
                    for (int kw = 0; kw < KS; ++kw) {
                        const int tap = kh * KS + kw;
                        if (tap >= LO && tap < HI) mma16(acc[tap - LO], af, bf[kw]);
                    }
                }
            }
        };
        if (SPLIT == 1 || half == 0) mfma_phase(std::integral_constant<int, 0>{});
        else mfma_phase(std::integral_constant<int, SPLIT - 1>{});
        WTS(5);
#ifdef MAS_TIMELINE
        ++tl_iter;
#endif
    }

    // ---- commit: dW[co][kh][kw][ci], fp32 atomics or this split's slab --------------------------------
    const int ci = ci0 + wci + l31;
    float* dwp = p.dw + (p.slabs ? (size_t)split * p.Cout * NTAP * p.Cin : 0);
    auto commit = [&](auto HALF) {
        constexpr int LO = decltype(HALF)::value * TPH, HI = (LO + TPH < NTAP) ? LO + TPH : NTAP;
        if (ci < p.Cin) {
#pragma unroll
            for (int t = LO; t < HI; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int co = co0 + wco + acc_row(lane, r);
                    if (co < p.Cout) {
                        float* dst = dwp + ((size_t)co * NTAP + t) * p.Cin + ci;
                        if (p.slabs) *dst = acc[t - LO][r];
                        else atomicAdd(dst, acc[t - LO][r]);
                    }
                }
        }
    };
    if (SPLIT == 1 || half == 0) commit(std::integral_constant<int, 0>{});
    else commit(std::integral_constant<int, SPLIT - 1>{});
    if (do_bias) commit_bias<NTW, EPU, DY_UPP>(smem, tid, dy_cu, bsum, p.dbias + (p.slabs ? (size_t)split * p.Cout : 0) + co0, p.Cout - co0, p.slabs);
}


// =========================================================================================================
// bf16 / stride-1 weight gradient on the LDS TRANSPOSE READ (ds_read_b64_tr_b16).
// The K dimension of the wgrad GEMM is the pixel index -- the OUTER dimension of both NHWC operands.  The kernel
// above transposes both operands while staging them; here they are staged in their natural [pixel][channel] layout
// (plain 16-byte copies) and the MFMA fragments are fetched with the hardware transpose read.  Semantics measured
// on gfx950 (tools/probes/tr_probe.hip): inside a 16-lane group lane s reads the 8 bytes at ITS address and lane l
// receives { M[4j + (l>>2)][l&3] : j=0..3 } where M[s][e] is element e of lane s's 8 bytes.  With lane s pointed at
// pixel (s>>2), channels 4(s&3)..+3, lane l ends up with channel l of 4 consecutive pixels -- an MFMA operand run.
// Per-lane addresses also make the 3x3 tap shifts free (any pixel offset is just an address).
typedef __attribute__((ext_vector_type(4))) short s16x4;
__device__ __forceinline__ bf16x8 tr_frag(const unsigned char* a0, const unsigned char* a1) {
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)a0);
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)a1);
    const __attribute__((ext_vector_type(8))) short v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return *reinterpret_cast<const bf16x8*>(&v);
}

template <int KS, int BCI_>
struct TrGeo {
    static constexpr int THW = 8, NPIX = THW * TWW, PH = THW + KS - 1, PW = TWW + KS - 1;
    static constexpr int RSY = BCO * 2 + 64;                 // dY row stride (bytes): 4 consecutive pixel rows x 64 B tile the 256-B bank row
    static constexpr int RSX = BCI_ == 32 ? 64 : BCI_ * 2 + 64;
    static constexpr int DY_BYTES = NPIX * RSY, X_BYTES = PH * PW * RSX;
    static constexpr size_t LDS_BYTES = (size_t)DY_BYTES + X_BYTES;
    static constexpr int NTILE = 4 * (BCI_ / 32), SPLIT = 8 / NTILE, NT = 512;
    static constexpr int TPH = (KS * KS + SPLIT - 1) / SPLIT;
};

template <int KS, int BCI_>
__global__ __launch_bounds__(512) void conv_wgrad_tr_kernel(WgradParams p) {
    using G = TrGeo<KS, BCI_>;
    using T = bf16_t;
    constexpr int NT = G::NT, THW = G::THW, PH = G::PH, PW = G::PW, RSY = G::RSY, RSX = G::RSX, NTAP = KS * KS;
    constexpr int SPLIT = G::SPLIT, TPH = G::TPH, NTILE = G::NTILE;
    constexpr int DY_PER_T = G::NPIX * (BCO / 8) / NT;                       // 16-byte slots per thread: 4
    constexpr int X_UPP = BCI_ / 8, X_UNITS = PH * PW * X_UPP, X_PER_T = (X_UNITS + NT - 1) / NT;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* dys = smem;                   // [128 px][RSY]
    unsigned char* xs = smem + G::DY_BYTES;      // [PH*PW px][RSX]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 5, l31 = lane & 31, G16 = (lane >> 4) & 1, sl = lane & 15;
    const int wtile = wave % NTILE, half = wave / NTILE;
    const int wco = (wtile & 3) * 32, wci = (wtile >> 2) * 32;

    int bid = blockIdx.x;
    const int split = bid % p.nsplit; bid /= p.nsplit;
    const int ci_t = bid % p.n_ci_t; const int co_t = bid / p.n_ci_t;
    const int co0 = co_t * BCO, ci0 = ci_t * BCI_;
    const T* __restrict__ X = reinterpret_cast<const T*>(p.x);
    const T* __restrict__ DY = reinterpret_cast<const T*>(p.dy);

    f32x16 acc[TPH];
#pragma unroll
    for (int t = 0; t < TPH; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;

    // transpose-read lane addressing: pixel (8g + (sl>>2)) of a k-chunk, channels 16*G16 + 4*(sl&3) .. +3
    const unsigned char* a_lane = dys + (8 * g + (sl >> 2)) * RSY + (wco + 16 * G16 + 4 * (sl & 3)) * 2;
    const unsigned char* b_lane = xs + (8 * g + (sl >> 2)) * RSX + (wci + 16 * G16 + 4 * (sl & 3)) * 2;

    const int dy_cu = tid & 15, x_cu = tid % X_UPP;
    const int cbd = co0 + dy_cu * 8, cbx = ci0 + x_cu * 8;
    float bsum[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) bsum[e] = 0.0f;
    const bool do_bias = (p.dbias != nullptr) && (ci_t == 0);

    auto coords = [&](int pt, int& n, int& h0, int& w0) {
        int t = pt;
        const int tw_i = t % p.tiles_w; t /= p.tiles_w;
        const int th_i = t % p.tiles_h; n = t / p.tiles_h;
        h0 = th_i * THW; w0 = tw_i * TWW;
    };
    u32x4 rdy[DY_PER_T];                         // dY slots of the next tile: in flight across the MFMA phase
    auto issue = [&](int pt) {
        int n, h0, w0;
        coords(pt, n, h0, w0);
#pragma unroll
        for (int i = 0; i < DY_PER_T; ++i) {
            const int pix = (tid >> 4) + i * (NT / 16);
            const int ho = h0 + (pix >> 4), wo = w0 + (pix & 15);
            const bool ok = (ho < p.Ho) && (wo < p.Wo) && (cbd < p.Cout);
            const size_t off = ok ? ((size_t)(n * p.Ho + ho) * p.Wo + wo) * p.Cout + cbd : 0;
            rdy[i] = *reinterpret_cast<const u32x4*>(DY + off);
        }
    };
    auto commit_tile = [&](int pt) {
        int n, h0, w0;
        coords(pt, n, h0, w0);
        u32x4 rx[X_PER_T];
        bool okx[X_PER_T];
#pragma unroll
        for (int i = 0; i < X_PER_T; ++i) {      // input patch slots: issued first, land while the dY stores run
            const int u = tid + i * NT;
            const int pp = u / X_UPP;
            const int pr = pp / PW, pc = pp - pr * PW;
            int ih = h0 + pr - p.pad_top, iw = w0 + pc - p.pad_left;
            okx[i] = (u < X_UNITS) && (ih >= 0) && (ih < p.Hl) && (iw >= 0) && (iw < p.Wl) && (cbx < p.Cin);
            if (p.upsample) { ih >>= 1; iw >>= 1; }
            const size_t off = okx[i] ? ((size_t)(n * p.H + ih) * p.W + iw) * p.Cin + cbx : 0;
            rx[i] = *reinterpret_cast<const u32x4*>(X + off);
        }
        f32x4 rss[4];
        if (p.act != MAS_ACT_NONE) {
            const int cc = (cbx + 8 <= p.Cin) ? cbx : (p.Cin - 8);
            const f32x4* sp = reinterpret_cast<const f32x4*>(p.ss + ((size_t)n * p.Cin + cc) * 2);
#pragma unroll
            for (int q = 0; q < 4; ++q) rss[q] = sp[q];
        }
#pragma unroll
        for (int i = 0; i < DY_PER_T; ++i) {     // natural layout: one ds_write_b128 per slot
            const int pix = (tid >> 4) + i * (NT / 16);
            const int ho = h0 + (pix >> 4), wo = w0 + (pix & 15);
            u32x4 raw = rdy[i];
            if (!((ho < p.Ho) && (wo < p.Wo) && (cbd < p.Cout))) raw = u32x4{0u, 0u, 0u, 0u};
            if (do_bias) {
                const T* rv = reinterpret_cast<const T*>(&raw);
#pragma unroll
                for (int e = 0; e < 8; ++e) bsum[e] += (float)rv[e];
            }
            *reinterpret_cast<u32x4*>(dys + pix * RSY + dy_cu * 16) = raw;
        }
#pragma unroll
        for (int i = 0; i < X_PER_T; ++i) {
            const int u = tid + i * NT;
            if (u >= X_UNITS) continue;
            u32x4 raw = rx[i];
            if (!okx[i]) raw = u32x4{0u, 0u, 0u, 0u};
            else if (p.act != MAS_ACT_NONE) {             // Cin % 8 == 0 on this path: the slot is entirely inside the tensor
                T* tv = reinterpret_cast<T*>(&raw);
                if (p.act == MAS_ACT_AFFINE_SILU) {       // wave-uniform: two straight-line bodies, no per-element selects
#pragma unroll
                    for (int e = 0; e < 8; ++e) tv[e] = (T)silu_f((float)tv[e] * rss[e >> 1][(e & 1) * 2] + rss[e >> 1][(e & 1) * 2 + 1]);
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) tv[e] = (T)((float)tv[e] * rss[e >> 1][(e & 1) * 2] + rss[e >> 1][(e & 1) * 2 + 1]);
                }
            }
            *reinterpret_cast<u32x4*>(xs + (u / X_UPP) * RSX + x_cu * 16) = raw;
        }
    };

    if (split < p.n_pt) issue(split);
    for (int pt = split; pt < p.n_pt; pt += p.nsplit) {
        __syncthreads();                         // previous tile's fragment reads are done
        commit_tile(pt);
        __syncthreads();
        issue((pt + p.nsplit < p.n_pt) ? pt + p.nsplit : pt);     // next tile's dY (a harmless re-read at the end)
        auto mfma_phase = [&](auto HALF) {
            constexpr int LO = decltype(HALF)::value * TPH, HI = (LO + TPH < NTAP) ? LO + TPH : NTAP;
            bf16x8 aw[KS];                           // sliding window of dY rows: row rr lives in aw[rr % KS] for KS patch rows
#pragma unroll
            for (int pr = 0; pr < PH; ++pr) {
                bf16x8 bf[KS];
#pragma unroll
                for (int kw = 0; kw < KS; ++kw) {
                    bool need = false;
#pragma unroll
                    for (int kh = 0; kh < KS; ++kh) {
                        const int rr = pr - kh, tap = kh * KS + kw;
                        if (rr >= 0 && rr < THW && tap >= LO && tap < HI) need = true;
                    }
                    if (!need) continue;
                    const unsigned char* b0 = b_lane + (pr * PW + kw) * RSX;    // patch pixels (pr, kw + 8g + j)
                    bf[kw] = tr_frag(b0, b0 + 4 * RSX);
                }
                if (pr < THW) {
                    const unsigned char* a0 = a_lane + (pr * 16) * RSY;          // dY pixels (pr, 8g + j)
                    aw[pr % KS] = tr_frag(a0, a0 + 4 * RSY);
                }
#pragma unroll
                for (int kh = 0; kh < KS; ++kh) {
                    const int rr = pr - kh;
                    if (rr < 0 || rr >= THW) continue;
                    if (kh * KS + KS <= LO || kh * KS >= HI) continue;
#pragma unroll
                    for (int kw = 0; kw < KS; ++kw) {
                        const int tap = kh * KS + kw;
                        if (tap >= LO && tap < HI) mma16(acc[tap - LO], aw[rr % KS], bf[kw]);
                    }
                }
            }
        };
        if (SPLIT == 1 || half == 0) mfma_phase(std::integral_constant<int, 0>{});
        else mfma_phase(std::integral_constant<int, SPLIT - 1>{});
    }

    const int ci = ci0 + wci + l31;
    float* dwp = p.dw + (p.slabs ? (size_t)split * p.Cout * NTAP * p.Cin : 0);
    auto commit = [&](auto HALF) {
        constexpr int LO = decltype(HALF)::value * TPH, HI = (LO + TPH < NTAP) ? LO + TPH : NTAP;
        if (ci < p.Cin) {
#pragma unroll
            for (int t = LO; t < HI; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int co = co0 + wco + acc_row(lane, r);
                    if (co < p.Cout) {
                        float* dst = dwp + ((size_t)co * NTAP + t) * p.Cin + ci;
                        if (p.slabs) *dst = acc[t - LO][r];
                        else atomicAdd(dst, acc[t - LO][r]);
                    }
                }
        }
    };
    if (SPLIT == 1 || half == 0) commit(std::integral_constant<int, 0>{});
    else commit(std::integral_constant<int, SPLIT - 1>{});
    if (do_bias) commit_bias<NT, 8, 16>(smem, tid, dy_cu, bsum, p.dbias + (p.slabs ? (size_t)split * p.Cout : 0) + co0, p.Cout - co0, p.slabs);
}

#ifndef MAS_WGRAD_TR_BCI
#define MAS_WGRAD_TR_BCI 64
#endif
template <int KS, int BCI_ = MAS_WGRAD_TR_BCI>
int launch_tr(WgradParams p, hipStream_t s, int* splits_only) {
    using G = TrGeo<KS, BCI_>;
    auto kern = conv_wgrad_tr_kernel<KS, BCI_>;
    static mas_devmask_t attr_mask{0};
    unsigned long long attr_bit;
    if (mas_attr_needed(attr_mask, &attr_bit)) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::LDS_BYTES) != hipSuccess)
            MAS_FAIL(MAS_ELAUNCH, "conv_wgrad_tr: cannot set dynamic LDS size %zu", (size_t)G::LDS_BYTES);
        mas_attr_done(attr_mask, attr_bit);
    }
    p.tiles_h = mas_cdiv(p.Ho, G::THW); p.tiles_w = mas_cdiv(p.Wo, TWW);
    p.n_pt = p.N * p.tiles_h * p.tiles_w;
    p.n_co_t = mas_cdiv(p.Cout, BCO); p.n_ci_t = mas_cdiv(p.Cin, BCI_);
    const int out_tiles = p.n_co_t * p.n_ci_t;
    int nsplit = mas_cdiv(mas_num_cus(), out_tiles);
    if (nsplit > p.n_pt) nsplit = p.n_pt;
    if (nsplit < 1) nsplit = 1;
    p.nsplit = nsplit;
    if (splits_only) { *splits_only = nsplit; return MAS_OK; }
    hipLaunchKernelGGL(kern, dim3((unsigned)(out_tiles * nsplit)), dim3(G::NT), G::LDS_BYTES, s, p);
    MAS_CHECK_LAUNCH("conv_wgrad_tr");
    return MAS_OK;
}

template <typename T, int KS, int STRIDE, int THW>
int launch(WgradParams p, hipStream_t s, int* splits_only) {
    using G = WGeo<T, KS, STRIDE, THW>;
    auto kern = conv_wgrad_kernel<T, KS, STRIDE, THW>;
    static mas_devmask_t attr_mask{0};
    unsigned long long attr_bit;
    if (mas_attr_needed(attr_mask, &attr_bit)) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::LDS_BYTES) != hipSuccess)
            MAS_FAIL(MAS_ELAUNCH, "conv_wgrad: cannot set dynamic LDS size %zu", (size_t)G::LDS_BYTES);
        mas_attr_done(attr_mask, attr_bit);
    }
    p.tiles_h = mas_cdiv(p.Ho, THW); p.tiles_w = mas_cdiv(p.Wo, TWW);
    p.n_pt = p.N * p.tiles_h * p.tiles_w;
    p.n_co_t = mas_cdiv(p.Cout, BCO); p.n_ci_t = mas_cdiv(p.Cin, BCI);
    const int out_tiles = p.n_co_t * p.n_ci_t;
    int nsplit = mas_cdiv(mas_num_cus(), out_tiles);   // one (8- or 16-wave) work-group per CU
    if (nsplit > p.n_pt) nsplit = p.n_pt;
    if (nsplit < 1) nsplit = 1;
    p.nsplit = nsplit;
    if (splits_only) { *splits_only = nsplit; return MAS_OK; }
    hipLaunchKernelGGL(kern, dim3((unsigned)(out_tiles * nsplit)), dim3(WSplit<KS>::NTW), G::LDS_BYTES, s, p);
    MAS_CHECK_LAUNCH("conv_wgrad");
    return MAS_OK;
}

template <typename T>
int launch_t(const WgradParams& p, int ks, int stride, hipStream_t s, int* splits_only = nullptr) {
    if constexpr (sizeof(T) == 2) {
        if (stride == 1 && (p.Cout % 8) == 0 && (p.Cin % 8) == 0) {
            if (ks == 3) return launch_tr<3>(p, s, splits_only);
            if (ks == 1) return launch_tr<1>(p, s, splits_only);
            // discriminator geometries (reference losses/discriminator.py:20-36): 4x4 stride 1 directly (32-channel input
            // slices: two waves share a 32x32 tile and split the 16 taps 8 / 8); 4x4 stride 2 as the 2x2 stride-1
            // convolution of the space-to-depth input (mas_space_to_depth2x; the host wrapper un-permutes dW)
            if (ks == 4) return launch_tr<4, 32>(p, s, splits_only);
            if (ks == 2) return launch_tr<2, 64>(p, s, splits_only);
        }
    }
    if (ks == 1 && stride == 1) return launch<T, 1, 1, 8>(p, s, splits_only);
    if (ks == 3 && stride == 1) return launch<T, 3, 1, 8>(p, s, splits_only);
    if (ks == 3 && stride == 2) return launch<T, 3, 2, 4>(p, s, splits_only);
    if (splits_only) { *splits_only = 0; return MAS_OK; }
    MAS_FAIL(MAS_EUNSUPPORTED, "conv_wgrad: unsupported ks=%d stride=%d", ks, stride);
}

WgradParams general_params(const MasConvDesc* d, const void* x, const float* scale_shift, const void* dy, float* dw, float* dbias, int slabs) {
    WgradParams p;
    p.dbg = nullptr;
    p.x = x; p.ss = scale_shift; p.dy = dy; p.dw = dw; p.dbias = dbias;
    p.N = d->N; p.H = d->H; p.W = d->W; p.Cin = d->Cin; p.Ho = d->Ho; p.Wo = d->Wo; p.Cout = d->Cout;
    p.Hl = d->upsample ? 2 * d->H : d->H; p.Wl = d->upsample ? 2 * d->W : d->W;
    p.pad_top = d->pad_top; p.pad_left = d->pad_left; p.act = d->act; p.upsample = d->upsample;
    p.tiles_h = p.tiles_w = p.n_pt = p.n_co_t = p.n_ci_t = p.nsplit = 0;
    p.slabs = slabs;
    return p;
}

}  // namespace

// The general kernels as split-K SLABS (round 6): every shape without a kernel of its own -- the 159-channel edge convolutions of VQ-SEG, the
// discriminator's 4x4 convolutions, the exact-fp32 parity mode -- used to commit with fp32 atomics, the only order-dependent sums left in a
// training step.  mas_conv_wgrad_splits / mas_conv_wgrad_partial (conv_wgrad_dma.hip) fall through to these two.
int mas_conv_wgrad_general_splits(const MasConvDesc* d) {
    if (!d || d->Cin % 4 != 0 || (d->upsample && d->stride != 1)) return 0;       // (mas_wgrad_reduce reads the slabs as float4 along Cin)
    const WgradParams p = general_params(d, nullptr, nullptr, nullptr, nullptr, nullptr, 1);
    int k = 0;
    if (d->in_dtype == MAS_BF16) launch_t<bf16_t>(p, d->ks, d->stride, nullptr, &k);
    else if (d->in_dtype == MAS_F32) launch_t<float>(p, d->ks, d->stride, nullptr, &k);
    return k;
}

int mas_conv_wgrad_general_partial(const MasConvDesc* d, const void* x, const float* scale_shift, const void* dy, float* part, float* part_bias,
                                   hipStream_t s) {
    const WgradParams p = general_params(d, x, scale_shift, dy, part, part_bias, 1);
    if (d->in_dtype == MAS_BF16) return launch_t<bf16_t>(p, d->ks, d->stride, s);
    if (d->in_dtype == MAS_F32) return launch_t<float>(p, d->ks, d->stride, s);
    MAS_FAIL(MAS_EUNSUPPORTED, "conv_wgrad: unsupported dtype %d", d->in_dtype);
}

int mas_conv_wgrad_dma_try(const MasConvDesc* d, const void* x, const float* scale_shift, const void* dy, float* dw, float* dbias,
                           hipStream_t s);   // conv_wgrad_dma.hip

extern "C" int mas_conv_wgrad(const MasConvDesc* d, const void* x, const float* scale_shift, const void* dy,
                              float* dw, float* dbias, void* stream) {
    MAS_ENTER();
    if (!d || !x || !dy || !dw) MAS_FAIL(MAS_EINVAL, "conv_wgrad: null argument");
    if (d->act != MAS_ACT_NONE && !scale_shift) MAS_FAIL(MAS_EINVAL, "conv_wgrad: act prologue needs scale_shift");
    if (d->upsample && d->stride != 1) MAS_FAIL(MAS_EUNSUPPORTED, "conv_wgrad: upsample fold needs stride 1");
    {   // the FLOP-carrying shapes (3x3, stride 1, bf16, Cin % 64 == 0, Cout % 128 == 0) take the LDS-DMA kernel
        const int rc = mas_conv_wgrad_dma_try(d, x, scale_shift, dy, dw, dbias, reinterpret_cast<hipStream_t>(stream));
        if (rc != 0) return rc < 0 ? rc : MAS_OK;
    }
    WgradParams p = general_params(d, x, scale_shift, dy, dw, dbias, 0);
#ifdef MAS_TIMELINE          // s_memtime timeline builds only (tools/build_variant.sh tl -DMAS_TIMELINE)
    if (const char* e = getenv("MAS_DBG_PTR")) p.dbg = reinterpret_cast<unsigned long long*>(strtoull(e, nullptr, 0));
#endif
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (d->in_dtype == MAS_BF16) return launch_t<bf16_t>(p, d->ks, d->stride, s);
    if (d->in_dtype == MAS_F32) return launch_t<float>(p, d->ks, d->stride, s);
    MAS_FAIL(MAS_EUNSUPPORTED, "conv_wgrad: unsupported dtype %d", d->in_dtype);
}
