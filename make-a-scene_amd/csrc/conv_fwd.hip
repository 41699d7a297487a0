// Implicit-GEMM convolution forward for gfx950 (also the stride-1 data gradient).
//
// Replaces F.conv2d at reference models/modules.py:49,68,93,100,113,145-160,219,236,345,364
// and models/vqvae.py:15,18, with the GroupNorm-apply + SiLU of modules.py:121-128 fused into
// the input loader and bias / residual add (modules.py:136,191) into the epilogue.
//
// GEMM view:  Y^T[cout][pixel] = sum_{tap,ci} Wp[tap][cout][ci] * A[pixel (+) tap][ci]
//   M = cout  (MFMA A operand = packed weights, K-contiguous rows)
//   N = pixel (MFMA B operand = activations; lane owns one pixel -> 4 consecutive couts per
//              accumulator quad -> 8/16-byte NHWC stores)
//   K = (tap, ci)
// Work-group: 256 threads = 4 waves; output tile = 8x16 pixels x BC couts.
// LDS im2col: the (8*s+ks-1) x (16*s+ks-1) input halo patch for a CK-channel chunk is loaded
// ONCE (coalesced 16-byte NHWC loads, prologue applied once per element, zero padding applied
// after the activation) and all ks*ks taps read their shifted B fragments out of it.
// Weights stream through a double-buffered LDS tile, prefetched through registers one tap
// ahead, so the steady state has one barrier per tap.
#include "mas_common.h"

namespace {

struct ConvParams {
    const void* x; const float* ss; const void* w; const float* bias; const void* res; void* y;
    int N, H, W, Cin, Ho, Wo, Cout;
    int Hl, Wl;            // logical input size (2H,2W when upsample)
    int pad_top, pad_left, act, upsample;
    int Cin_pad, Cout_pad; // packed weight dims
    int tiles_h, tiles_w, n_ct;
};

constexpr int TH = 8, TW = 16, NT = 256;

template <typename T, int KS, int STRIDE>
struct Geo {
    static constexpr int EPU = 16 / (int)sizeof(T);          // elements per 16-byte unit
    static constexpr int CK = 128 / (int)sizeof(T);          // channels per chunk (128 B per pixel)
    static constexpr int PH = (TH - 1) * STRIDE + KS, PW = (TW - 1) * STRIDE + KS;
    static constexpr int PSTR = CK + EPU;                    // padded pixel stride (elements): 144 B
    static constexpr int PATCH_ELEMS = PH * PW * PSTR;
};

template <typename T, typename TO, int KS, int STRIDE, int BC, int WC>
__global__ __launch_bounds__(NT) void conv_fwd_kernel(ConvParams p) {
    using G = Geo<T, KS, STRIDE>;
    using V8 = typename Vec8<T>::type;
    constexpr int EPU = G::EPU, CK = G::CK, PW = G::PW, PSTR = G::PSTR;
    constexpr int WP = 4 / WC;                 // waves along pixels
    constexpr int MI = BC / WC / 32;           // 32-cout tiles per wave
    constexpr int NI = (TH * TW) / WP / 32;    // 32-pixel tiles per wave
    constexpr int WSTR = CK + EPU;             // weight tile row stride (elements)
    constexpr int WT_ELEMS = BC * WSTR;
    constexpr int W_UNITS = BC * 8;            // 16-byte units per weight tile (8 per 128-B row)
    constexpr int W_PER_T = (W_UNITS + NT - 1) / NT;
    constexpr int P_UNITS = G::PH * G::PW * 8;
    constexpr int NTAP = KS * KS;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    T* patch = reinterpret_cast<T*>(smem);
    T* wbuf = patch + G::PATCH_ELEMS;          // 2 buffers

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wave_c = wave % WC, wave_p = wave / WC;
    const int g = lane >> 5, l31 = lane & 31;

    int bid = blockIdx.x;
    const int ct = bid % p.n_ct; bid /= p.n_ct;
    const int tw_i = bid % p.tiles_w; bid /= p.tiles_w;
    const int th_i = bid % p.tiles_h; const int n = bid / p.tiles_h;
    const int c0 = ct * BC, h0 = th_i * TH, w0 = tw_i * TW;

    const T* __restrict__ X = reinterpret_cast<const T*>(p.x);
    const T* __restrict__ Wp = reinterpret_cast<const T*>(p.w);

    f32x16 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    // per-lane LDS element offsets of the B (pixel) and A (cout) fragments
    int boff[NI];
#pragma unroll
    for (int j = 0; j < NI; ++j) {
        const int pix = (wave_p * NI + j) * 32 + l31;
        boff[j] = ((pix >> 4) * STRIDE * PW + (pix & 15) * STRIDE) * PSTR + g * 8;
    }
    int aoff[MI];
#pragma unroll
    for (int i = 0; i < MI; ++i) aoff[i] = ((wave_c * MI + i) * 32 + l31) * WSTR + g * 8;

    const int n_chunks = (p.Cin_pad + CK - 1) / CK;
    const bool vec_in = (p.Cin % EPU) == 0;
    const int cu = tid & 7;                    // this thread's 16-byte unit inside a 128-B pixel row (fixed)

    // ---- weight tile prefetch into registers ------------------------------------
    u32x4 wreg[W_PER_T];
    auto w_issue = [&](int tap, int ci0) {
#pragma unroll
        for (int i = 0; i < W_PER_T; ++i) {
            const int u = tid + i * NT;
            const int row = u >> 3;
            u32x4 v = {0u, 0u, 0u, 0u};
            if (u < W_UNITS && (c0 + row) < p.Cout_pad && (ci0 + cu * EPU) < p.Cin_pad) {
                const T* src = Wp + ((size_t)(tap * p.Cout_pad + c0 + row) * p.Cin_pad + ci0 + cu * EPU);
                v = *reinterpret_cast<const u32x4*>(src);
            }
            wreg[i] = v;
        }
    };
    auto w_commit = [&](int buf) {
#pragma unroll
        for (int i = 0; i < W_PER_T; ++i) {
            const int u = tid + i * NT;
            if (u < W_UNITS) {
                T* dst = wbuf + buf * WT_ELEMS + (u >> 3) * WSTR + cu * EPU;
                *reinterpret_cast<u32x4*>(dst) = wreg[i];
            }
        }
    };

    w_issue(0, 0);
    int wsel = 0;
    for (int ch = 0; ch < n_chunks; ++ch) {
        const int ci0 = ch * CK;
        __syncthreads();                       // all waves done with the previous chunk's patch
        // ---- stage the halo patch for channels [ci0, ci0+CK) ----------------------
        {
            float sc[EPU], sh[EPU];
            const int cb = ci0 + cu * EPU;
            if (p.act != MAS_ACT_NONE) {
#pragma unroll
                for (int e = 0; e < EPU; ++e) {
                    const int c = cb + e;
                    sc[e] = (c < p.Cin) ? p.ss[((size_t)n * p.Cin + c) * 2 + 0] : 0.0f;
                    sh[e] = (c < p.Cin) ? p.ss[((size_t)n * p.Cin + c) * 2 + 1] : 0.0f;
                }
            }
            for (int u = tid; u < P_UNITS; u += NT) {
                const int pp = u >> 3;
                const int pr = pp / PW, pc = pp - pr * PW;
                int ih = h0 * STRIDE + pr - p.pad_top, iw = w0 * STRIDE + pc - p.pad_left;
                const bool inb = (ih >= 0) && (ih < p.Hl) && (iw >= 0) && (iw < p.Wl);
                if (p.upsample) { ih >>= 1; iw >>= 1; }
                float v[EPU];
#pragma unroll
                for (int e = 0; e < EPU; ++e) v[e] = 0.0f;
                if (inb && cb < p.Cin) {
                    const T* src = X + ((size_t)(n * p.H + ih) * p.W + iw) * p.Cin + cb;
                    if (vec_in) {
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
                u32x4 outv;
                T* ov = reinterpret_cast<T*>(&outv);
#pragma unroll
                for (int e = 0; e < EPU; ++e) ov[e] = (T)v[e];
                *reinterpret_cast<u32x4*>(patch + pp * PSTR + cu * EPU) = outv;
            }
        }
        const int kk_n = min(CK, p.Cin_pad - ci0) / 16;     // Cin_pad is a multiple of 16
        for (int tap = 0; tap < NTAP; ++tap) {
            w_commit(wsel);
            __syncthreads();                   // weight tile (and, at tap 0, the patch) visible
            {   // prefetch the next weight tile while this one is consumed
                int ntap = tap + 1, nci0 = ci0;
                if (ntap == NTAP) { ntap = 0; nci0 = ci0 + CK; }
                if (nci0 < p.Cin_pad) w_issue(ntap, nci0);
            }
            const int kh = tap / KS, kw = tap - kh * KS;
            const T* pb = patch + (kh * PW + kw) * PSTR;
            const T* wb = wbuf + wsel * WT_ELEMS;
            for (int kk = 0; kk < kk_n; ++kk) {
                V8 bf[NI], af[MI];
#pragma unroll
                for (int j = 0; j < NI; ++j) bf[j] = ld8<T>(pb + boff[j] + kk * 16);
#pragma unroll
                for (int i = 0; i < MI; ++i) af[i] = ld8<T>(wb + aoff[i] + kk * 16);
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NI; ++j) mma16(acc[i][j], af[i], bf[j]);
            }
            wsel ^= 1;
        }
    }

    // ---- epilogue: bias + residual, NHWC store ------------------------------------
    TO* __restrict__ Y = reinterpret_cast<TO*>(p.y);
    const T* __restrict__ R = reinterpret_cast<const T*>(p.res);
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
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] += (float)R[obase + co + e];
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
}

template <typename T, typename TO, int KS, int STRIDE, int BC, int WC>
int launch(const ConvParams& p, hipStream_t s) {
    using G = Geo<T, KS, STRIDE>;
    constexpr int WSTR = G::CK + G::EPU;
    const size_t lds = (size_t)(G::PATCH_ELEMS + 2 * BC * WSTR) * sizeof(T);
    auto kern = conv_fwd_kernel<T, TO, KS, STRIDE, BC, WC>;
    static bool attr_done = false;
    if (!attr_done) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            MAS_FAIL(MAS_ELAUNCH, "conv_fwd: cannot set dynamic LDS size %zu", lds);
        attr_done = true;
    }
    ConvParams q = p;
    q.n_ct = mas_cdiv(p.Cout, BC);
    const long long blocks = (long long)p.N * p.tiles_h * p.tiles_w * q.n_ct;
    if (blocks <= 0 || blocks > 0x7fffffffLL) MAS_FAIL(MAS_EINVAL, "conv_fwd: bad grid %lld", blocks);
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(NT), lds, s, q);
    MAS_CHECK_LAUNCH("conv_fwd");
    return MAS_OK;
}

template <typename T, typename TO, int KS, int STRIDE>
int launch_bc(const ConvParams& p, hipStream_t s) {
    if (p.Cout <= 32) return launch<T, TO, KS, STRIDE, 32, 1>(p, s);
    if (p.Cout <= 64) return launch<T, TO, KS, STRIDE, 64, 1>(p, s);
    return launch<T, TO, KS, STRIDE, 128, 2>(p, s);
}

template <typename T, typename TO>
int launch_ks(const ConvParams& p, int ks, int stride, hipStream_t s) {
    if (ks == 1 && stride == 1) return launch_bc<T, TO, 1, 1>(p, s);
    if (ks == 3 && stride == 1) return launch_bc<T, TO, 3, 1>(p, s);
    if (ks == 3 && stride == 2) return launch_bc<T, TO, 3, 2>(p, s);
    MAS_FAIL(MAS_EUNSUPPORTED, "conv_fwd: unsupported ks=%d stride=%d", ks, stride);
}

}  // namespace

extern "C" int mas_conv_fwd(const MasConvDesc* d, const void* x, const float* scale_shift, const void* w_packed,
                            const float* bias, const void* residual, void* y, void* stream) {
    MAS_ENTER();
    if (!d || !x || !w_packed || !y) MAS_FAIL(MAS_EINVAL, "conv_fwd: null argument");
    if (d->act != MAS_ACT_NONE && !scale_shift) MAS_FAIL(MAS_EINVAL, "conv_fwd: act prologue needs scale_shift");
    if (d->N <= 0 || d->H <= 0 || d->W <= 0 || d->Cin <= 0 || d->Cout <= 0 || d->Ho <= 0 || d->Wo <= 0)
        MAS_FAIL(MAS_EINVAL, "conv_fwd: non-positive dimension");
    if (d->upsample && d->stride != 1) MAS_FAIL(MAS_EUNSUPPORTED, "conv_fwd: upsample fold needs stride 1");
    ConvParams p;
    p.x = x; p.ss = scale_shift; p.w = w_packed; p.bias = bias; p.res = residual; p.y = y;
    p.N = d->N; p.H = d->H; p.W = d->W; p.Cin = d->Cin; p.Ho = d->Ho; p.Wo = d->Wo; p.Cout = d->Cout;
    p.Hl = d->upsample ? 2 * d->H : d->H; p.Wl = d->upsample ? 2 * d->W : d->W;
    p.pad_top = d->pad_top; p.pad_left = d->pad_left; p.act = d->act; p.upsample = d->upsample;
    p.Cin_pad = mas_roundup(d->Cin, 16); p.Cout_pad = mas_roundup(d->Cout, 32);
    p.tiles_h = mas_cdiv(d->Ho, TH); p.tiles_w = mas_cdiv(d->Wo, TW); p.n_ct = 1;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (d->in_dtype == MAS_BF16 && d->out_dtype == MAS_BF16) return launch_ks<bf16_t, bf16_t>(p, d->ks, d->stride, s);
    if (d->in_dtype == MAS_BF16 && d->out_dtype == MAS_F32) return launch_ks<bf16_t, float>(p, d->ks, d->stride, s);
    if (d->in_dtype == MAS_F32 && d->out_dtype == MAS_F32) return launch_ks<float, float>(p, d->ks, d->stride, s);
    MAS_FAIL(MAS_EUNSUPPORTED, "conv_fwd: unsupported dtype pair in=%d out=%d", d->in_dtype, d->out_dtype);
}
