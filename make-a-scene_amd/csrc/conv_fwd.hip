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
// Weights: mas_pack_conv_weight emits the swizzled LDS image of every [tap][chunk][cout] row, so a
// weight tile is a linear, perfectly coalesced 16-byte copy (global -> registers one tap ahead ->
// ds_write_b128) into a double buffer; one barrier per tap.  All global loads are ordinary loads, so
// hipcc's counted s_waitcnt vmcnt(N) keeps the patch prefetch (issued one slot per tap, after that
// tap's weight loads) in flight across two taps and barriers.
#include "mas_common.h"

namespace {

struct ConvParams {
    const void* x; const float* ss; const void* w; const float* bias; const void* res; void* y;
    int N, H, W, Cin, Ho, Wo, Cout;
    int Hl, Wl;            // logical input size (2H,2W when upsample)
    int pad_top, pad_left, act, upsample;
    int n_chunks, Cout_pad; // packed weight dims
    int tiles_h, tiles_w, n_ct;
};

constexpr int TH = 8, TW = 16, NT = 256;

template <typename T, int KS, int STRIDE>
struct Geo {
    static constexpr int EPU = 16 / (int)sizeof(T);          // elements per 16-byte slot
    static constexpr int CK = 128 / (int)sizeof(T);          // channels per chunk (one 128-byte pixel row)
    static constexpr int PH = (TH - 1) * STRIDE + KS, PW = (TW - 1) * STRIDE + KS;
    static constexpr int PWL = (PW + 1) & ~1;                // even LDS pitch: slot parity == column parity
    static constexpr int PATCH_BYTES = PH * PWL * 128;
    static constexpr int P_UNITS = PH * PWL * 8;             // 16-byte slots in the patch
    static constexpr int NPU = (P_UNITS + NT - 1) / NT;      // slots per thread
    static constexpr int PB = NPU < 6 ? NPU : 6;             // slots per staging batch (registers)
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

template <typename T, typename TO, int KS, int STRIDE, int BC, int WC>
__global__ __launch_bounds__(NT, 2) void conv_fwd_kernel(ConvParams p) {
    using G = Geo<T, KS, STRIDE>;
    using V8 = typename Vec8<T>::type;
    constexpr int EPU = G::EPU, CK = G::CK, PW = G::PW, PWL = G::PWL, NPU = G::NPU, PB = G::PB;
    constexpr int WP = 4 / WC;                 // waves along pixels
    constexpr int MI = BC / WC / 32;           // 32-cout tiles per wave
    constexpr int NI = (TH * TW) / WP / 32;    // 32-pixel tiles per wave
    constexpr int WT_BYTES = BC * 128;         // one weight tile (BC rows x 128 B)
    constexpr int W_PER_T = BC * 8 / NT;       // 16-byte slots of a weight tile per thread
    constexpr int NTAP = KS * KS;
    constexpr bool PREFETCH = (NPU <= PB);     // whole chunk fits one register batch -> issue early / write late
    constexpr int PPT = (NPU + NTAP - 1) / NTAP;   // prefetch slots issued per tap

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* patch = smem;
    unsigned char* wbuf = smem + G::PATCH_BYTES;   // 2 buffers

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wave_c = wave % WC, wave_p = wave / WC;
    const int g = lane >> 5, l31 = lane & 31;

    int bid = blockIdx.x;
    const int ct = bid % p.n_ct; bid /= p.n_ct;
    const int tw_i = bid % p.tiles_w; bid /= p.tiles_w;
    const int th_i = bid % p.tiles_h; const int n = bid / p.tiles_h;
    const int c0 = ct * BC, h0 = th_i * TH, w0 = tw_i * TW;

    const T* __restrict__ X = reinterpret_cast<const T*>(p.x) + (size_t)n * p.H * p.W * p.Cin;
    const unsigned char* __restrict__ Wimg = reinterpret_cast<const unsigned char*>(p.w);

    f32x16 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    // ---- per-thread staging plan: slot s = tid&7 of patch pixel q = (tid>>3) + 32*i ---------------
    const int sl = tid & 7;
    int srcoff[NPU];                          // element offset of the pixel in this image, -1 = zero
    int dstoff[NPU];                          // byte offset in the LDS patch (swizzled), -1 = skip
#pragma unroll
    for (int i = 0; i < NPU; ++i) {
        const int q = (tid >> 3) + i * (NT / 8);
        const int pr = q / PWL, pc = q - pr * PWL;
        int ih = h0 * STRIDE + pr - p.pad_top, iw = w0 * STRIDE + pc - p.pad_left;
        const bool live = (q < G::PH * PWL) && (pc < PW);
        const bool inb = live && (ih >= 0) && (ih < p.Hl) && (iw >= 0) && (iw < p.Wl);
        if (p.upsample) { ih >>= 1; iw >>= 1; }
        srcoff[i] = inb ? (ih * p.W + iw) * p.Cin : -1;
        dstoff[i] = live ? q * 128 + ((sl ^ ((pc >> 1) & 7)) << 4) : -1;
    }
    const bool vec_in = (p.Cin % EPU) == 0;

    u32x4 preg[PB];
    auto p_issue_one = [&](int ci0, int b0, int k) {   // global -> register k of the batch starting at slot b0
        const int cb = ci0 + sl * EPU;
        const int i = b0 + k;
        u32x4 v = {0u, 0u, 0u, 0u};
        if (vec_in) {
            // unconditional load (clamped address; commit discards it for padding) keeps the per-wave
            // VMEM instruction count static, which hipcc's counted waits rely on
            const int so = (i < NPU && srcoff[i] >= 0 && cb < p.Cin) ? srcoff[i] + cb : 0;
            v = *reinterpret_cast<const u32x4*>(X + so);
        } else if (i < NPU && srcoff[i] >= 0 && cb < p.Cin) {
            const T* src = X + srcoff[i] + cb;
            T* tv = reinterpret_cast<T*>(&v);
#pragma unroll
            for (int e = 0; e < EPU; ++e) if (cb + e < p.Cin) tv[e] = src[e];
        }
        preg[k] = v;
    };
    auto p_issue = [&](int ci0, int b0) {
#pragma unroll
        for (int k = 0; k < PB; ++k) p_issue_one(ci0, b0, k);
    };
    auto p_commit = [&](int ci0, int b0) {    // registers -> (prologue) -> LDS
        const int cb = ci0 + sl * EPU;
        float sc[EPU], sh[EPU];
        if (p.act != MAS_ACT_NONE) {
#pragma unroll
            for (int e = 0; e < EPU; ++e) {
                const int c = cb + e;
                sc[e] = (c < p.Cin) ? p.ss[((size_t)n * p.Cin + c) * 2 + 0] : 0.0f;
                sh[e] = (c < p.Cin) ? p.ss[((size_t)n * p.Cin + c) * 2 + 1] : 0.0f;
            }
        }
#pragma unroll
        for (int k = 0; k < PB; ++k) {
            const int i = b0 + k;
            if (i >= NPU || dstoff[i] < 0) continue;
            u32x4 v = preg[k];
            if (srcoff[i] < 0 || cb >= p.Cin) v = u32x4{0u, 0u, 0u, 0u};   // zero padding / channels past Cin
            if (p.act != MAS_ACT_NONE && srcoff[i] >= 0) {     // padding stays exactly zero
                T* tv = reinterpret_cast<T*>(&v);
#pragma unroll
                for (int e = 0; e < EPU; ++e) {
                    float a = (float)tv[e] * sc[e] + sh[e];
                    if (p.act == MAS_ACT_AFFINE_SILU) a = silu_f(a);
                    tv[e] = (T)((cb + e < p.Cin) ? a : 0.0f);
                }
            }
            *reinterpret_cast<u32x4*>(patch + dstoff[i]) = v;
        }
    };
    // weight tile (tap, chunk): a linear 16-byte-per-thread copy of its pre-swizzled image
    u32x4 wreg[W_PER_T];
    auto w_issue = [&](int tap, int ch) {
        const unsigned char* src = Wimg + ((size_t)(tap * p.n_chunks + ch) * p.Cout_pad + c0) * 128 + tid * 16;
#pragma unroll
        for (int k = 0; k < W_PER_T; ++k) wreg[k] = *reinterpret_cast<const u32x4*>(src + k * NT * 16);
    };
    auto w_commit = [&](int buf) {
        unsigned char* dst = wbuf + buf * WT_BYTES + tid * 16;
#pragma unroll
        for (int k = 0; k < W_PER_T; ++k) *reinterpret_cast<u32x4*>(dst + k * NT * 16) = wreg[k];
    };

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

    w_issue(0, 0);
    if (PREFETCH) p_issue(0, 0);
    int wsel = 0;
    for (int ch = 0; ch < p.n_chunks; ++ch) {
        const int ci0 = ch * CK;
        if (ch > 0) __syncthreads();           // every wave is done reading the previous chunk's patch
        if (PREFETCH) p_commit(ci0, 0);
        else for (int b0 = 0; b0 < NPU; b0 += PB) { p_issue(ci0, b0); p_commit(ci0, b0); }
        const bool more = ch + 1 < p.n_chunks;
#pragma unroll
        for (int tap = 0; tap < NTAP; ++tap) {
            w_commit(wsel);
            __syncthreads();                   // weight tile `tap` (and, at tap 0, the patch) visible
            {
                int ntap = tap + 1, nch = ch;
                if (ntap == NTAP) { ntap = 0; nch = ch + 1; }
                if (nch < p.n_chunks) w_issue(ntap, nch);
            }
            if (PREFETCH && more) {            // next chunk's patch: PPT slots per tap, newer than this tap's weight loads
#pragma unroll
                for (int k = tap * PPT; k < (tap + 1) * PPT && k < NPU; ++k) p_issue_one(ci0 + CK, 0, k);
            }
            const int kh = tap / KS, kw = tap - kh * KS;
            const int hb = ((bcol + kw) >> 1) & 7;
            const unsigned char* wb = wbuf + wsel * WT_BYTES;
#pragma unroll
            for (int kk = 0; kk < CK / 16; ++kk) {   // channels past Cin are zero in both operands
                V8 bf[NI], af[MI];
#pragma unroll
                for (int j = 0; j < NI; ++j) bf[j] = ld_frag<T>(patch + (bq[j] + kh * PWL + kw) * 128, hb, kk, g);
#pragma unroll
                for (int i = 0; i < MI; ++i) af[i] = ld_frag<T>(wb + arow[i] * 128, (arow[i] >> 1) & 7, kk, g);
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NI; ++j) mma16(acc[i][j], af[i], bf[j]);
            }
            wsel ^= 1;
        }
    }

    // ---- epilogue: bias + residual, NHWC store ------------------------------------------------------
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
}

template <typename T, typename TO, int KS, int STRIDE, int BC, int WC>
int launch(const ConvParams& p, hipStream_t s) {
    using G = Geo<T, KS, STRIDE>;
    const size_t lds = (size_t)G::PATCH_BYTES + 2 * BC * 128;
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
    if ((long long)d->H * d->W * d->Cin > 0x7fffffffLL) MAS_FAIL(MAS_EUNSUPPORTED, "conv_fwd: one image exceeds 2^31 elements");
    ConvParams p;
    p.x = x; p.ss = scale_shift; p.w = w_packed; p.bias = bias; p.res = residual; p.y = y;
    p.N = d->N; p.H = d->H; p.W = d->W; p.Cin = d->Cin; p.Ho = d->Ho; p.Wo = d->Wo; p.Cout = d->Cout;
    p.Hl = d->upsample ? 2 * d->H : d->H; p.Wl = d->upsample ? 2 * d->W : d->W;
    p.pad_top = d->pad_top; p.pad_left = d->pad_left; p.act = d->act; p.upsample = d->upsample;
    const int ck = d->in_dtype == MAS_BF16 ? 64 : 32;
    p.n_chunks = mas_cdiv(d->Cin, ck); p.Cout_pad = mas_roundup(d->Cout, 128);
    p.tiles_h = mas_cdiv(d->Ho, TH); p.tiles_w = mas_cdiv(d->Wo, TW); p.n_ct = 1;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (d->in_dtype == MAS_BF16 && d->out_dtype == MAS_BF16) return launch_ks<bf16_t, bf16_t>(p, d->ks, d->stride, s);
    if (d->in_dtype == MAS_BF16 && d->out_dtype == MAS_F32) return launch_ks<bf16_t, float>(p, d->ks, d->stride, s);
    if (d->in_dtype == MAS_F32 && d->out_dtype == MAS_F32) return launch_ks<float, float>(p, d->ks, d->stride, s);
    MAS_FAIL(MAS_EUNSUPPORTED, "conv_fwd: unsupported dtype pair in=%d out=%d", d->in_dtype, d->out_dtype);
}
