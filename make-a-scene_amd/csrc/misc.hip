// Error plumbing, weight packing and the small NHWC helpers of libmas_hip.so.
#include "mas_common.h"
#include <stdarg.h>

static thread_local char g_err[512] = "";

void mas_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int mas_num_cus() {
    static std::atomic<int> cus[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;   // MI355X
    int c = cus[dev & 63].load(std::memory_order_relaxed);
    if (c == 0) {
        int n = 0;
        c = (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) ? n : 256;
        cus[dev & 63].store(c, std::memory_order_relaxed);
    }
    return c;
}

extern "C" const char* mas_last_error(void) { return g_err; }
static std::atomic<const char*> g_last_kernel{""};          // process-wide: the backward's launches come from autograd's own threads
void mas_note_kernel(const char* name) { g_last_kernel.store(name, std::memory_order_relaxed); }
extern "C" const char* mas_last_kernel(void) { return g_last_kernel.load(std::memory_order_relaxed); }
extern "C" int mas_abi_version(void) {
    MAS_ENTER(); return MAS_ABI_VERSION; }

namespace {
constexpr int NT = 256;

// OIHW fp32 -> the LDS image the conv kernels copy linearly: [chunk][tap][row][128 B] (consecutive tap-steps of a tile's
// K loop are consecutive in memory: the stream kernel addresses step t at t * rows_pad * 128), where a 128-byte row
// holds CK = 128/sizeof(T) consecutive K elements as eight 16-byte slots and slot position sp stores logical
// slot sp ^ ((row>>1)&7) (the bank-conflict swizzle of conv_fwd.hip).  See mas_hip.h for the two modes.
// One thread per 16-byte slot (8 bf16 / 4 fp32 consecutive K elements of one row and tap): 32-bit index arithmetic, two runtime
// divisions per slot, one 16-byte store.  (A thread per (chunk, row, slot position) looping over the taps reads the OIHW source in
// contiguous runs but leaves too few threads in flight: 2.5 ms instead of 0.6 ms for the 160 images of a VQ-IMG step.)
template <typename T>
__device__ __forceinline__ void pack_weight_body(const float* __restrict__ w, T* __restrict__ out, int Cout, int Cin, int ks, int transpose,
                                                 int rows_pad, int n_chunks, long long first, long long stride) {
    constexpr int EPU = 16 / (int)sizeof(T), CK = 128 / (int)sizeof(T);
    const int rows = transpose ? Cin : Cout, cols = transpose ? Cout : Cin;
    const unsigned kk = (unsigned)(ks * ks);
    const unsigned total = kk * (unsigned)n_chunks * (unsigned)rows_pad * 8u;       // slots
    for (unsigned u = (unsigned)first; u < total; u += (unsigned)stride) {
        const unsigned sp = u & 7u, r_ = u >> 3;
        const unsigned q = r_ / (unsigned)rows_pad, row = r_ - q * (unsigned)rows_pad;
        const unsigned ch = q / kk, t = q - ch * kk;
        const int kh = (int)t / ks, kw = (int)t - kh * ks;
        const int col0 = (int)ch * CK + (int)((sp ^ ((row >> 1) & 7u)) * EPU);
        u32x4 o;
        T* ov = reinterpret_cast<T*>(&o);
#pragma unroll
        for (int e = 0; e < EPU; ++e) {
            const int col = col0 + e;
            float v = 0.0f;
            if ((int)row < rows && col < cols) {
                if (!transpose) v = w[(((size_t)row * Cin + col) * ks + kh) * ks + kw];
                // rows = input channels (the dgrad's "Cout"), cols = output channels (its "Cin"), taps flipped
                else v = w[(((size_t)col * Cin + row) * ks + (ks - 1 - kh)) * ks + (ks - 1 - kw)];
            }
            ov[e] = (T)v;
        }
        *reinterpret_cast<u32x4*>(reinterpret_cast<unsigned char*>(out) + (size_t)u * 16) = o;
    }
}
template <typename T>
__global__ __launch_bounds__(NT) void pack_weight_kernel(const float* __restrict__ w, T* __restrict__ out, int Cout, int Cin,
                                                         int ks, int transpose, int rows_pad, int n_chunks) {
    pack_weight_body<T>(w, out, Cout, Cin, ks, transpose, rows_pad, n_chunks, (long long)blockIdx.x * NT + threadIdx.x, (long long)gridDim.x * NT);
}

// MAS_WLAYOUT_K32 (bf16): [chunk32][tap][row][64 B]; slot position sp stores logical slot sp ^ ((row>>2)&3); inside every
// 128-row tile the rows are PERMUTED: LDS row 32 i + l holds filter row 4 l + i (conv3x3_wide.hip: a lane then owns 4 consecutive
// output channels, one per accumulator tile, and stores them with one 8-byte store)
__device__ __forceinline__ void pack_weight_k32_body(const float* __restrict__ w, bf16_t* __restrict__ out, int Cout, int Cin, int ks, int transpose,
                                                     int rows_pad, int n_chunks, long long first, long long stride) {
    const int rows = transpose ? Cin : Cout, cols = transpose ? Cout : Cin;
    const unsigned kk = (unsigned)(ks * ks);
    const unsigned total = kk * (unsigned)n_chunks * (unsigned)rows_pad * 4u;       // 16-byte slots of the 64-byte rows
    for (unsigned u = (unsigned)first; u < total; u += (unsigned)stride) {
        const unsigned sp = u & 3u, r_ = u >> 2;
        const unsigned q = r_ / (unsigned)rows_pad, row = r_ - q * (unsigned)rows_pad;
        const unsigned ch = q / kk, t = q - ch * kk;
        const int kh = (int)t / ks, kw = (int)t - kh * ks;
        const int col0 = (int)ch * 32 + (int)((sp ^ ((row >> 2) & 3u)) * 8);
        const int frow = (int)((row & ~127u) + 4u * (row & 31u) + ((row & 127u) >> 5));      // filter row stored at LDS row `row`
        u32x4 o;
        bf16_t* ov = reinterpret_cast<bf16_t*>(&o);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int col = col0 + e;
            float v = 0.0f;
            if (frow < rows && col < cols) {
                if (!transpose) v = w[(((size_t)frow * Cin + col) * ks + kh) * ks + kw];
                else v = w[(((size_t)col * Cin + frow) * ks + (ks - 1 - kh)) * ks + (ks - 1 - kw)];
            }
            ov[e] = (bf16_t)v;
        }
        *reinterpret_cast<u32x4*>(reinterpret_cast<unsigned char*>(out) + (size_t)u * 16) = o;
    }
}
__global__ __launch_bounds__(NT) void pack_weight_k32_kernel(const float* __restrict__ w, bf16_t* __restrict__ out, int Cout, int Cin,
                                                             int ks, int transpose, int rows_pad, int n_chunks) {
    pack_weight_k32_body(w, out, Cout, Cin, ks, transpose, rows_pad, n_chunks, (long long)blockIdx.x * NT + threadIdx.x, (long long)gridDim.x * NT);
}

// MAS_WLAYOUT_UP2 (bf16, 3x3): the sub-pixel form of Upsample + conv (conv_up2.hip) -- [phase 2a+b][chunk32][tap 2r+s][row][64 B], rows
// and slots as K32; element = sum of the 3x3 taps (kh in R(a, r), kw in R(b, s)) in fp32, rounded once.  transpose = 1 (data
// gradient): in/out swapped and tap (1 - r, 1 - s) stored at (r, s).  R as a bit mask over {0,1,2}: index 2a + r.
__device__ __forceinline__ void pack_weight_up2_body(const float* __restrict__ w, bf16_t* __restrict__ out, int Cout, int Cin, int transpose,
                                                     int rows_pad, int n_chunks, long long first, long long stride) {
    const int rows = transpose ? Cin : Cout, cols = transpose ? Cout : Cin;
    const unsigned total = 16u * (unsigned)n_chunks * (unsigned)rows_pad * 4u;      // 16-byte slots
    for (unsigned u = (unsigned)first; u < total; u += (unsigned)stride) {
        const unsigned sp = u & 3u, r_ = u >> 2;
        const unsigned q = r_ / (unsigned)rows_pad, row = r_ - q * (unsigned)rows_pad;
        const unsigned t = q & 3u, qq = q >> 2;
        const unsigned ph = qq / (unsigned)n_chunks, ch = qq - ph * (unsigned)n_chunks;
        int tr = (int)(t >> 1), ts = (int)(t & 1u);
        if (transpose) { tr = 1 - tr; ts = 1 - ts; }
        const unsigned masks = 0x4361u;                                 // R(0,0) = 001b, R(0,1) = 110b, R(1,0) = 011b, R(1,1) = 100b (nibbles, index 2a + r)
        const unsigned mh = (masks >> (4 * (2 * (ph >> 1) + tr))) & 7u, mw = (masks >> (4 * (2 * (ph & 1u) + ts))) & 7u;
        const int col0 = (int)ch * 32 + (int)((sp ^ ((row >> 2) & 3u)) * 8);
        const int frow = (int)((row & ~127u) + 4u * (row & 31u) + ((row & 127u) >> 5));
        u32x4 o;
        bf16_t* ov = reinterpret_cast<bf16_t*>(&o);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int col = col0 + e;
            float v = 0.0f;
            if (frow < rows && col < cols) {
                const float* src = transpose ? w + ((size_t)col * Cin + frow) * 9 : w + ((size_t)frow * Cin + col) * 9;
#pragma unroll
                for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                    for (int kw = 0; kw < 3; ++kw)
                        if (((mh >> kh) & 1u) && ((mw >> kw) & 1u)) v += src[kh * 3 + kw];
            }
            ov[e] = (bf16_t)v;
        }
        *reinterpret_cast<u32x4*>(reinterpret_cast<unsigned char*>(out) + (size_t)u * 16) = o;
    }
}
__global__ __launch_bounds__(NT) void pack_weight_up2_kernel(const float* __restrict__ w, bf16_t* __restrict__ out, int Cout, int Cin,
                                                             int transpose, int rows_pad, int n_chunks) {
    pack_weight_up2_body(w, out, Cout, Cin, transpose, rows_pad, n_chunks, (long long)blockIdx.x * NT + threadIdx.x, (long long)gridDim.x * NT);
}

// Every stale packed weight of a step in ONE launch (the per-weight launches are ~8 us each of dependent-launch latency, ~160 per
// VQ-IMG step): work-group b looks its item up in the block-offset table (items sorted by first_block) and packs its share.
__global__ __launch_bounds__(NT) void pack_weight_batch_kernel(const MasPackItem* __restrict__ items, int n_items) {
    int lo = 0, hi = n_items - 1;
    const int b = blockIdx.x;
    while (lo < hi) {                                 // last item with first_block <= b
        const int mid = (lo + hi + 1) >> 1;
        if (items[mid].first_block <= b) lo = mid; else hi = mid - 1;
    }
    const MasPackItem it = items[lo];
    const long long first = (long long)(b - it.first_block) * NT + threadIdx.x, stride = (long long)it.n_blocks * NT;
    const int rows = it.transpose ? it.Cin : it.Cout, cols = it.transpose ? it.Cout : it.Cin;
    const int rows_pad = (rows + 127) / 128 * 128;
    auto cdiv = [](int a, int d) { return (a + d - 1) / d; };
    if (it.layout == MAS_WLAYOUT_UP2)
        pack_weight_up2_body(it.w_oihw, (bf16_t*)it.packed, it.Cout, it.Cin, it.transpose, rows_pad, cdiv(cols, 32), first, stride);
    else if (it.layout == MAS_WLAYOUT_K32)
        pack_weight_k32_body(it.w_oihw, (bf16_t*)it.packed, it.Cout, it.Cin, it.ks, it.transpose, rows_pad, cdiv(cols, 32), first, stride);
    else if (it.dtype == MAS_BF16)
        pack_weight_body<bf16_t>(it.w_oihw, (bf16_t*)it.packed, it.Cout, it.Cin, it.ks, it.transpose, rows_pad, cdiv(cols, 64), first, stride);
    else
        pack_weight_body<float>(it.w_oihw, (float*)it.packed, it.Cout, it.Cin, it.ks, it.transpose, rows_pad, cdiv(cols, 32), first, stride);
}

// ---- every bf16 image of a parameter from ONE read of it, tile-transposed through LDS (mas_pack_conv_weight_tiles) ------------------
// The gather kernels above produce a 16-byte slot from 8 reads at a 36-byte stride (3x3), once per image: 0.65 ms for the ~160 images
// of a VQ-IMG step.  Here a work-group owns a 64 (cout) x 64 (cin) x taps tile of the OIHW parameter: its 64 rows are contiguous runs
// of 64 * taps floats (coalesced 16-byte loads), staged once in LDS as bf16, and every image the parameter has -- forward and data-
// gradient operand, K64 or K32 layout -- is written from there with whole 16-byte slots in the image's own order (1 KiB+ runs).  The
// tile grid covers the PADDED extents (rows up to the next multiple of 128 of either image), so the zero padding the convolution
// kernels rely on is rewritten too.  Same bytes as the gather kernels (tests/test_gpu_pack_tiles.py: bit for bit).
__device__ __forceinline__ unsigned pk2(const bf16_t* l, int a, int b) {   // two LDS bf16 -> one dword
    return (unsigned)reinterpret_cast<const unsigned short*>(l)[a] | ((unsigned)reinterpret_cast<const unsigned short*>(l)[b] << 16);
}
__global__ __launch_bounds__(NT) void pack_weight_tiles_kernel(const MasPackTileItem* __restrict__ items, int n_items) {
    extern __shared__ __attribute__((aligned(16))) unsigned char tsm[];
    bf16_t* L = reinterpret_cast<bf16_t*>(tsm);                    // [64 o][64 i][kk]  (a straight copy of the 64 source runs)
    int lo = 0, hi = n_items - 1;
    const int b = blockIdx.x, tid = threadIdx.x;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (items[mid].first_block <= b) lo = mid; else hi = mid - 1;
    }
    const MasPackTileItem it = items[lo];
    const int Cout = it.Cout, Cin = it.Cin, kk = it.ks * it.ks;
    const int tiles_i = (Cin + 127) / 128 * 2;
    const int tb = b - it.first_block, o0 = (tb / tiles_i) * 64, i0 = (tb % tiles_i) * 64;
    const int run = 64 * kk;                                       // floats of one row's 64-channel run
    // ---- stage: row ol = the run W[o0 + ol][i0 .. i0 + 63][all taps]; zero outside the parameter
    const bool full = i0 + 64 <= Cin && ((size_t)Cin * kk) % 4 == 0 && (reinterpret_cast<uintptr_t>(it.w_oihw) & 15) == 0;
    if (full) {
        const int v4 = run / 4;                                    // 16 * kk float4 per row
        for (int u = tid; u < 64 * v4; u += NT) {
            const int ol = u / v4, q = u - ol * v4;
            f32x4 v = {0.0f, 0.0f, 0.0f, 0.0f};
            if (o0 + ol < Cout) v = *reinterpret_cast<const f32x4*>(it.w_oihw + ((size_t)(o0 + ol) * Cin + i0) * kk + 4 * q);
            const bf16x4 o = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
            *reinterpret_cast<bf16x4*>(L + (size_t)ol * run + 4 * q) = o;
        }
    } else {
        const int valid = i0 < Cin ? (Cin - i0 < 64 ? Cin - i0 : 64) * kk : 0;
        for (int u = tid; u < 64 * run; u += NT) {
            const int ol = u / run, f = u - ol * run;
            float v = 0.0f;
            if (o0 + ol < Cout && f < valid) v = it.w_oihw[((size_t)(o0 + ol) * Cin + i0) * kk + f];
            L[u] = (bf16_t)v;
        }
    }
    __syncthreads();
    // ---- emit every image.  A slot = 8 consecutive columns of one (row, tap): forward image rows = couts, columns = cins;
    //      data-gradient image rows = cins, columns = couts, taps flipped (see mas_pack_conv_weight)
    for (int im = 0; im < it.n_img; ++im) {
        const int tr = it.transpose[im], k32 = it.layout[im] == MAS_WLAYOUT_K32;
        const int rows = tr ? Cin : Cout, cols = tr ? Cout : Cin;
        const int r0 = tr ? i0 : o0, c0 = tr ? o0 : i0;            // this tile's rows / columns in the image's terms
        const int rows_pad = (rows + 127) / 128 * 128;
        if (r0 >= rows_pad) continue;
        unsigned char* out = reinterpret_cast<unsigned char*>(it.img[im]);
        if (!k32) {
            const int ch = c0 / 64;
            if (ch >= (cols + 63) / 64) continue;
            for (int u = tid; u < kk * 512; u += NT) {             // (tap, row, slot position): 8 KiB contiguous per tap
                const int t = u >> 9, row_l = (u >> 3) & 63, sp = u & 7;
                const int row = r0 + row_l, ls = sp ^ ((row >> 1) & 7);
                const int ts = tr ? kk - 1 - t : t;
                u32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int ca = ls * 8 + 2 * e, cb = ca + 1;    // column inside the tile
                    const int ia = tr ? (ca * 64 + row_l) * kk + ts : (row_l * 64 + ca) * kk + ts;
                    const int ib = tr ? (cb * 64 + row_l) * kk + ts : (row_l * 64 + cb) * kk + ts;
                    o[e] = pk2(L, ia, ib);
                }
                *reinterpret_cast<u32x4*>(out + ((size_t)(ch * kk + t) * rows_pad + row) * 128 + sp * 16) = o;
            }
        } else {
            // K32: 64-byte rows; inside every 128-row block LDS row 32 i + l holds filter row 4 l + i; this tile holds filter rows
            // [r0, r0 + 64) = l in [16 half, 16 half + 16), i = 0..3: four runs of 16 rows (1 KiB each) per (chunk, tap)
            const int half = (r0 >> 6) & 1, base = r0 & ~127;
            for (int h = 0; h < 2; ++h) {
                const int ch = c0 / 32 + h;
                if (ch >= (cols + 31) / 32) continue;
                for (int u = tid; u < kk * 256; u += NT) {         // (tap, i, l, slot position)
                    const int t = u >> 8, i_ = (u >> 6) & 3, l16 = (u >> 2) & 15, sp = u & 3;
                    const int row = base + 32 * i_ + l16 + 16 * half;          // LDS row of the image
                    const int row_l = 4 * l16 + i_;                             // filter row inside the tile (0..63)
                    const int ls = sp ^ ((row >> 2) & 3);
                    const int ts = tr ? kk - 1 - t : t;
                    u32x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int ca = h * 32 + ls * 8 + 2 * e, cb = ca + 1;
                        const int ia = tr ? (ca * 64 + row_l) * kk + ts : (row_l * 64 + ca) * kk + ts;
                        const int ib = tr ? (cb * 64 + row_l) * kk + ts : (row_l * 64 + cb) * kk + ts;
                        o[e] = pk2(L, ia, ib);
                    }
                    *reinterpret_cast<u32x4*>(out + ((size_t)(ch * kk + t) * rows_pad + row) * 64 + sp * 16) = o;
                }
            }
        }
    }
}

template <typename T>
__global__ __launch_bounds__(NT) void upsample2x_kernel(const T* __restrict__ x, T* __restrict__ y, int N, int H, int W, int C) {
    // one thread per 16-byte unit of the OUTPUT
    constexpr int EPU = 16 / (int)sizeof(T);
    const int upp = C / EPU;
    const long long total = (long long)N * 2 * H * 2 * W * upp;
    for (long long i = (long long)blockIdx.x * NT + threadIdx.x; i < total; i += (long long)gridDim.x * NT) {
        const int cu = (int)(i % upp); long long r = i / upp;
        const int wo = (int)(r % (2 * W)); r /= 2 * W;
        const int ho = (int)(r % (2 * H)); const int n = (int)(r / (2 * H));
        const u32x4 v = *reinterpret_cast<const u32x4*>(x + (((size_t)n * H + (ho >> 1)) * W + (wo >> 1)) * C + cu * EPU);
        *reinterpret_cast<u32x4*>(y + (size_t)i * EPU) = v;
    }
}

template <typename T>
__global__ __launch_bounds__(NT) void sumpool2x_kernel(const T* __restrict__ x, T* __restrict__ y, int N, int Ho, int Wo, int C) {
    constexpr int EPU = 16 / (int)sizeof(T);
    const int upp = C / EPU;
    const long long total = (long long)N * Ho * Wo * upp;
    const int Hi = 2 * Ho, Wi = 2 * Wo;
    for (long long i = (long long)blockIdx.x * NT + threadIdx.x; i < total; i += (long long)gridDim.x * NT) {
        const int cu = (int)(i % upp); long long r = i / upp;
        const int wo = (int)(r % Wo); r /= Wo;
        const int ho = (int)(r % Ho); const int n = (int)(r / Ho);
        float acc[EPU];
#pragma unroll
        for (int e = 0; e < EPU; ++e) acc[e] = 0.0f;
#pragma unroll
        for (int dh = 0; dh < 2; ++dh)
#pragma unroll
            for (int dw = 0; dw < 2; ++dw) {
                const u32x4 v = *reinterpret_cast<const u32x4*>(x + (((size_t)n * Hi + 2 * ho + dh) * Wi + 2 * wo + dw) * C + cu * EPU);
                const T* ev = reinterpret_cast<const T*>(&v);
#pragma unroll
                for (int e = 0; e < EPU; ++e) acc[e] += (float)ev[e];
            }
        u32x4 ov; T* o = reinterpret_cast<T*>(&ov);
#pragma unroll
        for (int e = 0; e < EPU; ++e) o[e] = (T)acc[e];
        *reinterpret_cast<u32x4*>(y + (size_t)i * EPU) = ov;
    }
}

// y[n][2h][2w] = x[n][h][w], everything else 0; y is [N,Hout,Wout,C]
template <typename T>
__global__ __launch_bounds__(NT) void zero_stuff2x_kernel(const T* __restrict__ x, T* __restrict__ y, int N, int H, int W, int C,
                                                          int Hout, int Wout) {
    constexpr int EPU = 16 / (int)sizeof(T);
    const int upp = C / EPU;
    const long long total = (long long)N * Hout * Wout * upp;
    for (long long i = (long long)blockIdx.x * NT + threadIdx.x; i < total; i += (long long)gridDim.x * NT) {
        const int cu = (int)(i % upp); long long r = i / upp;
        const int wo = (int)(r % Wout); r /= Wout;
        const int ho = (int)(r % Hout); const int n = (int)(r / Hout);
        u32x4 v = {0u, 0u, 0u, 0u};
        if (!(ho & 1) && !(wo & 1) && (ho >> 1) < H && (wo >> 1) < W)
            v = *reinterpret_cast<const u32x4*>(x + (((size_t)n * H + (ho >> 1)) * W + (wo >> 1)) * C + cu * EPU);
        *reinterpret_cast<u32x4*>(y + (size_t)i * EPU) = v;
    }
}

// y[n][h'][w'][(dy*2+dx)*C + c] = x[n][2h'+dy-pad][2w'+dx-pad][c]  (0 outside): a stride-2 KxK convolution of x is the stride-1
// (K/2)x(K/2) convolution of y -- used for the WEIGHT gradient of the discriminator's 4x4 stride-2 convolutions
// (reference losses/discriminator.py:20,27), which then runs on the stride-1 transpose-read wgrad kernel with ks = 2
template <typename T>
__global__ __launch_bounds__(NT) void space_to_depth2x_kernel(const T* __restrict__ x, T* __restrict__ y, int N, int H, int W, int C,
                                                              int Ho, int Wo, int pad) {
    constexpr int EPU = 16 / (int)sizeof(T);
    const int upp = C / EPU;
    const long long total = (long long)N * Ho * Wo * 4 * upp;
    for (long long i = (long long)blockIdx.x * NT + threadIdx.x; i < total; i += (long long)gridDim.x * NT) {
        const int cu = (int)(i % upp); long long r = i / upp;
        const int q = (int)(r % 4); r /= 4;
        const int wo = (int)(r % Wo); r /= Wo;
        const int ho = (int)(r % Ho); const int n = (int)(r / Ho);
        const int ih = 2 * ho + (q >> 1) - pad, iw = 2 * wo + (q & 1) - pad;
        u32x4 v = {0u, 0u, 0u, 0u};
        if (ih >= 0 && ih < H && iw >= 0 && iw < W) v = *reinterpret_cast<const u32x4*>(x + (((size_t)n * H + ih) * W + iw) * C + cu * EPU);
        *reinterpret_cast<u32x4*>(y + (size_t)i * EPU) = v;
    }
}

int grid_for(long long total) {
    long long b = (total + NT - 1) / NT;
    if (b > 8192) b = 8192;
    return b < 1 ? 1 : (int)b;
}
}  // namespace

extern "C" size_t mas_packed_weight_elems(int Cout, int Cin, int ks) {
    // large enough for either packing mode and either dtype (fp32 chunks are 32 channels, bf16 64)
    const size_t a = (size_t)mas_roundup(Cout, 128) * mas_roundup(Cin, 64);
    const size_t b = (size_t)mas_roundup(Cin, 128) * mas_roundup(Cout, 64);
    return (size_t)ks * ks * (a > b ? a : b);
}

extern "C" size_t mas_packed_weight_elems_up2(int Cout, int Cin) {   // 4 phases x 4 taps of the MAS_WLAYOUT_UP2 image, either operand
    const size_t a = (size_t)mas_roundup(Cout, 128) * mas_roundup(Cin, 32);
    const size_t b = (size_t)mas_roundup(Cin, 128) * mas_roundup(Cout, 32);
    return 16 * (a > b ? a : b);
}

extern "C" int mas_pack_conv_weight(const float* w_oihw, void* packed, int Cout, int Cin, int ks, int transpose, int dtype,
                                    void* stream) {
    MAS_ENTER();
    if (!w_oihw || !packed) MAS_FAIL(MAS_EINVAL, "pack_conv_weight: null argument");
    if (Cout <= 0 || Cin <= 0 || (ks < 1 || ks > 4)) MAS_FAIL(MAS_EINVAL, "pack_conv_weight: bad shape");
    const int rows = transpose ? Cin : Cout, cols = transpose ? Cout : Cin;
    const int rows_pad = mas_roundup(rows, 128);
    const int ck = dtype == MAS_BF16 ? 64 : 32;
    const int n_chunks = mas_cdiv(cols, ck);
    const long long total = (long long)ks * ks * n_chunks * rows_pad * ck;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (dtype == MAS_BF16)
        hipLaunchKernelGGL(pack_weight_kernel<bf16_t>, dim3(grid_for(total)), dim3(NT), 0, s, w_oihw, (bf16_t*)packed, Cout, Cin, ks, transpose, rows_pad, n_chunks);
    else if (dtype == MAS_F32)
        hipLaunchKernelGGL(pack_weight_kernel<float>, dim3(grid_for(total)), dim3(NT), 0, s, w_oihw, (float*)packed, Cout, Cin, ks, transpose, rows_pad, n_chunks);
    else MAS_FAIL(MAS_EUNSUPPORTED, "pack_conv_weight: dtype %d", dtype);
    MAS_CHECK_LAUNCH("pack_conv_weight");
    return MAS_OK;
}

extern "C" int mas_pack_conv_weight_layout(const float* w_oihw, void* packed, int Cout, int Cin, int ks, int transpose, int dtype,
                                           int layout, void* stream) {
    if (layout == MAS_WLAYOUT_K64) return mas_pack_conv_weight(w_oihw, packed, Cout, Cin, ks, transpose, dtype, stream);
    MAS_ENTER();
    if (!w_oihw || !packed) MAS_FAIL(MAS_EINVAL, "pack_conv_weight: null argument");
    if (layout == MAS_WLAYOUT_UP2) {
        if (dtype != MAS_BF16 || ks != 3 || Cout <= 0 || Cin <= 0) MAS_FAIL(MAS_EUNSUPPORTED, "pack_conv_weight: the UP2 image is bf16 / 3x3 only");
        const int rows = transpose ? Cin : Cout, cols = transpose ? Cout : Cin;
        const int rows_pad = mas_roundup(rows, 128), n_chunks = mas_cdiv(cols, 32);
        const long long total = 16LL * n_chunks * rows_pad * 4;
        hipLaunchKernelGGL(pack_weight_up2_kernel, dim3(grid_for(total)), dim3(NT), 0, reinterpret_cast<hipStream_t>(stream), w_oihw,
                           (bf16_t*)packed, Cout, Cin, transpose, rows_pad, n_chunks);
        MAS_CHECK_LAUNCH("pack_conv_weight_up2");
        return MAS_OK;
    }
    if (layout != MAS_WLAYOUT_K32) MAS_FAIL(MAS_EINVAL, "pack_conv_weight: bad layout %d", layout);
    if (dtype != MAS_BF16 || ks != 3 || Cout <= 0 || Cin <= 0) MAS_FAIL(MAS_EUNSUPPORTED, "pack_conv_weight: the K32 image is bf16 / 3x3 only");
    const int rows = transpose ? Cin : Cout, cols = transpose ? Cout : Cin;
    const int rows_pad = mas_roundup(rows, 128), n_chunks = mas_cdiv(cols, 32);
    const long long total = 9LL * n_chunks * rows_pad * 32;
    hipLaunchKernelGGL(pack_weight_k32_kernel, dim3(grid_for(total)), dim3(NT), 0, reinterpret_cast<hipStream_t>(stream), w_oihw,
                       (bf16_t*)packed, Cout, Cin, ks, transpose, rows_pad, n_chunks);
    MAS_CHECK_LAUNCH("pack_conv_weight_k32");
    return MAS_OK;
}

extern "C" int mas_pack_batch_blocks(int Cout, int Cin, int ks, int transpose, int dtype, int layout) {
    if (Cout <= 0 || Cin <= 0 || ks < 1 || ks > 4) return 0;
    const int rows = transpose ? Cin : Cout, cols = transpose ? Cout : Cin;
    const int ck = layout == MAS_WLAYOUT_K64 ? (dtype == MAS_BF16 ? 64 : 32) : 32;
    if (layout == MAS_WLAYOUT_UP2 && (ks != 3 || dtype != MAS_BF16)) return 0;
    const long long total = (long long)(layout == MAS_WLAYOUT_UP2 ? 16 : ks * ks) * mas_cdiv(cols, ck) * mas_roundup(rows, 128) * (layout == MAS_WLAYOUT_K64 ? 8 : 4);
    long long b = (total + NT - 1) / NT;                               // one 16-byte slot per thread
    if (b > 512) b = 512;
    return b < 1 ? 1 : (int)b;
}

extern "C" int mas_pack_conv_weight_batch(const MasPackItem* items_device, int n_items, int total_blocks, void* stream) {
    MAS_ENTER();
    if (!items_device || n_items <= 0 || total_blocks <= 0) MAS_FAIL(MAS_EINVAL, "pack_conv_weight_batch: empty batch");
    hipLaunchKernelGGL(pack_weight_batch_kernel, dim3((unsigned)total_blocks), dim3(NT), 0, reinterpret_cast<hipStream_t>(stream), items_device, n_items);
    MAS_CHECK_LAUNCH("pack_conv_weight_batch");
    return MAS_OK;
}

extern "C" int mas_pack_tile_blocks(int Cout, int Cin, int ks) {
    if (Cout <= 0 || Cin <= 0 || ks < 1 || ks > 4) return 0;
    return ((Cout + 127) / 128 * 2) * ((Cin + 127) / 128 * 2);
}

extern "C" int mas_pack_conv_weight_tiles(const MasPackTileItem* items_device, int n_items, int total_blocks, int max_ks, void* stream) {
    MAS_ENTER();
    if (!items_device || n_items <= 0 || total_blocks <= 0 || max_ks < 1 || max_ks > 4) MAS_FAIL(MAS_EINVAL, "pack_conv_weight_tiles: bad batch");
    const size_t lds = (size_t)64 * 64 * max_ks * max_ks * 2;
    static mas_devmask_t attr{0};
    unsigned long long bit;
    if (mas_attr_needed(attr, &bit)) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(pack_weight_tiles_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 64 * 16 * 2) != hipSuccess)
            MAS_FAIL(MAS_ELAUNCH, "pack_conv_weight_tiles: cannot reserve LDS");
        mas_attr_done(attr, bit);
    }
    hipLaunchKernelGGL(pack_weight_tiles_kernel, dim3((unsigned)total_blocks), dim3(NT), lds, reinterpret_cast<hipStream_t>(stream), items_device, n_items);
    MAS_CHECK_LAUNCH("pack_conv_weight_tiles");
    return MAS_OK;
}

// ---- weight-gradient commit: acc [Cout][ks][ks][Cin] fp32 (the split-K accumulator mas_conv_wgrad's kernels add into, followed by
// [Cout] bias sums when dbias != NULL) -> dw in the parameter's own OIHW layout, db; the accumulator is ZEROED on the way out, so the
// caller keeps ONE persistent scratch per stream instead of a fresh zero-filled tensor (a fill launch) and a permute copy per call
__global__ __launch_bounds__(256) void wgrad_commit_kernel(float* __restrict__ acc, float* __restrict__ dw, float* __restrict__ db,
                                                           int Cout, int Cin, int kk, long long npair) {
    // one thread per (cout, cin) pair: its kk taps are read at stride Cin (consecutive threads = consecutive cin: coalesced) and
    // written as kk consecutive floats (a wave covers 64 * kk contiguous floats of the OIHW tensor)
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx < npair) {
        const int i = (int)(idx % Cin), o = (int)(idx / Cin);
        float* src = acc + (long long)o * kk * Cin + i;
        float* dst = dw + idx * kk;
        for (int t = 0; t < kk; ++t) { dst[t] = src[(long long)t * Cin]; src[(long long)t * Cin] = 0.0f; }
    } else if (db && idx < npair + Cout) {
        float* src = acc + npair * kk + (idx - npair);
        db[idx - npair] = *src;
        *src = 0.0f;
    }
}

extern "C" int mas_wgrad_commit(float* acc, float* dw_oihw, float* dbias, int Cout, int Cin, int ks, void* stream) {
    MAS_ENTER();
    if (!acc || !dw_oihw || Cout <= 0 || Cin <= 0 || ks < 1) MAS_FAIL(MAS_EINVAL, "wgrad_commit: bad argument");
    const long long npair = (long long)Cout * Cin, tot = npair + (dbias ? Cout : 0);
    hipLaunchKernelGGL(wgrad_commit_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       acc, dw_oihw, dbias, Cout, Cin, ks * ks, npair);
    MAS_CHECK_LAUNCH("wgrad_commit");
    return MAS_OK;
}

// ---- fixed-order reduction of split-K weight-gradient partials (mas_conv_wgrad_partial): part [nsplit][Cout][kk][Cin] -> dw OIHW,
// part_bias [nsplit][Cout] -> db.  A block owns (one output channel) x (64 input channels) x (all taps): thread (el, g) sums slabs
// g, g + 4, g + 8, ... of its input channel for each tap (coalesced 256-byte rows, every load independent), the four group sums are
// added in order -- the same association every run, whatever the arrival order of the work-groups that wrote the slabs -- and the
// 64 x kk results leave as ONE contiguous run of the OIHW tensor.  The last cdiv(Cout, 64) blocks do the bias the same way.
// G = 16 (many slabs, few output channels: 128 -> 128 channels has 128 slabs and only 256 (o, 64 ci) blocks): 16 slab groups per block,
// blockIdx.y picks the tap set {y, y + 4, y + 8, y + 12}.
template <int G>
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ part, const float* __restrict__ part_bias, int nsplit,
                                                           float* __restrict__ dw, float* __restrict__ db, int Cout, int Cin, int kk,
                                                           int ci_chunks) {
    __shared__ float red[G][64 / G][64];                            // [slab group][tap slot][ci]
    const int el = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int nwb = Cout * ci_chunks;
    if ((int)blockIdx.x >= nwb) {                                   // bias (blockIdx.y == 0 only)
        if (blockIdx.y) return;
        const int o = ((int)blockIdx.x - nwb) * 64 + el;
        float acc = 0.0f;
        if (o < Cout)
            for (int sp = grp; sp < nsplit; sp += 4) acc += part_bias[(long long)sp * Cout + o];
        red[grp][0][el] = acc;
        __syncthreads();
        if (grp == 0 && o < Cout) db[o] = ((red[0][0][el] + red[1][0][el]) + red[2][0][el]) + red[3][0][el];
        return;
    }
    const int o = (int)blockIdx.x / ci_chunks, i0 = ((int)blockIdx.x % ci_chunks) * 64;
    const long long slab = (long long)Cout * kk * Cin;
    // 16 float4 lanes span the 64 input channels; the other 16 ways = G slab groups x 16 / G tap sets {ts, ts + 4, ts + 8, ts + 12}: up
    // to 8 independent 16-byte loads in flight per thread (a scalar version left the latency exposed: 3.5 TB/s)
    const int q = threadIdx.x & 15, r = threadIdx.x >> 4;
    const int g = G == 4 ? (r & 3) : r, ts = G == 4 ? (r >> 2) : (int)blockIdx.y;
    const int i = i0 + q * 4;
    if (i < Cin) {
        f32x4 acc[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[k] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        const float* src = part + ((long long)o * kk + ts) * Cin + i;
#pragma unroll 2
        for (int sp = g; sp < nsplit; sp += G) {
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (ts + 4 * k < kk) acc[k] += *reinterpret_cast<const f32x4*>(src + (long long)sp * slab + (long long)(4 * k) * Cin);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (ts + 4 * k < kk) *reinterpret_cast<f32x4*>(&red[g][G == 4 ? ts + 4 * k : k][q * 4]) = acc[k];
    }
    __syncthreads();
    const int ni = Cin - i0 < 64 ? Cin - i0 : 64;
    float* dst = dw + ((long long)o * Cin + i0) * kk;               // [i][t]
    if (G == 4) {                                                   // all taps here: one contiguous run
        for (int j = threadIdx.x; j < ni * kk; j += 256) {
            const int il = j / kk, t = j - il * kk;
            dst[j] = ((red[0][t][il] + red[1][t][il]) + red[2][t][il]) + red[3][t][il];
        }
    } else {
        const int nk = (kk - ts + 3) / 4;                           // taps of this block's set
        for (int j = threadIdx.x; j < ni * nk; j += 256) {
            const int il = j / nk, k = j - il * nk;
            float v = 0.0f;
#pragma unroll
            for (int gg = 0; gg < G; ++gg) v += red[gg][k][il];
            dst[il * kk + ts + 4 * k] = v;
        }
    }
}

// ---- Upsample + conv, sub-pixel form (conv_up2.hip, conv_wgrad_dma.hip KS = 2): part [4 phases][nsplit][Cout][2x2][Cin] -> dw OIHW 3x3.
// dWp[a][b][r][s] (the fixed-order sum of the phase's slabs) is the gradient of the phase weight Wp[a][b][r][s] = sum of W[kh][kw] over
// kh in R(a, r), kw in R(b, s); so dW[kh][kw] = sum over (a, b) of dWp[a][b][r(a, kh)][s(b, kw)], r(0, .) = {0, 1, 1}, r(1, .) = {0, 0, 1}.
// A block owns (one output channel) x (64 input channels): thread (q, g) sums slabs g, g + 8, ... of its four input channels for all
// 16 phase taps, the eight group sums are added in order (bitwise reproducible), the 64 x 9 results leave as one contiguous run.
__global__ __launch_bounds__(128) void wgrad_reduce_up2_kernel(const float* __restrict__ part, const float* __restrict__ part_bias, int nsplit,
                                                               float* __restrict__ dw, float* __restrict__ db, int Cout, int Cin, int ci_chunks) {
    __shared__ float red[8][16][64];                                // [slab group][phase tap][ci]: 32 KiB
    const int nwb = Cout * ci_chunks;
    if ((int)blockIdx.x >= nwb) {                                   // bias: part_bias [4 * nsplit][Cout]
        const int o = ((int)blockIdx.x - nwb) * 128 + (int)threadIdx.x;
        if (o < Cout) {
            float acc = 0.0f;
            for (int sp = 0; sp < 4 * nsplit; ++sp) acc += part_bias[(long long)sp * Cout + o];
            db[o] = acc;
        }
        return;
    }
    const int o = (int)blockIdx.x / ci_chunks, i0 = ((int)blockIdx.x % ci_chunks) * 64;
    const int q = threadIdx.x & 15, g = threadIdx.x >> 4;            // 16 float4 lanes x 8 slab groups
    const int i = i0 + q * 4;
    const long long slab = (long long)Cout * 4 * Cin;
    if (i < Cin) {
        f32x4 acc[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) acc[k] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        for (int sp = g; sp < nsplit; sp += 8) {
#pragma unroll
            for (int ph = 0; ph < 4; ++ph) {
                const float* src = part + ((long long)ph * nsplit + sp) * slab + ((long long)o * 4) * Cin + i;
#pragma unroll
                for (int t = 0; t < 4; ++t) acc[ph * 4 + t] += *reinterpret_cast<const f32x4*>(src + (long long)t * Cin);
            }
        }
#pragma unroll
        for (int k = 0; k < 16; ++k) *reinterpret_cast<f32x4*>(&red[g][k][q * 4]) = acc[k];
    }
    __syncthreads();
    const int ni = Cin - i0 < 64 ? Cin - i0 : 64;
    float* dst = dw + ((long long)o * Cin + i0) * 9;                 // [i][kh][kw]
    for (int j = threadIdx.x; j < ni * 9; j += 128) {
        const int il = j / 9, t = j - il * 9, kh = t / 3, kw = t - kh * 3;
        float v = 0.0f;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const int r = a == 0 ? (kh > 0) : (kh > 1), s_ = b == 0 ? (kw > 0) : (kw > 1);
                const int k = (2 * a + b) * 4 + 2 * r + s_;
                float u = 0.0f;
#pragma unroll
                for (int gg = 0; gg < 8; ++gg) u += red[gg][k][il];
                v += u;
            }
        dst[j] = v;
    }
}

extern "C" int mas_wgrad_reduce_up2(const float* part, const float* part_bias, int nsplit, float* dw_oihw, float* dbias, int Cout, int Cin,
                                    void* stream) {
    MAS_ENTER();
    if (!part || !dw_oihw || nsplit <= 0 || Cout <= 0 || Cin <= 0) MAS_FAIL(MAS_EINVAL, "wgrad_reduce_up2: bad argument");
    if (dbias && !part_bias) MAS_FAIL(MAS_EINVAL, "wgrad_reduce_up2: dbias without part_bias");
    if (Cin % 4 != 0 || (reinterpret_cast<uintptr_t>(part) & 15)) MAS_FAIL(MAS_EINVAL, "wgrad_reduce_up2: Cin % 4 == 0 and a 16-byte aligned table required");
    const int ci_chunks = (Cin + 63) / 64;
    const long long blocks = (long long)Cout * ci_chunks + (dbias ? (Cout + 127) / 128 : 0);
    hipLaunchKernelGGL(wgrad_reduce_up2_kernel, dim3((unsigned)blocks), dim3(128), 0, reinterpret_cast<hipStream_t>(stream), part, part_bias, nsplit,
                       dw_oihw, dbias, Cout, Cin, ci_chunks);
    MAS_CHECK_LAUNCH("wgrad_reduce_up2");
    return MAS_OK;
}

extern "C" int mas_wgrad_reduce(const float* part, const float* part_bias, int nsplit, float* dw_oihw, float* dbias, int Cout, int Cin,
                                int ks, void* stream) {
    MAS_ENTER();
    if (!part || !dw_oihw || nsplit <= 0 || Cout <= 0 || Cin <= 0 || ks < 1 || ks > 4) MAS_FAIL(MAS_EINVAL, "wgrad_reduce: bad argument");
    if (dbias && !part_bias) MAS_FAIL(MAS_EINVAL, "wgrad_reduce: dbias without part_bias");
    if (Cin % 4 != 0 || (reinterpret_cast<uintptr_t>(part) & 15)) MAS_FAIL(MAS_EINVAL, "wgrad_reduce: Cin % 4 == 0 and a 16-byte aligned table required");
    const int ci_chunks = (Cin + 63) / 64;
    const long long blocks = (long long)Cout * ci_chunks + (dbias ? (Cout + 63) / 64 : 0);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (nsplit >= 32 && ks > 1)                                     // a function of the shape and the CU budget only: the association is fixed
        hipLaunchKernelGGL(wgrad_reduce_kernel<16>, dim3((unsigned)blocks, ks * ks < 4 ? ks * ks : 4), dim3(256), 0, s, part, part_bias,
                           nsplit, dw_oihw, dbias, Cout, Cin, ks * ks, ci_chunks);
    else
        hipLaunchKernelGGL(wgrad_reduce_kernel<4>, dim3((unsigned)blocks), dim3(256), 0, s, part, part_bias, nsplit, dw_oihw, dbias, Cout,
                           Cin, ks * ks, ci_chunks);
    MAS_CHECK_LAUNCH("wgrad_reduce");
    return MAS_OK;
}

#define MAS_DISPATCH_NHWC(NAME, KERN, TOTAL, ...)                                                                   \
    do {                                                                                                            \
        const int epu_ = dtype == MAS_BF16 ? 8 : 4;                                                                 \
        if (C % epu_) MAS_FAIL(MAS_EUNSUPPORTED, NAME ": C=%d must be a multiple of %d", C, epu_);                  \
        hipStream_t s_ = reinterpret_cast<hipStream_t>(stream);                                                     \
        const long long total_ = (TOTAL) / epu_;                                                                    \
        if (dtype == MAS_BF16) hipLaunchKernelGGL(KERN<bf16_t>, dim3(grid_for(total_)), dim3(NT), 0, s_, (const bf16_t*)x, (bf16_t*)y, __VA_ARGS__); \
        else if (dtype == MAS_F32) hipLaunchKernelGGL(KERN<float>, dim3(grid_for(total_)), dim3(NT), 0, s_, (const float*)x, (float*)y, __VA_ARGS__); \
        else MAS_FAIL(MAS_EUNSUPPORTED, NAME ": dtype %d", dtype);                                                  \
        MAS_CHECK_LAUNCH(NAME);                                                                                     \
        return MAS_OK;                                                                                              \
    } while (0)

extern "C" int mas_upsample2x(const void* x, void* y, int dtype, int N, int H, int W, int C, void* stream) {
    MAS_ENTER();
    if (!x || !y) MAS_FAIL(MAS_EINVAL, "upsample2x: null argument");
    MAS_DISPATCH_NHWC("upsample2x", upsample2x_kernel, (long long)N * 4 * H * W * C, N, H, W, C);
}
extern "C" int mas_sumpool2x(const void* x, void* y, int dtype, int N, int Ho, int Wo, int C, void* stream) {
    MAS_ENTER();
    if (!x || !y) MAS_FAIL(MAS_EINVAL, "sumpool2x: null argument");
    MAS_DISPATCH_NHWC("sumpool2x", sumpool2x_kernel, (long long)N * Ho * Wo * C, N, Ho, Wo, C);
}
extern "C" int mas_space_to_depth2x(const void* x, void* y, int dtype, int N, int H, int W, int C, int Ho, int Wo, int pad, void* stream) {
    MAS_ENTER();
    if (!x || !y) MAS_FAIL(MAS_EINVAL, "space_to_depth2x: null argument");
    MAS_DISPATCH_NHWC("space_to_depth2x", space_to_depth2x_kernel, (long long)N * Ho * Wo * 4 * C, N, H, W, C, Ho, Wo, pad);
}
extern "C" int mas_zero_stuff2x(const void* x, void* y, int dtype, int N, int H, int W, int C, int Hout, int Wout, void* stream) {
    MAS_ENTER();
    if (!x || !y) MAS_FAIL(MAS_EINVAL, "zero_stuff2x: null argument");
    MAS_DISPATCH_NHWC("zero_stuff2x", zero_stuff2x_kernel, (long long)N * Hout * Wout * C, N, H, W, C, Hout, Wout);
}
