// 1x1 / stride-1 / bf16 convolution = the plain GEMM  y[pixel][cout] = sum_ci a[pixel][ci] * w[cout][ci] + bias (+ residual)  on NHWC rows
// (reference models/modules.py:106-108 `nin_shortcut`, :145-160 AttnBlock q / k / v / proj_out, and the data gradients of the same sites).
//
// Why its own kernel.  conv_fwd.hip stages a halo patch per 64-channel chunk and then runs the filter taps over it: nine taps of MFMA
// work per staging for a 3x3 filter, ONE for a 1x1 -- the same barriers, waits and register-staged copies for a ninth of the arithmetic
// (profiles/r03_conv_shapes.txt: 512 -> 1536 @16^2 52 us = 247 TFLOP/s, the data gradient 37 us, all 1x1 launches 1.7 ms of a 60 ms step).
// Without a halo there is nothing to stage but two plain tiles, so this is a textbook LDS-tiled GEMM in the idiom of the other kernels:
//   * tile = 128 pixels x 128 couts, K chunks of 64 channels (128-byte rows); both operands go global -> LDS by DMA
//     (`buffer_load_dwordx4 ... lds`, inline assembly: no staging registers, and no compiler-inserted vmcnt(0) in front of the LDS reads),
//     double-buffered, ONE work-group barrier per chunk: chunk c + 1 is requested right after the barrier that publishes chunk c;
//   * the weights come straight from the K64 image conv_fwd.hip uses (mas_pack_conv_weight_layout: [chunk][tap][cout][128 B], 16-byte slot
//     ^ ((row >> 1) & 7)); the pixel rows get the same XOR on the SOURCE address of the DMA, so every ds_read_b128 is conflict-free;
//   * 4 waves, wave tile 64 couts x 64 pixels (4 accumulator tiles of v_mfma_f32_32x32x16_bf16: 1 fragment read per MFMA);
//   * epilogue as in conv3x3_stream.hip: lanes l / l + 32 swap accumulator quads so that a lane owns 8 consecutive couts of its pixel:
//     16-byte bias / residual loads and stores.
// LDS 64 KiB, <= 128 VGPRs: two work-groups per CU.
#include "mas_common.h"
#include <type_traits>

namespace {

typedef __attribute__((ext_vector_type(4))) int pw_i32x4;
__device__ __forceinline__ void pw_dma16(pw_i32x4 rs, unsigned lds, int vo) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" :: "s"(lds), "v"(vo), "s"(rs) : "memory", "m0");
}
__device__ __forceinline__ pw_i32x4 pw_rsrc(const void* base, unsigned bytes) {
    const unsigned long long a = (unsigned long long)base;
    pw_i32x4 r = {(int)(unsigned)a, (int)(unsigned)(a >> 32), (int)bytes, 0x00020000};
    r[0] = __builtin_amdgcn_readfirstlane(r[0]); r[1] = __builtin_amdgcn_readfirstlane(r[1]);
    r[2] = __builtin_amdgcn_readfirstlane(r[2]); r[3] = __builtin_amdgcn_readfirstlane(r[3]);
    return r;
}

struct PwParams {
    const unsigned char* a; const unsigned char* w; const float* bias; const unsigned char* res; unsigned char* y;
    int M, K, N;                               // pixels, input channels, output channels
    int rows_pad, n_chunks, n_nt;              // weight image: rows per chunk (Cout rounded up to 128), 64-channel chunks; cout tiles
    int xcd_bands;                             // grid % 8 == 0 and more than one cout tile: the banded block -> tile map (see the kernel)
};

constexpr int PW_NT = 256;
constexpr int PW_TILE = 128 * 128;             // 128 rows x 128 B
constexpr int PW_STAGE = 2 * PW_TILE;          // pixel rows, then weight rows
constexpr int PW_LDS = 2 * PW_STAGE;
constexpr int PW_OOB = (int)0x80000000;

__global__ __launch_bounds__(PW_NT, 2) void conv1x1_kernel(PwParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char pw_smem[];
    const unsigned lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) unsigned char*)pw_smem;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 5, l31 = lane & 31;
    const int wave_n = wave & 1, wave_m = wave >> 1;               // 2 x 64 couts, 2 x 64 pixels
    // cout tile fastest: the work-groups that share a pixel tile are adjacent in the tile list -- and, since round 6, on the SAME XCD: work-group b
    // runs on XCD b % 8 (observed dispatch order; speed only), so with the plain b -> tile map the n_nt work-groups of a pixel tile sit on
    // n_nt different XCDs and its rows cross the fabric once per XCD.  Banded map: XCD x takes the contiguous eighth [x G / 8, (x + 1) G / 8).
    int tile = (int)blockIdx.x;
    if (p.xcd_bands) tile = (tile & 7) * ((int)gridDim.x >> 3) + (tile >> 3);
    const int nt = tile % p.n_nt, mt = tile / p.n_nt;
    const int m0 = mt * 128, n0 = nt * 128;

    const pw_i32x4 rs_a = pw_rsrc(p.a, (unsigned)((size_t)p.M * p.K * 2));
    const pw_i32x4 rs_w = pw_rsrc(p.w, (unsigned)((size_t)p.n_chunks * p.rows_pad * 128));

    // ---- DMA plan: a tile is 16 pieces of 8 rows x 128 B; wave w moves pieces 4 w .. 4 w + 3 of both tiles.  LDS image lane-linear
    //      (row 8 piece + (lane >> 3), physical slot lane & 7); the pixel rows carry the swizzle on the SOURCE slot, the weight image
    //      is stored swizzled already.
    int va[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int row = (wave * 4 + j) * 8 + (lane >> 3), ps = lane & 7;
        va[j] = (m0 + row < p.M) ? (m0 + row) * p.K * 2 + ((ps ^ ((row >> 1) & 7)) << 4) : PW_OOB;
    }
    const int vw = n0 * 128 + wave * 4096 + lane * 16;
    auto issue = [&](int c, int stage) {
        const int wc = c * p.rows_pad * 128;                        // uniform
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const unsigned dst = lds0 + stage * PW_STAGE + (wave * 4 + j) * 1024;
            pw_dma16(rs_a, __builtin_amdgcn_readfirstlane(dst), va[j] + c * 128);          // (PW_OOB + c * 128 stays out of range)
            pw_dma16(rs_w, __builtin_amdgcn_readfirstlane(dst + PW_TILE), vw + wc + j * 1024);
        }
    };

    // ---- fragment addresses (lane parts; stage and k-step are immediates / XOR constants)
    int aoff[2], boff[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = wave_n * 64 + i * 32 + l31;                 // cout row of the weight tile
        aoff[i] = PW_TILE + row * 128 + ((g ^ ((row >> 1) & 7)) << 4);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int row = wave_m * 64 + j * 32 + l31;                 // pixel row
        boff[j] = row * 128 + ((g ^ ((row >> 1) & 7)) << 4);
    }

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    issue(0, 0);
    auto chunk = [&](int c, auto stage_c) {
        constexpr int ST = decltype(stage_c)::value;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // this wave's pieces of chunk c have landed ...
        __builtin_amdgcn_s_barrier();                                // ... everybody's have, and everybody is done reading chunk c - 1
        asm volatile("" ::: "memory");
        if (c + 1 < p.n_chunks) issue(c + 1, ST ^ 1);
        const unsigned char* sb = pw_smem + ST * PW_STAGE;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            bf16x8 afr[2], bfr[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) afr[i] = *reinterpret_cast<const bf16x8*>(sb + (aoff[i] ^ (kk << 5)));
#pragma unroll
            for (int j = 0; j < 2; ++j) bfr[j] = *reinterpret_cast<const bf16x8*>(sb + (boff[j] ^ (kk << 5)));
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) mma16(acc[i][j], afr[i], bfr[j]);               // D[cout][pixel]: registers = couts, lane = pixel
        }
    };
    for (int c = 0; c < p.n_chunks; c += 2) {
        chunk(c, std::integral_constant<int, 0>{});
        if (c + 1 < p.n_chunks) chunk(c + 1, std::integral_constant<int, 1>{});
    }

    // ---- epilogue (bias and residual by UNCONDITIONAL buffer loads: a null pointer is a zero-length descriptor that returns zeros)
    const unsigned out_bytes = (unsigned)((size_t)p.M * p.N * 2);
    const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, out_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_r = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(p.res ? p.res : p.y), 0, p.res ? out_bytes : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc(p.bias ? (void*)p.bias : (void*)p.y, 0, p.bias ? (unsigned)(p.N * 4) : 0u, 0x00020000);
    f32x4 bv[2][2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int qp = 0; qp < 2; ++qp) {
            const int cb = (n0 + wave_n * 64 + i * 32 + 16 * qp + 8 * g) * 4;
            bv[i][qp][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_b, cb, 0, 0));
            bv[i][qp][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_b, cb + 16, 0, 0));
        }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int pix = m0 + wave_m * 64 + j * 32 + l31;
        const int obase = pix < p.M ? (pix * p.N + n0 + wave_n * 64 + 8 * g) * 2 : PW_OOB;   // the stores add (i * 32 + qp * 16) * 2
        u32x4 rv[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int qp = 0; qp < 2; ++qp) rv[i][qp] = __builtin_amdgcn_raw_buffer_load_b128(rs_r, obase + (i * 32 + qp * 16) * 2, 0, 0);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int qp = 0; qp < 2; ++qp) {
                float v[8];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float qa = acc[i][j][(2 * qp) * 4 + e], qb = acc[i][j][(2 * qp + 1) * 4 + e];
                    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(qa), __float_as_uint(qb), false, false);
                    v[e] = __uint_as_float(r[0]); v[4 + e] = __uint_as_float(r[1]);
                }
                const bf16_t* rb = reinterpret_cast<const bf16_t*>(&rv[i][qp]);
                u32x4 o;
                bf16_t* ob = reinterpret_cast<bf16_t*>(&o);
#pragma unroll
                for (int e = 0; e < 8; ++e) ob[e] = (bf16_t)(v[e] + bv[i][qp][e >> 2][e & 3] + (float)rb[e]);
                __builtin_amdgcn_raw_buffer_store_b128(o, rs_y, obase + (i * 32 + qp * 16) * 2, 0, 0);
            }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// weight gradient of the same layers:  dW[cout][cin] = sum over pixels of dy[pixel][cout] * a[pixel][cin]  -- a GEMM whose reduction
// dimension is the OUTER dimension of both NHWC operands.  Both are staged in their natural [pixel][channel] layout (64 pixels x 128
// channels = 64 rows of 256 B per chunk, by LDS-DMA) and the MFMA fragments come from the LDS transpose read (ds_read_b64_tr_b16;
// semantics in conv_wgrad.hip), 64-byte blocks XOR-swizzled with (pixel & 3) as in conv_wgrad_dma.hip.  Split-K over the pixel chunks
// into partial slabs [nsplit][Cout][Cin] that mas_wgrad_reduce adds in a fixed order (no atomics: bitwise reproducible).  Tile 128 couts x
// 128 cins, 4 waves of 64 x 64; the conv_wgrad_tr_kernel<1, 64> instance it replaces spent 66 us on 512 -> 1536 @16^2 (194 TFLOP/s).
struct PwWgradParams {
    const unsigned char* dy; const unsigned char* x; float* part; float* part_bias;   // part_bias [nsplit][Cout] or NULL
    int M, Cout, Cin, n_co_t, n_ci_t, nsplit, n_chunks;
};

__global__ __launch_bounds__(PW_NT, 2) void wgrad1x1_kernel(PwWgradParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char pw_smem[];
    const unsigned lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) unsigned char*)pw_smem;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 5, l31 = lane & 31, G16 = (lane >> 4) & 1, sl = lane & 15;
    const int wave_co = wave & 1, wave_ci = wave >> 1;
    int bid = blockIdx.x;
    const int split = bid % p.nsplit; bid /= p.nsplit;
    const int ci_t = bid % p.n_ci_t, co_t = bid / p.n_ci_t;
    const int co0 = co_t * 128, ci0 = ci_t * 128;

    const pw_i32x4 rs_dy = pw_rsrc(p.dy, (unsigned)((size_t)p.M * p.Cout * 2));
    const pw_i32x4 rs_x = pw_rsrc(p.x, (unsigned)((size_t)p.M * p.Cin * 2));

    // ---- DMA plan: a chunk tile = 16 pieces of 4 pixel rows x 256 B; wave w moves pieces 4 w .. 4 w + 3 of both tensors.  Lane: pixel
    //      lane >> 4 of the piece, physical 64-byte block (lane >> 2) & 3 holding LOGICAL block ^ (pixel & 3), 16-byte slot lane & 3
    const int lp = lane >> 4;
    const int lsrc = ((((lane >> 2) & 3) ^ lp) << 6) + ((lane & 3) << 4);
    auto issue = [&](int c, int stage) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int pix = c * 64 + (wave * 4 + j) * 4 + lp;
            const bool ok = pix < p.M;
            const unsigned dst = lds0 + stage * PW_STAGE + (wave * 4 + j) * 1024;
            pw_dma16(rs_dy, __builtin_amdgcn_readfirstlane(dst), ok ? (pix * p.Cout + co0) * 2 + lsrc : PW_OOB);
            pw_dma16(rs_x, __builtin_amdgcn_readfirstlane(dst + PW_TILE), ok ? (pix * p.Cin + ci0) * 2 + lsrc : PW_OOB);
        }
    };
    // ---- transpose-read lane addressing: pixel 8 g + (sl >> 2) (+ 4 for the second read) of a 16-pixel k-step, channels
    //      16 G16 + 4 (sl & 3) ..+3 of a 32-channel group; the group's 64-byte block ^ (pixel & 3)
    const int t4 = sl >> 2;
    int a_off[2], b_off[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        a_off[i] = (8 * g + t4) * 256 + (((wave_co * 2 + i) ^ t4) << 6) + 32 * G16 + 8 * (sl & 3);
        b_off[i] = PW_TILE + (8 * g + t4) * 256 + (((wave_ci * 2 + i) ^ t4) << 6) + 32 * G16 + 8 * (sl & 3);
    }
    auto tr = [](const unsigned char* a0) {
        typedef __attribute__((ext_vector_type(4))) short s16x4_;
        const s16x4_ lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_*)a0);
        const s16x4_ hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_*)(a0 + 4 * 256));
        const __attribute__((ext_vector_type(8))) short v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        return *reinterpret_cast<const bf16x8*>(&v);
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    // bias gradient = dy^T x 1: one more MFMA per k-step and cout tile against an all-ones operand, by the waves of the first cin tile
    const bool do_bias = p.part_bias != nullptr && ci_t == 0 && wave_ci == 0;     // (wave-uniform)
    f32x16 accb[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) accb[i][r] = 0.0f;
    bf16x8 ones;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = (bf16_t)1.0f;

    const int n_mine = (p.n_chunks - split + p.nsplit - 1) / p.nsplit;          // chunks split, split + nsplit, ...
    if (n_mine > 0) issue(split, 0);
    auto chunk = [&](int it, auto stage_c) {
        constexpr int ST = decltype(stage_c)::value;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (it + 1 < n_mine) issue(split + (it + 1) * p.nsplit, ST ^ 1);
        const unsigned char* sb = pw_smem + ST * PW_STAGE;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            bf16x8 afr[2], bfr[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) { afr[i] = tr(sb + a_off[i] + ks * 16 * 256); bfr[i] = tr(sb + b_off[i] + ks * 16 * 256); }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) mma16(acc[i][j], afr[i], bfr[j]);               // D[cout][cin]
            if (do_bias) { mma16(accb[0], afr[0], ones); mma16(accb[1], afr[1], ones); }   // every column = sum over the 16 pixels of dy[.][cout]
        }
    };
    for (int it = 0; it < n_mine; it += 2) {
        chunk(it, std::integral_constant<int, 0>{});
        if (it + 1 < n_mine) chunk(it + 1, std::integral_constant<int, 1>{});
    }
    float* pw = p.part + (size_t)split * ((size_t)p.Cout * p.Cin);               // this work-group's slab: plain coalesced stores
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co0 + wave_co * 64 + i * 32 + acc_row(lane, r);
                const int ci = ci0 + wave_ci * 64 + j * 32 + l31;
                pw[(size_t)co * p.Cin + ci] = acc[i][j][r];
            }
    if (do_bias && l31 == 0) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) p.part_bias[(size_t)split * p.Cout + co0 + wave_co * 64 + i * 32 + acc_row(lane, r)] = accb[i][r];
    }
}

}  // namespace

// Returns 1 if the convolution is a plain 1x1 GEMM this kernel takes (and the launch was made), 0 if the caller should use conv_fwd.hip.
int mas_conv1x1_try(const MasConvDesc* d, const void* x, const void* w_packed, const float* bias, const void* residual, void* y, hipStream_t s) {
    static const int on = mas_env_int("MAS_CONV1X1", 1);
    if (!on) return 0;
    if (d->ks != 1 || d->stride != 1 || d->upsample || d->act != MAS_ACT_NONE || d->pad_top || d->pad_left) return 0;
    if (d->in_dtype != MAS_BF16 || d->out_dtype != MAS_BF16 || d->w_layout != MAS_WLAYOUT_K64) return 0;
    if (d->Cin % 64 || d->Cout % 128 || d->Ho != d->H || d->Wo != d->W) return 0;
    const long long M = (long long)d->N * d->H * d->W;
    if (M * d->Cin * 2 >= 0x7fffffffLL || M * d->Cout * 2 >= 0x7fffffffLL) return 0;
    PwParams p;
    p.a = (const unsigned char*)x; p.w = (const unsigned char*)w_packed; p.bias = bias; p.res = (const unsigned char*)residual; p.y = (unsigned char*)y;
    p.M = (int)M; p.K = d->Cin; p.N = d->Cout;
    p.rows_pad = mas_roundup(d->Cout, 128); p.n_chunks = d->Cin / 64; p.n_nt = d->Cout / 128;
    static mas_devmask_t attr{0};               // per device, like every other launcher (mas_common.h)
    unsigned long long attr_bit;
    if (mas_attr_needed(attr, &attr_bit)) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(conv1x1_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, PW_LDS) != hipSuccess) { (void)hipGetLastError(); return 0; }
        mas_attr_done(attr, attr_bit);
    }
    const long long grid = (M + 127) / 128 * p.n_nt;
    static const int bands = mas_env_int("MAS_CONV_XCD_BANDS", 1);
    p.xcd_bands = (bands && p.n_nt > 1 && grid % 8 == 0) ? 1 : 0;
    hipLaunchKernelGGL(conv1x1_kernel, dim3((unsigned)grid), dim3(PW_NT), PW_LDS, s, p);
    MAS_CHECK_LAUNCH("conv1x1");
    return 1;
}

static bool pw_wgrad_setup(const MasConvDesc* d, PwWgradParams& p) {
    static const int on = mas_env_int("MAS_CONV1X1", 1);
    if (!on) return false;
    if (d->ks != 1 || d->stride != 1 || d->upsample || d->act != MAS_ACT_NONE || d->pad_top || d->pad_left) return false;
    if (d->in_dtype != MAS_BF16 || d->Cin % 128 || d->Cout % 128 || d->Ho != d->H || d->Wo != d->W) return false;
    const long long M = (long long)d->N * d->H * d->W;
    if (M * d->Cin * 2 >= 0x7fffffffLL || M * d->Cout * 2 >= 0x7fffffffLL) return false;
    p.M = (int)M; p.Cout = d->Cout; p.Cin = d->Cin; p.n_co_t = d->Cout / 128; p.n_ci_t = d->Cin / 128;
    p.n_chunks = (int)((M + 63) / 64);
    const int tiles = p.n_co_t * p.n_ci_t;
    int ns = mas_cdiv(2 * mas_num_cus(), tiles);                               // two work-groups per CU
    if (ns > p.n_chunks / 2) ns = p.n_chunks / 2;                                 // at least two chunks per work-group
    if (ns > 256) ns = 256;
    if (ns < 1) ns = 1;
    p.nsplit = ns;
    return true;
}

// ---- fp32 1x1 weight gradient (quant_conv / post_quant_conv around the quantiser: reference models/vqvae.py, fp32 also under autocast) --
// dW[co][ci] = sum_p dy[p][co] x[p][ci] in exact fp32 FMAs, split over the pixels into slabs [nsplit][Cout][Cin] that mas_wgrad_reduce
// folds in a fixed order: bitwise run-to-run deterministic, where the general fp32 kernel of conv_wgrad.hip commits with fp32 atomics
// (round 5: these two layers and the codebook were the last parameter gradients of the VQ-IMG step that were not).  A work-group =
// 64 couts x 64 cins x one pixel slice, 32 pixels per LDS stage, 4 x 4 outputs per thread.  1 GFLOP per layer: no MFMA needed.
namespace {
constexpr int PF_NT = 256, PF_T = 64, PF_PX = 32;
struct PwF32Params {
    const float* dy; const float* x; float* part; float* part_bias;
    int M, Cout, Cin, n_co_t, n_ci_t, nsplit, px_per_split;
};
__global__ __launch_bounds__(PF_NT) void wgrad1x1_f32_kernel(PwF32Params p) {
    __shared__ __attribute__((aligned(16))) float dys[PF_PX][PF_T];
    __shared__ __attribute__((aligned(16))) float xs[PF_PX][PF_T];
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    int b = blockIdx.x;
    const int split = b % p.nsplit; b /= p.nsplit;
    const int ci_t = b % p.n_ci_t, co_t = b / p.n_ci_t;
    const int co0 = co_t * PF_T, ci0 = ci_t * PF_T;
    const int p_lo = split * p.px_per_split, p_hi = min(p.M, p_lo + p.px_per_split);
    float acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.0f;
    float bsum[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    for (int pb = p_lo; pb < p_hi; pb += PF_PX) {
        __syncthreads();
#pragma unroll
        for (int u = tid; u < PF_PX * PF_T / 4; u += PF_NT) {                 // two float4 per thread and tensor
            const int pr = u / (PF_T / 4), c4 = (u % (PF_T / 4)) * 4;
            const int px = pb + pr;
            f32x4 a = {0.0f, 0.0f, 0.0f, 0.0f}, c = {0.0f, 0.0f, 0.0f, 0.0f};
            if (px < p_hi) {
                if (co0 + c4 < p.Cout) a = *reinterpret_cast<const f32x4*>(p.dy + (size_t)px * p.Cout + co0 + c4);       // (Cout, Cin % 4 == 0)
                if (ci0 + c4 < p.Cin) c = *reinterpret_cast<const f32x4*>(p.x + (size_t)px * p.Cin + ci0 + c4);
            }
            *reinterpret_cast<f32x4*>(&dys[pr][c4]) = a;
            *reinterpret_cast<f32x4*>(&xs[pr][c4]) = c;
        }
        __syncthreads();
#pragma unroll 8
        for (int pr = 0; pr < PF_PX; ++pr) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(&dys[pr][ty * 4]);
            const f32x4 c = *reinterpret_cast<const f32x4*>(&xs[pr][tx * 4]);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                bsum[i] += a[i];
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_fmaf(a[i], c[j], acc[i][j]);
            }
        }
    }
    float* pw = p.part + (size_t)split * ((size_t)p.Cout * p.Cin);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int co = co0 + ty * 4 + i, ci = ci0 + tx * 4;
        if (co < p.Cout && ci < p.Cin) *reinterpret_cast<f32x4*>(pw + (size_t)co * p.Cin + ci) = f32x4{acc[i][0], acc[i][1], acc[i][2], acc[i][3]};
        if (p.part_bias && ci_t == 0 && tx == 0 && co < p.Cout) p.part_bias[(size_t)split * p.Cout + co] = bsum[i];
    }
}
bool pw_f32_setup(const MasConvDesc* d, PwF32Params& p) {
    static const int on = mas_env_int("MAS_WGRAD1X1_F32", 1);
    if (!on) return false;
    if (d->ks != 1 || d->stride != 1 || d->upsample || d->act != MAS_ACT_NONE || d->pad_top || d->pad_left) return false;
    if (d->in_dtype != MAS_F32 || d->Cin % 4 || d->Cout % 4 || d->Ho != d->H || d->Wo != d->W) return false;
    const long long M = (long long)d->N * d->H * d->W;
    if (M <= 0 || M >= 0x7fffffffLL) return false;
    p.M = (int)M; p.Cout = d->Cout; p.Cin = d->Cin; p.n_co_t = mas_cdiv(d->Cout, PF_T); p.n_ci_t = mas_cdiv(d->Cin, PF_T);
    const int tiles = p.n_co_t * p.n_ci_t;
    int ns = mas_cdiv(2 * mas_num_cus(), tiles);                               // ~two work-groups per CU
    const int max_ns = mas_cdiv(p.M, 4 * PF_PX);                                  // at least four LDS stages per work-group
    if (ns > max_ns) ns = max_ns;
    if (ns > 256) ns = 256;
    if (ns < 1) ns = 1;
    p.px_per_split = mas_roundup(mas_cdiv(p.M, ns), PF_PX);
    p.nsplit = mas_cdiv(p.M, p.px_per_split);
    return true;
}
}  // namespace

// split-K factor of the 1x1 weight-gradient kernels for this convolution, 0 when neither takes it (mas_conv_wgrad_splits forwards here)
int mas_wgrad1x1_splits(const MasConvDesc* d) {
    PwWgradParams p;
    if (pw_wgrad_setup(d, p)) return p.nsplit;
    PwF32Params q;
    return pw_f32_setup(d, q) ? q.nsplit : 0;
}

// part [nsplit][Cout][Cin] fp32 and (non-NULL) part_bias [nsplit][Cout], every element written once (see mas_conv_wgrad_partial)
int mas_wgrad1x1_partial(const MasConvDesc* d, const void* x, const void* dy, float* part, float* part_bias, hipStream_t s) {
    PwWgradParams p;
    if (!pw_wgrad_setup(d, p)) {
        PwF32Params q;
        if (!pw_f32_setup(d, q)) return 0;
        q.dy = (const float*)dy; q.x = (const float*)x; q.part = part; q.part_bias = part_bias;
        hipLaunchKernelGGL(wgrad1x1_f32_kernel, dim3((unsigned)(q.n_co_t * q.n_ci_t * q.nsplit)), dim3(PF_NT), 0, s, q);
        MAS_CHECK_LAUNCH("wgrad1x1_f32");
        return 1;
    }
    p.dy = (const unsigned char*)dy; p.x = (const unsigned char*)x; p.part = part; p.part_bias = part_bias;
    hipLaunchKernelGGL(wgrad1x1_kernel, dim3((unsigned)(p.n_co_t * p.n_ci_t * p.nsplit)), dim3(PW_NT), PW_LDS, s, p);
    MAS_CHECK_LAUNCH("wgrad1x1");
    return 1;
}
