// `Upsample` + 3x3 convolution in its sub-pixel form, forward and data gradient (reference models/modules.py:44-59: nearest x2, then
// Conv2d(3, 1, 1), and autograd of the same site with respect to the input).
//
// The reference convolves a nearest-x2 image: output pixel (2i + a, 2j + b) sees only the 2x2 window
//     x[i + a - 1 + r][j + b - 1 + s],  r, s in {0, 1}
// of the LOW-resolution input, with weights that are sums of the 3x3 taps falling on the same source pixel,
//     Wp[a][b][r][s] = sum of W[kh][kw] over kh in R(a, r), kw in R(b, s);   R(0,0) = {0}, R(0,1) = {1,2}, R(1,0) = {0,1}, R(1,1) = {2}
// -- 16 tap-pixels per low-resolution pixel instead of 36, i.e. 2.25x fewer FLOPs for the same result (the kernels before this
// one folded the x2 into the address arithmetic and still ran all 9 taps at the high resolution).  Under the package power cap
// (profiles/r05_energy_budget.txt: the convolution kernels' time is joules / cap) fewer FLOPs are the one lever that is not paid
// back by the clock.
//
// One kernel, two uses, both walking the LOW-resolution grid in the wide kernel's geometry (conv3x3_wide.hip: 16 x 32-pixel tiles,
// 8 waves, wave tile = 128 couts x 64 pixels, 32-channel chunks, LDS-DMA for everything HBM-facing, XOR-swizzled 64-byte rows):
//   forward        : input plain [N,H,W,Cin]; a tile = (spatial tile, output phase (a, b), 128-cout tile); K = Cin x 4 taps; output
//                    stored at pixel stride 2 into [N,2H,2W,Cout]; optional GroupNorm statistics of the output (one table row per
//                    tile and phase), as in the wide kernel;
//   data gradient  : input = the four phase images of dy [N,2H,2W,Cout] (pixel stride 2: only the DMA plan changes), part of the K
//                    loop: K = 4 phases x Cout x 4 taps; output plain [N,H,W,Cin] -- the x2 sum-pooling pass of the old path
//                    (`mas_sumpool2x` over a 4x larger tensor) disappears.
//       da[i][j] = sum over (a, b, r, s) of Wp[a][b][r][s]^T dy[2 (i + 1 - a - r) + a][2 (j + 1 - b - s) + b]
//     = phase image (a, b) read through the 2x2 window at patch offset (1 - a, 1 - b) with the taps flipped: the packed image of the
//       data gradient (mas_pack_conv_weight_layout, MAS_WLAYOUT_UP2, transpose = 1) stores tap (1 - r, 1 - s), in/out swapped.
// A stage = one 32-channel chunk x 4 taps (64 MFMAs per wave and barrier: the 3x3 kernel's stage has 48), 32 KiB of weights.
// LDS: 2 x 32 KiB weights + 2 x 39 KiB patch + bias + statistics scratch = 158 KiB, one 512-thread work-group per CU.
// Tiles that share a patch (the 4 phases x Cout / 128 tiles of a spatial tile) are mapped to the SAME XCD (block b runs on XCD b % 8):
// the input is fetched into one L2 and hit there by the others.
#include "mas_common.h"
#include <algorithm>
#include <utility>

namespace {

template <int... I, typename F>
__device__ __forceinline__ void u_static_for(std::integer_sequence<int, I...>, F&& f) { (f(std::integral_constant<int, I>{}), ...); }

struct Up2Params {
    const unsigned char* x; const unsigned char* w; const float* bias; unsigned char* y;
    float* stats;                              // optional [N][tiles_h * tiles_w * o_phases][Cout][2]
    int N, H, W;                               // the low-resolution grid the tiles walk
    int Cin, Cout;                             // channels of THIS launch's input / output tensor
    int n_chunks, Cout_pad;
    int k_phases, o_phases;                    // (1, 4) forward, (4, 1) data gradient
    int in_px, in_row, in_ph_row, in_ph_px;    // byte strides between low-resolution neighbours of the input; byte offset of phase row / column
    unsigned in_img;                           // bytes of one input image
    int out_px, out_row, out_ph_row, out_ph_px;
    unsigned out_img, out_bytes;
    int tiles_h, tiles_w, n_ct, n_spatial, group;     // group = o_phases * n_ct tiles share a patch
    unsigned m_group, m_ct, m_tw, m_th;        // ceil(2^32 / d)
    int flip;                                  // patch offset of phase (a, b): (1 - a, 1 - b) instead of (a, b)
};

constexpr int U_PWL = 34;
constexpr int U_NPIX = 18 * 34;
constexpr int U_NPIECE = 39;
constexpr int U_PATCH = U_NPIECE * 1024;
constexpr int U_WT = 128 * 64;
constexpr int U_WSTAGE = 4 * U_WT;             // the four taps of one chunk
constexpr int U_WBUF = 0;
constexpr int U_PBUF = 2 * U_WSTAGE;
constexpr int U_BIAS = U_PBUF + 2 * U_PATCH;
constexpr int U_MAXCOUT = 1024;
constexpr int U_STAT = U_BIAS + U_MAXCOUT * 4;            // [8 waves][128][2] fp32 + {table row, cout offset}
constexpr int U_NEXT = U_STAT + 8 * 128 * 2 * 4 + 16;     // [512 threads][2] the next tile's output offsets, then per wave {w0, statistics row}
constexpr int U_LDS = U_NEXT + 512 * 8 + 8 * 8;
constexpr int U_NSLOT = 5;
constexpr int U_OOB = (int)0x80000000;
static_assert(U_LDS <= 160 * 1024, "LDS budget");

#define U_WAIT_BARRIER(N) do { asm volatile("s_waitcnt vmcnt(" #N ") lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); \
                               asm volatile("" ::: "memory"); } while (0)

template <bool STATS>
__global__ __launch_bounds__(512, 2) void conv_up2_kernel(Up2Params p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* const wbuf = smem + U_WBUF;
    unsigned char* const patch = smem + U_PBUF;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // wave w owns tile rows 2w, 2w+1 and all 128 couts
    const int g = lane >> 5, l31 = lane & 31;

    const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, p.out_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(p.w), 0,
                                                                           (unsigned)(16 * p.n_chunks * p.Cout_pad * 64), 0x00020000);

    // ---- tiles.  Virtual tile t: XCD t & 7 (= the XCD of the work-group that runs it: the grid is a multiple of 8), k = t >> 3;
    //      spatial tile (k / group) * 8 + xcd, member k % group = phase * n_ct + cout tile.  Spatial tiles past the end (the grid of
    //      spatial tiles is rounded up to 8) are walked with every load out of range and every store dropped.
    const int total_tiles = ((p.n_spatial + 7) >> 3) * 8 * p.group;
    struct Tile { int n, h0, w0, c0, ph; bool valid; };
    auto udiv = [](int t, unsigned m, int d, int& q, int& r) {
        q = (int)__umulhi((unsigned)t, m);
        if (d == 1) q = t;
        r = t - q * d;
    };
    auto decode = [&](int t) {
        Tile tc;
        const int xcd = t & 7;
        int k = t >> 3, kg, within, ct, q, tw_i, th_i;
        udiv(k, p.m_group, p.group, kg, within);
        udiv(within, p.m_ct, p.n_ct, tc.ph, ct);
        int sp = kg * 8 + xcd;
        tc.valid = sp < p.n_spatial;
        udiv(sp, p.m_tw, p.tiles_w, q, tw_i); sp = q;
        udiv(sp, p.m_th, p.tiles_h, q, th_i);
        tc.n = q; tc.c0 = ct * 128; tc.h0 = th_i * 16; tc.w0 = tw_i * 32;
        return tc;
    };

    // ---- patch plan (as conv3x3_wide.hip: slot k of this thread = pixel (lane >> 2) of DMA piece wave + 8 k; wave 7 repeats piece 38)
    auto slot_pix = [&](int k, int& q, int& pr, int& pc) -> bool {
        const int piece = (k < 4) ? wave + 8 * k : (wave < 7 ? 32 + wave : 38);
        q = piece * 16 + (lane >> 2);
        pr = (q * 1928) >> 16;                   // q / 34 for q < 640
        pc = q - pr * U_PWL;
        return q < U_NPIX;
    };
    auto make_plan = [&](const Tile& tc, int (&vo)[U_NSLOT], int (&ob)[4]) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {           // byte offset of (n, row, w0 + 4 g, phase, c0 + 4 l31); the epilogue adds the pixel column
            const int i = tc.h0 + 2 * wave + j;
            ob[j] = (tc.valid && i < p.H) ? (int)((unsigned)tc.n * p.out_img) + i * p.out_row + (tc.w0 + 4 * g) * p.out_px
                                                + (p.o_phases == 4 ? (tc.ph >> 1) * p.out_ph_row + (tc.ph & 1) * p.out_ph_px : 0) + (tc.c0 + 4 * l31) * 2
                                          : U_OOB;
        }
        ob[2] = p.W - tc.w0 - 4 * g;
        ob[3] = tc.valid ? ((tc.n * p.tiles_h + (tc.h0 >> 4)) * p.tiles_w + (tc.w0 >> 5)) * p.o_phases + (p.o_phases == 4 ? tc.ph : 0) : -1;
#pragma unroll
        for (int k = 0; k < U_NSLOT; ++k) {
            int q, pr, pc;
            const bool live = slot_pix(k, q, pr, pc);
            const int ih = tc.h0 + pr - 1, iw = tc.w0 + pc - 1;
            const bool inb = live && tc.valid && (ih >= 0) && (ih < p.H) && (iw >= 0) && (iw < p.W);
            const int sl = (lane & 3) ^ ((q >> 2) & 3);
            vo[k] = inb ? ih * p.in_row + iw * p.in_px + sl * 16 : U_OOB;
        }
    };
    auto p_dma = [&](__amdgpu_buffer_rsrc_t rs, const int (&vo)[U_NSLOT], int soff_, int buf) {
        const int soff = __builtin_amdgcn_readfirstlane(soff_);
#pragma unroll
        for (int k = 0; k < U_NSLOT; ++k) {
            const int piece = (k < 4) ? wave + 8 * k : (wave < 7 ? 32 + wave : 38);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(patch + buf * U_PATCH + piece * 1024),
                                                     16, vo[k], soff, 0, 0);
        }
    };
    // ---- weight stage: 4 tap tiles of 8 KiB; wave w moves piece w (rows 16w..16w+15) of each
    const int wlane = lane * 16;
    const int wstride = p.Cout_pad * 64;
    auto w_issue = [&](int t0, int c0, int sel) {          // taps t0 .. t0 + 3 (t = (phase * n_chunks + chunk) * 4 + tap) of cout tile c0
        const int soff = __builtin_amdgcn_readfirstlane(t0 * wstride + c0 * 64 + wave * 1024);
        unsigned char* dst = wbuf + sel * U_WSTAGE + wave * 1024;
#pragma unroll
        for (int t = 0; t < 4; ++t)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (__attribute__((address_space(3))) void*)(dst + t * U_WT), 16, wlane, soff + t * wstride, 0, 0);
    };

    // ---- per-lane fragment addressing (conv3x3_wide.hip)
    const int a_off = l31 * 64 + ((g ^ ((l31 >> 2) & 3)) << 4);
    const int pj0 = (2 * wave) * U_PWL + l31;
    auto b_addr = [&](int P) { return P * 64 + (((g ^ (P >> 2)) & 3) << 4); };

    // ---- prologue
    int tile = blockIdx.x;
    int c0_cur, c0_nxt, ph_cur, ph_nxt, ob_cur[4];
    int vo[U_NSLOT];
    __amdgpu_buffer_rsrc_t rs_x;
    auto img_rsrc = [&](int n) {
        const int nn = n < p.N ? n : p.N - 1;
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(p.x) + (size_t)nn * p.in_img, 0, p.in_img, 0x00020000);
    };
    {
        const Tile t0 = decode(tile);
        make_plan(t0, vo, ob_cur);
        c0_cur = c0_nxt = t0.c0; ph_cur = ph_nxt = t0.ph;
        rs_x = img_rsrc(t0.n);
    }
    {
        float* bl = reinterpret_cast<float*>(smem + U_BIAS);
        for (int c = tid; c < p.Cout; c += 512) bl[c] = p.bias ? p.bias[c] : 0.0f;
    }
    const bool kph4 = p.k_phases == 4;
    // stage q of a tile: input phase kq = q / n_chunks (data gradient; 0 forward), chunk cq = q % n_chunks
    auto w_t0 = [&](int ph_tile, int kq, int cq) { return (((kph4 ? kq : ph_tile) * p.n_chunks) + cq) * 4; };
    auto x_soff = [&](int kq, int cq) { return cq * 64 + (kq >> 1) * p.in_ph_row + (kq & 1) * p.in_ph_px; };
    w_issue(w_t0(ph_cur, 0, 0), c0_cur, 0);
    p_dma(rs_x, vo, x_soff(0, 0), 0);

    auto stats_flush = [&]() {
        if (tid < 256) {
            const float* sl = reinterpret_cast<const float*>(smem + U_STAT);
            const int* meta = reinterpret_cast<const int*>(smem + U_STAT + 8 * 128 * 2 * 4);
            float t = 0.0f;
#pragma unroll
            for (int w = 0; w < 8; ++w) t += sl[w * 256 + tid];
            if (meta[0] >= 0) p.stats[((size_t)meta[0] * p.Cout + meta[1]) * 2 + tid] = t;
        }
    };
    bool stores_in_flight = false;
    const int NQ = p.k_phases * p.n_chunks;      // even: Cin % 64 == 0

    for (;;) {
        const int next_tile = tile + (int)gridDim.x;
        const bool has_next = next_tile < total_tiles;

        f32x16 acc[4][2];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

        int kq = 0, cq = 0;                      // phase / chunk of the stage being COMPUTED
        for (int q = 0; q < NQ; q += 2) {
            u_static_for(std::make_integer_sequence<int, 2>{}, [&](auto s_c) {
                constexpr int s = decltype(s_c)::value;            // = buffer of this stage
                const bool last = q + s + 1 == NQ;
                // ---- barrier: this stage's weights and patch have landed (they are the youngest vector-memory operations, except for
                //      the previous tile's 32 epilogue stores at a tile's first stage); the other buffers are free
                if (s == 0 && q == 0 && stores_in_flight) {
                    U_WAIT_BARRIER(32); stores_in_flight = false;
                    if constexpr (STATS) stats_flush();
                } else U_WAIT_BARRIER(0);
                // ---- the next stage's coordinates; at a tile's last stage: the next tile's plan
                int kn = kq, cn = cq + 1;
                if (cn == p.n_chunks) { cn = 0; ++kn; }
                if (last) {
                    kn = 0; cn = 0;
                    if (has_next) {
                        const Tile nt = decode(next_tile);
                        int ob_n[4];
                        make_plan(nt, vo, ob_n);
                        *reinterpret_cast<u32x2*>(smem + U_NEXT + tid * 8) = u32x2{(unsigned)ob_n[0], (unsigned)ob_n[1]};
                        // (uniform values, written and read back by the SAME wave: LDS operations of a wave execute in order, no barrier needed)
                        if (lane == 0) { int* m = reinterpret_cast<int*>(smem + U_NEXT + 512 * 8) + 2 * wave; m[0] = nt.w0; m[1] = ob_n[3]; }
                        c0_nxt = nt.c0; ph_nxt = nt.ph;
                        rs_x = img_rsrc(nt.n);
                    }
                }
                // ---- DMA for the next stage: 4 weight pieces, 5 patch pieces per wave
                w_issue(w_t0(last ? ph_nxt : ph_cur, kn, cn), last ? c0_nxt : c0_cur, s ^ 1);
                asm volatile("" ::: "memory");
                p_dma(rs_x, vo, x_soff(kn, cn), s ^ 1);
                asm volatile("" ::: "memory");
                // ---- 4 taps x 2 k-steps of 8 MFMAs; the 2x2 window sits at patch offset (por, poc) = the phase (flipped for the data gradient)
                const int code = p.flip ? 3 - (kph4 ? kq : ph_cur) : (kph4 ? kq : ph_cur);
                // (the offset is a run-time, wave-uniform value: six fragment addresses per stage are recomputed from it -- four compile-time
                //  instances of the MFMA block behind a branch made hipcc spill 2000 registers)
                {
                    const int pq = pj0 + (code >> 1) * U_PWL + (code & 1);
                    int ba[3][2];
#pragma unroll
                    for (int rr = 0; rr < 3; ++rr)
#pragma unroll
                        for (int ts = 0; ts < 2; ++ts) ba[rr][ts] = b_addr(pq + rr * U_PWL + ts);
                    const unsigned char* wb = wbuf + s * U_WSTAGE;
                    const unsigned char* pb = patch + s * U_PATCH;
                    bf16x8 afr[2][4], bfr[2][2];
                    auto ld_k = [&](int n, int b) {           // n = 0..7: tap = n >> 1 = (tr, ts), kk = n & 1
                        const int tap = n >> 1, kx = (n & 1) << 5;
                        const int tr = tap >> 1, ts = tap & 1;
#pragma unroll
                        for (int j = 0; j < 2; ++j)
                            bfr[b][j] = *reinterpret_cast<const bf16x8*>(pb + (ba[j + tr][ts] ^ kx));
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                            afr[b][i] = *reinterpret_cast<const bf16x8*>(wb + tap * U_WT + i * 2048 + (a_off ^ kx));
                    };
                    ld_k(0, 0);
#pragma unroll
                    for (int n = 0; n < 8; ++n) {
                        if (n + 1 < 8) ld_k(n + 1, (n + 1) & 1);
#pragma unroll
                        for (int i = 0; i < 4; ++i)
#pragma unroll
                            for (int j = 0; j < 2; ++j) mma16(acc[i][j], bfr[n & 1][j], afr[n & 1][i]);
                    }
                    __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);
#pragma unroll
                    for (int n = 0; n + 1 < 8; ++n) {
                        __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);
                        __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
                    }
                    __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
                }
                kq = kn; cq = cn;
            });
        }

        // ---- epilogue (conv3x3_wide.hip): a lane owns 4 consecutive couts of the 16 pixels of each register block: one 8-byte store
        //      per pixel and lane, a half-wave writes one whole NHWC pixel row of the cout tile
        {
            const f32x4 bv = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(smem + U_BIAS) + c0_cur + 4 * l31);
            f32x2 st_s[2] = {f32x2{0.0f, 0.0f}, f32x2{0.0f, 0.0f}}, st_q[2] = {f32x2{0.0f, 0.0f}, f32x2{0.0f, 0.0f}};
            if constexpr (STATS) {
                if (!__all(ob_cur[2] > 27 && ob_cur[0] != U_OOB && ob_cur[1] != U_OOB)) {   // ragged tiles: pixels outside the map count as 0
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int pc = (r & 3) + 8 * (r >> 2);
                            if (!((pc < ob_cur[2]) && (ob_cur[j] != U_OOB))) {
#pragma unroll
                                for (int i = 0; i < 4; ++i) acc[i][j][r] = -bv[i];
                            }
                        }
                }
            }
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float v[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] = acc[i][j][r] + bv[i];
                    if constexpr (STATS) {
                        const f32x2 a = {v[0], v[1]}, b = {v[2], v[3]};
                        st_s[0] += a; st_s[1] += b;
                        st_q[0] = a * a + st_q[0]; st_q[1] = b * b + st_q[1];
                    }
                    u32x2 o;
                    bf16_t* ob = reinterpret_cast<bf16_t*>(&o);
#pragma unroll
                    for (int i = 0; i < 4; ++i) ob[i] = (bf16_t)v[i];
                    const int pc = (r & 3) + 8 * (r >> 2);
                    const bool ok = (pc < ob_cur[2]) && (ob_cur[j] != U_OOB);
                    __builtin_amdgcn_raw_buffer_store_b64(o, rs_y, ok ? ob_cur[j] : U_OOB, pc * p.out_px, 0);
                }
            stores_in_flight = true;
            if constexpr (STATS) {
                float* sl = reinterpret_cast<float*>(smem + U_STAT) + wave * 256 + (4 * l31) * 2;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float ss_ = st_s[i >> 1][i & 1], qq_ = st_q[i >> 1][i & 1];
                    const float a = ss_ + __shfl_xor(ss_, 32), b = qq_ + __shfl_xor(qq_, 32);
                    if (g == 0) { sl[2 * i] = a; sl[2 * i + 1] = b; }
                }
                if (tid == 0) { int* meta = reinterpret_cast<int*>(smem + U_STAT + 8 * 128 * 2 * 4); meta[0] = ob_cur[3]; meta[1] = c0_cur; }
            }
        }
        if (!has_next) break;
        tile = next_tile; c0_cur = c0_nxt; ph_cur = ph_nxt;
        {
            const u32x2 nx = *reinterpret_cast<const u32x2*>(smem + U_NEXT + tid * 8);
            const int* m = reinterpret_cast<const int*>(smem + U_NEXT + 512 * 8) + 2 * wave;
            ob_cur[0] = (int)nx[0]; ob_cur[1] = (int)nx[1]; ob_cur[2] = p.W - m[0] - 4 * g; ob_cur[3] = m[1];
        }
    }
    if constexpr (STATS) {
        __syncthreads();
        stats_flush();
    }
}

template <bool STATS>
int launch_up2(const Up2Params& p, hipStream_t s, const char* name) {
    auto kern = conv_up2_kernel<STATS>;
    static mas_devmask_t attr_mask{0};
    unsigned long long attr_bit;
    if (mas_attr_needed(attr_mask, &attr_bit)) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, U_LDS) != hipSuccess)
            MAS_FAIL(MAS_ELAUNCH, "%s: cannot set dynamic LDS size %d", name, U_LDS);
        mas_attr_done(attr_mask, attr_bit);
    }
    const long long tiles = (long long)((p.n_spatial + 7) / 8) * 8 * p.group;
    long long resident = 4LL * mas_num_cus();              // (conv3x3_wide.hip: 4x oversubscription)
    static const int wgs_per_cu = mas_env_int("MAS_CONV_WGS_PER_CU", 0);
    if (wgs_per_cu > 0) resident = (long long)wgs_per_cu * mas_num_cus();
    resident = resident / 8 * 8;                           // the tile -> XCD map assumes a grid that is a multiple of 8
    if (resident < 8) resident = 8;
    const unsigned blocks = (unsigned)(tiles < resident ? tiles : resident);       // (tiles is a multiple of 8)
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(512), U_LDS, s, p);
    MAS_CHECK_LAUNCH(name);
    return MAS_OK;
}

unsigned up2_magic(int dv) { return (unsigned)((0x100000000ULL + (unsigned)dv - 1) / (unsigned)dv); }

// geometry shared by both directions: d is the FORWARD convolution (upsample = 1; H x W the input map, Ho x Wo = 2H x 2W)
bool up2_geometry_ok(const MasConvDesc* d) {
    static const int mode = mas_env_int("MAS_CONV_UP2", 1);
    if (!mode || !d) return false;
    if (!d->upsample || d->ks != 3 || d->stride != 1 || d->pad_top != 1 || d->pad_left != 1) return false;
    if (d->in_dtype != MAS_BF16 || d->out_dtype != MAS_BF16 || d->act != MAS_ACT_NONE) return false;
    if (d->Ho != 2 * d->H || d->Wo != 2 * d->W || d->N <= 0 || d->H <= 0 || d->W <= 0) return false;
    static const int any_width = mas_env_int("MAS_CONV_WIDE_ANY_WIDTH", 0);   // tests: ragged tile columns at small sizes
    if (!any_width && (d->W < 32 || 4 * d->W < 3 * 32 * mas_cdiv(d->W, 32))) return false;
    const long long in_bytes = (long long)d->H * d->W * d->Cin * 2, out_img = (long long)d->Ho * d->Wo * d->Cout * 2;
    if (in_bytes * d->N >= 0x7fffffffLL || out_img * d->N >= 0x7fffffffLL) return false;
    return true;
}

bool up2_tiles_ok(const MasConvDesc* d, int group) {
    const long long n_spatial = (long long)d->N * mas_cdiv(d->H, 16) * mas_cdiv(d->W, 32);
    const long long tiles = (n_spatial + 7) / 8 * 8 * group;
    static const int min_per_cu = mas_env_int("MAS_CONV_WIDE_MIN_TILES_PER_CU", 1);
    if (tiles < (long long)min_per_cu * mas_num_cus() || tiles > 0x3fffffffLL) return false;
    const long long dmax = std::max<long long>(group, std::max(mas_cdiv(d->H, 16), mas_cdiv(d->W, 32)));
    return tiles * dmax < 0x100000000LL;                   // the multiply-high tile decode is exact below this
}

void up2_tiles(Up2Params& p) {
    p.tiles_h = mas_cdiv(p.H, 16); p.tiles_w = mas_cdiv(p.W, 32); p.n_ct = p.Cout / 128;
    p.n_spatial = p.N * p.tiles_h * p.tiles_w; p.group = p.o_phases * p.n_ct;
    p.m_group = up2_magic(p.group); p.m_ct = up2_magic(p.n_ct); p.m_tw = up2_magic(p.tiles_w); p.m_th = up2_magic(p.tiles_h);
    p.Cout_pad = mas_roundup(p.Cout, 128);
}

}  // namespace

// forward: does this convolution take the sub-pixel kernel (and therefore the MAS_WLAYOUT_UP2 weight image)?
bool mas_conv_up2_fwd_eligible(const MasConvDesc* d) {
    if (!up2_geometry_ok(d)) return false;
    if (d->Cin % 64 != 0 || d->Cout % 128 != 0 || d->Cout > U_MAXCOUT) return false;
    return up2_tiles_ok(d, 4 * (d->Cout / 128));
}

int mas_conv_up2_stat_rows(const MasConvDesc* d) { return mas_cdiv(d->H, 16) * mas_cdiv(d->W, 32) * 4; }

int mas_conv_up2_fwd_launch(const MasConvDesc* d, const void* x, const void* w_packed, const float* bias, void* y, float* stats, hipStream_t s) {
    Up2Params p;
    p.x = (const unsigned char*)x; p.w = (const unsigned char*)w_packed; p.bias = bias; p.y = (unsigned char*)y; p.stats = stats;
    p.N = d->N; p.H = d->H; p.W = d->W; p.Cin = d->Cin; p.Cout = d->Cout; p.n_chunks = d->Cin / 32;
    p.k_phases = 1; p.o_phases = 4; p.flip = 0;
    p.in_px = d->Cin * 2; p.in_row = d->W * d->Cin * 2; p.in_ph_row = 0; p.in_ph_px = 0; p.in_img = (unsigned)((size_t)d->H * d->W * d->Cin * 2);
    p.out_ph_px = d->Cout * 2; p.out_ph_row = d->Wo * d->Cout * 2; p.out_px = 2 * p.out_ph_px; p.out_row = 2 * p.out_ph_row;
    p.out_img = (unsigned)((size_t)d->Ho * d->Wo * d->Cout * 2); p.out_bytes = (unsigned)((size_t)d->N * p.out_img);
    up2_tiles(p);
    return stats ? launch_up2<true>(p, s, "conv_up2_fwd") : launch_up2<false>(p, s, "conv_up2_fwd");
}

extern "C" int mas_conv_up2_dgrad_supported(const MasConvDesc* d) {
    if (!up2_geometry_ok(d)) return 0;
    if (d->Cout % 64 != 0 || d->Cin % 128 != 0 || d->Cin > U_MAXCOUT) return 0;       // (the kernel's outputs are the forward's input channels)
    return up2_tiles_ok(d, d->Cin / 128) ? 1 : 0;
}

extern "C" int mas_conv_up2_dgrad(const MasConvDesc* d, const void* dy, const void* w_packed_t, void* dx, void* stream) {
    MAS_ENTER();
    if (!d || !dy || !w_packed_t || !dx) MAS_FAIL(MAS_EINVAL, "conv_up2_dgrad: null argument");
    if (!mas_conv_up2_dgrad_supported(d)) MAS_FAIL(MAS_EUNSUPPORTED, "conv_up2_dgrad: unsupported convolution (mas_conv_up2_dgrad_supported == 0)");
    Up2Params p;
    p.x = (const unsigned char*)dy; p.w = (const unsigned char*)w_packed_t; p.bias = nullptr; p.y = (unsigned char*)dx; p.stats = nullptr;
    p.N = d->N; p.H = d->H; p.W = d->W; p.Cin = d->Cout; p.Cout = d->Cin; p.n_chunks = d->Cout / 32;
    p.k_phases = 4; p.o_phases = 1; p.flip = 1;
    p.in_ph_px = d->Cout * 2; p.in_ph_row = d->Wo * d->Cout * 2; p.in_px = 2 * p.in_ph_px; p.in_row = 2 * p.in_ph_row;
    p.in_img = (unsigned)((size_t)d->Ho * d->Wo * d->Cout * 2);
    p.out_px = d->Cin * 2; p.out_row = d->W * d->Cin * 2; p.out_ph_row = 0; p.out_ph_px = 0;
    p.out_img = (unsigned)((size_t)d->H * d->W * d->Cin * 2); p.out_bytes = (unsigned)((size_t)d->N * p.out_img);
    up2_tiles(p);
    return launch_up2<false>(p, reinterpret_cast<hipStream_t>(stream), "conv_up2_dgrad");
}
