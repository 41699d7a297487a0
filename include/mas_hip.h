/*
 * mas_hip.h -- C ABI of libmas_hip.so, the MI355X (gfx950) kernels behind the
 * Make-A-Scene hot path (VQ-IMG / VQ-SEG conv stack, vector quantiser, attention).
 *
 * The reference (CasualGANPapers/Make-A-Scene) has no FFI layer: its arithmetic is
 * reached through torch.nn call sites.  Each entry point below names the reference
 * call site(s) it replaces.  The Python host (make-a-scene_amd/mas_hip/) binds these
 * with ctypes; INTEGRATION.md shows the binding.
 *
 * Conventions
 *   - plain pointers and sizes only; every pointer is a DEVICE pointer unless noted.
 *   - the caller owns every buffer (incl. workspaces); the library never allocates or
 *     frees device memory and keeps no pointer after return.
 *   - asynchronous on the caller's hipStream_t (passed as void*); no hidden syncs.
 *   - return 0 on success, a negative MAS_E* code otherwise; mas_last_error() gives a
 *     thread-local message.  No C++ exception crosses the ABI.
 *   - activations are NHWC ("channels last"), element type MAS_BF16 or MAS_F32,
 *     accumulation is always fp32.
 */
#ifndef MAS_HIP_H
#define MAS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MAS_ABI_VERSION 9

enum { MAS_OK = 0, MAS_EINVAL = -1, MAS_EUNSUPPORTED = -2, MAS_ELAUNCH = -3, MAS_EWORKSPACE = -4 };
enum { MAS_F32 = 0, MAS_BF16 = 1 };
/* input prologue fused into the conv / wgrad loaders */
enum { MAS_ACT_NONE = 0,      /* a = x                                   */
       MAS_ACT_AFFINE = 1,    /* a = x*scale[n,c] + shift[n,c]  (GroupNorm apply, modules.py:40-41) */
       MAS_ACT_AFFINE_SILU = 2 /* a = silu(x*scale + shift)     (+ swish, modules.py:35-37)         */ };

typedef struct MasConvDesc {
    int32_t N, H, W, Cin;       /* input  [N,H,W,Cin]  (the tensor in memory; before any upsample fold) */
    int32_t Ho, Wo, Cout;       /* output [N,Ho,Wo,Cout] */
    int32_t ks;                 /* 1 or 3 (square); bf16 also 4 (forward / data gradient, stride 1 or 2) and 2 / 4 stride 1 (weight gradient):
                                   the PatchGAN discriminator of the loss stack, losses/discriminator.py:20-36 */
    int32_t stride;             /* 1 or 2           */
    int32_t pad_top, pad_left;  /* input row = ho*stride + kh - pad_top ; rows/cols outside [0,H)x[0,W) read 0
                                   (covers padding=1, and the right/bottom-only pad of Downsample, modules.py:76-78) */
    int32_t in_dtype;           /* MAS_F32 | MAS_BF16 : x, residual, (dy for wgrad) */
    int32_t out_dtype;          /* MAS_F32 | MAS_BF16 : y */
    int32_t act;                /* MAS_ACT_* prologue applied to x on load (padding stays exactly 0) */
    int32_t upsample;           /* 1: logical input is nearest-x2 of x (F.interpolate, modules.py:56): pixel (h,w) reads x[h>>1][w>>1];
                                   H,W above are then the PHYSICAL (pre-upsample) size */
    int32_t w_layout;           /* MAS_WLAYOUT_*: how w_packed was packed; mas_conv_fwd requires mas_conv_weight_layout(d)
                                   or MAS_WLAYOUT_K64 (always accepted); ignored by mas_conv_wgrad */
} MasConvDesc;

int         mas_abi_version(void);
const char* mas_last_error(void);
/* Name of the kernel the last successful launch of this process ran ("conv3x3_wide", "conv_s2_fwd", "wgrad_thin", ...; "" before the
 * first one; process-wide, because a backward pass launches from autograd's threads): every entry point dispatches on shape, and
 * tests / per-shape profiles assert which kernel a shape really took.  A diagnostic: meaningful while one thread launches at a time. */
const char* mas_last_kernel(void);

/* ---- weight packing (host-visible layout contract) -------------------------------
 * Packs an OIHW fp32 parameter (nn.Conv2d.weight, e.g. modules.py:93-104) into the
 * kernel layout [ks*ks][Cout_pad][Cin_pad] (dtype `dtype`, zero padded; Cout_pad =
 * roundup(Cout,128), Cin_pad = roundup(Cin,64), so the kernels load tiles without bounds checks).
 *   transpose=0 : forward operand            Wp[t][o][i] = W[o][i][kh][kw], t = kh*ks+kw
 *   transpose=1 : data-gradient operand      Wp[t][i][o] = W[o][i][ks-1-kh][ks-1-kw]
 *                 (conv of dY with the flipped, in/out-swapped filter; then "Cout"=Cin)
 */
size_t mas_packed_weight_elems(int Cout, int Cin, int ks);
int    mas_pack_conv_weight(const float* w_oihw, void* packed, int Cout, int Cin, int ks,
                            int transpose, int dtype, void* stream);
/* Two LDS images exist.  MAS_WLAYOUT_K64 (what mas_pack_conv_weight emits; every kernel but one reads it):
 * [Cin/64 chunks][tap][Cout_pad][128 B], 16-byte slots XOR-swizzled with (row>>1)&7.  MAS_WLAYOUT_K32 (bf16 only; the
 * wide 3x3 kernel, conv3x3_wide.hip): [Cin/32 chunks][tap][Cout_pad][64 B], slots XOR-swizzled with (row>>2)&3.
 * mas_conv_weight_layout(d) tells which image mas_conv_fwd prefers for a convolution; a K64 image is always accepted
 * (the call then takes the kernels that read it).  Same buffer size for both (mas_packed_weight_elems).
 * MAS_WLAYOUT_UP2 (bf16, 3x3 only; conv_up2.hip): the sub-pixel form of `Upsample` + conv (reference models/modules.py:44-59) --
 * output pixel (2i + a, 2j + b) of the convolution over the nearest-x2 image sees the 2x2 window x[i + a - 1 + r][j + b - 1 + s] with
 * Wp[a][b][r][s] = sum of W[kh][kw] over kh in R(a, r), kw in R(b, s), R(0,0) = {0}, R(0,1) = {1,2}, R(1,0) = {0,1}, R(1,1) = {2}
 * (summed in fp32, then rounded).  Image: [phase 2a+b][Cin/32 chunks][tap 2r+s][Cout_pad][64 B], rows and slots as K32;
 * transpose = 1 (data-gradient operand): in/out swapped and tap (1-r, 1-s) stored at (r, s).  mas_packed_weight_elems_up2 elements. */
enum { MAS_WLAYOUT_K64 = 0, MAS_WLAYOUT_K32 = 1, MAS_WLAYOUT_UP2 = 2 };
size_t mas_packed_weight_elems_up2(int Cout, int Cin);
int    mas_conv_weight_layout(const MasConvDesc* d);
int    mas_pack_conv_weight_layout(const float* w_oihw, void* packed, int Cout, int Cin, int ks,
                                   int transpose, int dtype, int layout, void* stream);

/* Batched packing: every item of a device-resident table in ONE launch (a training step repacks ~160 weight images after the
 * optimizer step; one launch instead of 160 dependent 8-us launches).  The table is built by the caller: item i covers work-groups
 * [first_block, first_block + n_blocks) of the grid, n_blocks = mas_pack_batch_blocks(...) for its shape, items in ascending
 * first_block order, total_blocks = their sum.  Same images as mas_pack_conv_weight_layout (K32: bf16 3x3 only).               */
typedef struct MasPackItem {
    const float* w_oihw; void* packed;
    int Cout, Cin, ks, transpose, dtype, layout, first_block, n_blocks;
} MasPackItem;
int    mas_pack_batch_blocks(int Cout, int Cin, int ks, int transpose, int dtype, int layout);
int    mas_pack_conv_weight_batch(const MasPackItem* items_device, int n_items, int total_blocks, void* stream);

/* The bf16 images of many parameters in ONE launch, each parameter READ ONCE: a work-group stages a 64 x 64 x taps tile of the OIHW
 * tensor in LDS (coalesced runs) and writes every image the item lists -- forward / data-gradient operand, K64 / K32 -- from it,
 * zero padding included.  Bitwise the images of mas_pack_conv_weight_layout.  Item i covers work-groups [first_block, first_block +
 * mas_pack_tile_blocks(Cout, Cin, ks)), items in ascending first_block order; max_ks = the largest ks in the table (sizes the LDS).    */
typedef struct MasPackTileItem {
    const float* w_oihw; void* img[4];
    int transpose[4], layout[4];
    int n_img, Cout, Cin, ks, first_block, pad_;
} MasPackTileItem;
int    mas_pack_tile_blocks(int Cout, int Cin, int ks);
int    mas_pack_conv_weight_tiles(const MasPackTileItem* items_device, int n_items, int total_blocks, int max_ks, void* stream);

/* ---- Adam over many tensors in ONE launch (replaces torch.optim.Adam.step of reference train.py:99-103 for fp32 parameters; same
 * arithmetic in the same order as torch's fused kernel: g' = g + wd p; m = b1 m + (1-b1) g'; v = b2 v + (1-b2) g'^2;
 * p -= (lr / bias_correction1) * m / (sqrt(v) / sqrt(bias_correction2) + eps), bias_correction_k = 1 - beta_k^step from the caller).
 * Item i covers work-groups [first_block, first_block + mas_adam_blocks(n)); items in ascending first_block order, table in DEVICE memory.
 * No amsgrad, no maximize; p, g, m, v fp32 of n elements each (16-byte aligned tensors take the vector path).                        */
typedef struct MasAdamItem {
    float* p; const float* g; float* m; float* v;
    long long n;
    int first_block, pad_;
} MasAdamItem;
int    mas_adam_blocks(long long numel);
int    mas_adam_multi(const MasAdamItem* items_device, int n_items, int total_blocks, float lr, float beta1, float beta2, float eps,
                      float weight_decay, double bias_correction1, double bias_correction2, void* stream);

/* ---- GroupNorm statistics (replaces the reduction half of torch.nn.GroupNorm,
 * modules.py:40-41).  x: [N,HW,C] NHWC.  Outputs:
 *   mean_rstd [N][G][2] fp32, scale_shift [N][C][2] fp32 with
 *   scale = rstd*gamma, shift = beta - mean*rstd*gamma  (consumed by the conv prologue).
 * workspace: mas_gn_stats_workspace(N,C) bytes.                                      */
size_t mas_gn_stats_workspace(int N, int C);
int    mas_gn_stats(const void* x, int dtype, int N, int HW, int C, int G, float eps,
                    const float* gamma, const float* beta, float* mean_rstd, float* scale_shift,
                    void* workspace, size_t ws_bytes, void* stream);

/* ---- GroupNorm(+SiLU) backward.  Given da = dL/d act(gn(x)) computes
 *   dx [N,HW,C] (+ dres if non-NULL, the residual-branch gradient), dgamma[C], dbeta[C].
 * act is MAS_ACT_AFFINE or MAS_ACT_AFFINE_SILU.                                       */
size_t mas_gn_bwd_workspace(int N, int C);
int    mas_gn_bwd(const void* x, const void* da, const void* dres, int dtype, int N, int HW, int C, int G,
                  int act, const float* gamma, const float* mean_rstd, const float* scale_shift,
                  void* dx, float* dgamma, float* dbeta, void* workspace, size_t ws_bytes, void* stream);
/* mas_gn_bwd takes the small-map kernel (bf16, h*w <= 512 pixels, see mas_gn_small_supported) where it applies and otherwise
 * mas_gn_bwd_3pass: reduce / finalize / apply as three launches (x and da are read twice; every dtype and shape), also callable
 * directly.  Same arguments, same workspace; bitwise reproducible run to run.  (Round 4's one-launch kernels lost to it on MI355X and
 * are shelved: docs/history/experiments/r4_gn_queue.patch, profiles/r04_gn_queue_v2.txt.)                                            */
int    mas_gn_bwd_3pass(const void* x, const void* da, const void* dres, int dtype, int N, int HW, int C, int G,
                        int act, const float* gamma, const float* mean_rstd, const float* scale_shift,
                        void* dx, float* dgamma, float* dbeta, void* workspace, size_t ws_bytes, void* stream);

/* ---- materialised GroupNorm(+SiLU) output: a [N,HW,C] = act(x * scale + shift), scale_shift [N][C][2] from mas_gn_stats, act
 * MAS_ACT_AFFINE or MAS_ACT_AFFINE_SILU, rounded to `dtype` exactly as the fused loaders of mas_conv_fwd / mas_conv_wgrad round it
 * (reference models/modules.py:121-128 as a tensor).  The optional alternative to the fused prologue: one read + one write, after
 * which the convolution and its weight gradient run prologue-free (act = NONE) on `a`.                                          */
int mas_gn_act(const void* x, void* a, int dtype, int N, int HW, int C, int act, const float* scale_shift, void* stream);

/* Small maps (bf16, h*w <= 1024 pixels, C % 64 == 0, whole groups inside a 64-channel block: mas_gn_small_supported != 0): the
 * statistics of mas_gn_stats AND the tensor of mas_gn_act in ONE launch -- a work-group keeps an (image, 64-channel block) slab in
 * registers between the reduction and the element-wise phase; three dependent launches become one.  mas_gn_bwd takes the matching
 * backward kernel for such tensors by itself (one launch + the batch sums for dgamma / dbeta).                                     */
int mas_gn_small_supported(int dtype, int HW, int C, int G);
int mas_gn_stats_act(const void* x, void* a, int dtype, int N, int HW, int C, int G, float eps, const float* gamma,
                     const float* beta, int act, float* mean_rstd, float* scale_shift, void* stream);

/* ---- convolution forward  (replaces F.conv2d at modules.py:49,68,93,100,113,145-160,
 * 219,236,345,364 and vqvae.py:15,18, with the GroupNorm-apply/SiLU of modules.py:121-128
 * fused into the input loader and bias / residual add (modules.py:136,191) into the epilogue).
 *   x         [N,H,W,Cin]     in_dtype
 *   scale_shift [N][Cin][2]   fp32, required iff act != NONE
 *   w_packed  from mas_pack_conv_weight (same dtype as x)
 *   bias      [Cout] fp32 or NULL
 *   residual  [N,Ho,Wo,Cout]  in_dtype or NULL (added in fp32 before the store)
 *   y         [N,Ho,Wo,Cout]  out_dtype
 * The data gradient of a stride-1 conv is this same entry point called on dy with the
 * transpose=1 packing and pad = ks-1-pad.                                             */
int mas_conv_fwd(const MasConvDesc* d, const void* x, const float* scale_shift, const void* w_packed,
                 const float* bias, const void* residual, void* y, void* stream);

/* Fused GroupNorm statistics: mas_conv_fwd_stats is mas_conv_fwd that ALSO writes, per output tile, the sum and the sum of squares
 * of every output channel (of the values as stored, i.e. after the bf16 rounding) into stats_partial [N][rows][Cout][2] fp32, where
 * rows = mas_conv_stat_rows(d) > 0 (0: this convolution's kernel has no fused statistics -- call mas_conv_fwd and mas_gn_stats).
 * mas_gn_stats_from_partials then replaces mas_gn_stats' pass over the tensor for the GroupNorm that consumes y (same outputs).
 * No atomics: the table is written once per tile and summed in a fixed order (bitwise run-to-run deterministic).                   */
int mas_conv_stat_rows(const MasConvDesc* d);
int mas_conv_fwd_stats(const MasConvDesc* d, const void* x, const float* scale_shift, const void* w_packed,
                       const float* bias, const void* residual, void* y, float* stats_partial, void* stream);
int mas_gn_stats_from_partials(const float* partial, int N, int HW, int C, int G, int rows, float eps, const float* gamma,
                               const float* beta, float* mean_rstd, float* scale_shift, void* stream);

/* ---- convolution weight gradient  (autograd of the F.conv2d sites above)
 *   dw [Cout][ks][ks][Cin] fp32 (caller zero-fills; accumulated with fp32 atomics),
 *   dbias [Cout] fp32 or NULL (same).  x / scale_shift / act as in mas_conv_fwd
 *   (the activated input is recomputed in the loader, never stored).                   */
int mas_conv_wgrad(const MasConvDesc* d, const void* x, const float* scale_shift, const void* dy,
                   float* dw, float* dbias, void* stream);
/* Commit of a weight gradient: acc = the zero-initialised accumulator handed to mas_conv_wgrad as dw ([Cout][ks][ks][Cin] fp32,
 * immediately followed by [Cout] bias sums when dbias != NULL) -> dw_oihw [Cout][Cin][ks][ks] (nn.Conv2d.weight.grad's layout),
 * dbias [Cout]; acc is zeroed again while it is read, so ONE scratch per stream serves every convolution of a step without fill
 * launches (the caller still owns it).                                                                                              */
int mas_wgrad_commit(float* acc, float* dw_oihw, float* dbias, int Cout, int Cin, int ks, void* stream);
/* Downsample's data gradient without the zero-stuffed tensor (reference models/modules.py:62-81 under autograd): d describes the
 * FORWARD convolution (3x3, stride 2, pads 0); w_packed_t = mas_pack_conv_weight_layout(transpose = 1, MAS_WLAYOUT_K64).           */
int mas_conv_s2_dgrad_supported(const MasConvDesc* d);
int mas_conv_s2_dgrad(const MasConvDesc* d, const void* dy, const void* w_packed_t, void* dx, void* stream);
/* Deterministic split-K (ABI v3): for the convolutions that carry the FLOPs (mas_conv_wgrad_splits(d) = nsplit > 0: bf16, stride 1, no
 * prologue; 3x3 with Cin % 64 == 0, Cout % 128 == 0, or 1x1 with Cin % 128 == 0, Cout % 128 == 0: part is then [nsplit][Cout][ks][ks][Cin]) mas_conv_wgrad_partial writes the nsplit partial sums -- part [nsplit][Cout][3][3][Cin] fp32,
 * part_bias [nsplit][Cout] or NULL, every element exactly once, plain stores, no initialisation required -- and mas_wgrad_reduce adds
 * the slabs in a fixed order into dw_oihw [Cout][Cin][ks][ks] / dbias [Cout]: no atomics, bitwise reproducible run to run.
 * Round 6 (no ABI change): every other geometry mas_conv_wgrad accepts (ks 1..4, stride 1 / 2, bf16 / fp32, prologue or not, Cin % 4 == 0)
 * reports nsplit > 0 too and runs the general kernels in slab mode; 0 is returned only for Cin % 4 != 0 (or MAS_WGRAD_GENERAL_SLABS=0),
 * and only then is mas_conv_wgrad's atomic commit the route.                                                                         */
int mas_conv_wgrad_splits(const MasConvDesc* d);
int mas_conv_wgrad_partial(const MasConvDesc* d, const void* x, const float* scale_shift, const void* dy,
                           float* part, float* part_bias, void* stream);
int mas_wgrad_reduce(const float* part, const float* part_bias, int nsplit, float* dw_oihw, float* dbias, int Cout, int Cin, int ks,
                     void* stream);

/* `Upsample` + convolution in its sub-pixel form (conv_up2.hip; MAS_WLAYOUT_UP2 above): 2.25x fewer FLOPs than the 3x3 convolution over
 * the x2 image.  d describes the FORWARD convolution as for mas_conv_fwd (upsample = 1, H x W the input map, Ho x Wo = 2H x 2W, 3x3,
 * stride 1, pads 1, bf16, no prologue).  Forward: mas_conv_up2_supported(d) != 0 -> pack the weight with MAS_WLAYOUT_UP2, set
 * d->w_layout to it and call mas_conv_fwd / mas_conv_fwd_stats (no residual).  Data gradient with respect to the LOW-resolution input:
 * dx [N,H,W,Cin] from dy [N,Ho,Wo,Cout] and w_packed_t = the MAS_WLAYOUT_UP2 image with transpose = 1 -- replaces the stride-1
 * data-gradient convolution at the high resolution plus the x2 sum-pooling pass.                                                   */
int mas_conv_up2_supported(const MasConvDesc* d);
int mas_conv_up2_dgrad_supported(const MasConvDesc* d);
int mas_conv_up2_dgrad(const MasConvDesc* d, const void* dy, const void* w_packed_t, void* dx, void* stream);
/* The weight gradient of the same layer in the same form (conv_wgrad_dma.hip with 2x2 taps per phase): mas_conv_up2_wgrad_splits(d) =
 * slabs PER PHASE (0: unsupported, take mas_conv_wgrad_partial); mas_conv_up2_wgrad_partial writes part [4][nsplit][Cout][2][2][Cin] fp32
 * and (when non-NULL) part_bias [4 * nsplit][Cout], every element once, plain stores; mas_wgrad_reduce_up2 adds the slabs in a fixed
 * order and folds the 4 x 4 phase taps into dw_oihw [Cout][Cin][3][3] / dbias [Cout] (bitwise reproducible run to run).              */
int mas_conv_up2_wgrad_splits(const MasConvDesc* d);
int mas_conv_up2_wgrad_partial(const MasConvDesc* d, const void* x, const void* dy, float* part, float* part_bias, void* stream);
int mas_wgrad_reduce_up2(const float* part, const float* part_bias, int nsplit, float* dw_oihw, float* dbias, int Cout, int Cin, void* stream);

/* ---- vector quantiser  (replaces Codebook.forward's distance / argmin / gather / loss,
 * modules.py:501-509; never materialises d[M,K]).
 *   z [M][D] fp32 (NHWC latent rows), codebook [K][D] fp32.
 *   idx [M] int64 : argmin_k ( (|z|^2+|e_k|^2) - 2 z.e_k ), first minimum on ties
 *   zq  [M][D] fp32 = codebook[idx]
 *   sqerr [1] fp32 = sum (zq - z)^2      (loss = (1+beta) * sqerr / (M*D), modules.py:509)
 * workspace: mas_vq_workspace(M,K) bytes.                                             */
size_t mas_vq_workspace(int M, int K);
int    mas_vq_argmin_fwd(const float* z, const float* codebook, int M, int K, int D,
                         int64_t* idx, float* zq, float* sqerr, void* workspace, size_t ws_bytes, void* stream);
/* backward of (z_q straight-through, loss):  dz = g_zq + g_loss*2/(M*D)*(z-e[idx]);
 * dcodebook[idx] += g_loss*beta*2/(M*D)*(e[idx]-z).  For D <= 256, D % 4 == 0 (every codebook_dim of mas_vq_argmin_fwd) EVERY row of
 * dcodebook is written, summed in a fixed order (no atomics: bitwise run-to-run deterministic, no zero fill needed); other D add
 * with fp32 atomics into a buffer the caller zero-filled.  g_loss is a device scalar.                                            */
int    mas_vq_bwd(const float* z, const float* codebook, const int64_t* idx, const float* g_zq,
                  const float* g_loss, float beta, int M, int K, int D, float* dz, float* dcodebook, void* stream);

/* ---- causal multi-head self-attention forward (flash style)  (replaces calculate_attention + Softmax +
 * matmul(probs, v), models/transformer.py:44-71,90-97, training configuration: pb-relax shift is softmax-invariant,
 * the net mask is pure causal -- SURVEY 3.4).  q,k,v: element (b, s, h, d) at ptr[b*bs + s*ld + h*hd + d] (so the
 * fused [B,S,3*H*hd] qkv tensor is addressed in place); o: [B,S,H*hd] contiguous; lse: [B,H,S] fp32 (log-sum-exp of
 * the scaled scores, for the backward) or NULL.  hd in {16,32,64,128}.                                          */
int mas_attn_causal_fwd(const void* q, const void* k, const void* v, void* o, float* lse, int dtype, int B, int H,
                        int S, int hd, int ld_q, int ld_k, int ld_v, long long q_bs, long long k_bs, long long v_bs,
                        float scale, void* stream);

/* backward of mas_attn_causal_fwd on the fused projection: qkv and dqkv are [B,S,3*H*hd] contiguous (q|k|v stacked on
 * the last axis), o / dout [B,S,H*hd], lse from the forward, delta [B,H,S] fp32 scratch.  Every element of dqkv is
 * written exactly once (no atomics, deterministic).                                                             */
int mas_attn_causal_bwd(const void* qkv, const void* o, const void* dout, const float* lse, float* delta, void* dqkv,
                        int dtype, int B, int H, int S, int hd, float scale, void* stream);

/* ---- single-head spatial self-attention core of AttnBlock  (replaces the two torch.bmm + softmax of models/modules.py:174-187 and
 * their autograd).  qkv: [N, S, 3C] bf16, q | k | v stacked on the channel axis (the fused 1x1 projection, NHWC with S = h*w);
 * out [N, S, C] = softmax_keys(q k^T * C^-1/2) v;  lse [N, S] fp32 (log-sum-exp of the scaled scores; may be NULL when no backward
 * follows).  Backward: dout [N, S, C] -> dqkv [N, S, 3C] (every element written exactly once, no atomics); delta [N, S] fp32 scratch.
 * bf16 only, S <= 256 tokens, C <= 512 and C % 32 == 0 (the reference's blocks: 16x16x512, 8x8x512).                                */
int mas_spatial_attn_fwd(const void* qkv, void* out, float* lse, int dtype, int N, int S, int C, void* stream);
int mas_spatial_attn_bwd(const void* qkv, const void* dout, const float* lse, float* delta, void* dqkv, int dtype, int N, int S, int C,
                         void* stream);

/* ---- decode-time (KV-cached) attention  (replaces the cached branch of SelfAttention.forward, models/transformer.py:73-115,
 * for token-by-token sampling: SURVEY 8(f) rank 3).  nq new queries of every (batch, head) against a cache of past + nq keys /
 * values: query i (0 <= i < nq) attends to keys 0 .. past + i (causal inside the block).  q: element (b, i, h, d) at
 * q[b*q_bs + i*ld_q + h*hd + d]; k_cache / v_cache: element (b, s, h, d) at ptr[b*bs + s*ld + h*hd + d], rows 0 .. past+nq-1
 * valid (the caller appends the new rows BEFORE the call); o likewise with (o_bs, ld_o).  HBM-bound: each key / value row is
 * read once per query; no [nq, S] score tensor exists.  hd in {16,32,64,128}; rows 16-byte aligned.                        */
int mas_attn_decode(const void* q, const void* k_cache, const void* v_cache, void* o, int dtype, int B, int H, int nq,
                    int past, int hd, int ld_q, int ld_k, int ld_v, int ld_o, long long q_bs, long long k_bs,
                    long long v_bs, long long o_bs, float scale, void* stream);

/* ---- small NHWC helpers on the path -------------------------------------------------
 * nearest x2 upsample (F.interpolate, modules.py:56) and its adjoint (2x2 sum);
 * zero-stuffing used by the stride-2 data gradient (adjoint of modules.py:76-78).      */
int mas_upsample2x(const void* x, void* y, int dtype, int N, int H, int W, int C, void* stream);
int mas_sumpool2x(const void* x, void* y, int dtype, int N, int Ho, int Wo, int C, void* stream);
int mas_zero_stuff2x(const void* x, void* y, int dtype, int N, int H, int W, int C, int Hout, int Wout, void* stream);
/* y[n][h'][w'][(dy*2+dx)*C + c] = x[n][2h'+dy-pad][2w'+dx-pad][c] (0 outside), y is [N,Ho,Wo,4C]: a stride-2 K x K convolution of x
 * equals the stride-1 (K/2) x (K/2) convolution of y -- the weight gradient of the discriminator's 4x4 stride-2 convolutions
 * (losses/discriminator.py:20,27) runs as mas_conv_wgrad with ks = 2 on y.                                                       */
int mas_space_to_depth2x(const void* x, void* y, int dtype, int N, int H, int W, int C, int Ho, int Wo, int pad, void* stream);

/* -------------------------------------------------------------------------------------------
 * Transformer row operators (streaming, HBM-bound).
 * mas_gelu_tanh_*: OpenAI tanh-GELU of reference models/transformer.py:11-14 (MLP.forward :129) over n elements.
 * mas_layernorm_*: torch.nn.LayerNorm(D, eps) as used four times per TransformerLayer
 *   (models/transformer.py:159-163,197-210), rows x D, with an optional fused residual  y = residual + LN(x)
 *   (residual / y / dy have out_dtype; x / dx have in_dtype; gamma, beta, dgamma, dbeta fp32).
 *   mean_rstd [rows][2] fp32 is written by the forward (may be NULL when no backward follows) and read by the
 *   backward.  D must be a multiple of the 16-byte vector (8 for bf16->bf16, else 4) and at most 256 vectors.      */
int mas_gelu_tanh_fwd(const void* x, void* y, int dtype, long long n, void* stream);
int mas_gelu_tanh_bwd(const void* x, const void* dy, void* dx, int dtype, long long n, void* stream);
/* mas_gelu_tanh_bwd_colsum (ABI v9): mas_gelu_tanh_bwd over a [rows][cols] tensor that ALSO returns dx_colsum[cols] (fp32) = the column sums of
 *   the dx it writes (as stored; fixed summation order): the bias gradient of the Linear layer that produced x (`lin1`, reference
 *   models/transformer.py:125,129: grad_bias = grad_output.sum(0) with grad_output = this dx), otherwise a mas_colsum pass over dx.
 *   cols % 8 == 0 (bf16) / % 4 (fp32); workspace: mas_gelu_tanh_bwd_colsum_workspace bytes (0 = shape not supported), caller-owned.                */
size_t mas_gelu_tanh_bwd_colsum_workspace(int dtype, int cols);
int mas_gelu_tanh_bwd_colsum(const void* x, const void* dy, void* dx, float* dx_colsum, int dtype, long long rows, int cols, void* workspace,
                             size_t workspace_bytes, void* stream);
int mas_layernorm_fwd(const void* x, const float* gamma, const float* beta, const void* residual, void* y,
                      float* mean_rstd, int in_dtype, int out_dtype, int rows, int D, float eps, void* stream);
size_t mas_layernorm_bwd_workspace(int rows, int D);
int mas_layernorm_bwd(const void* x, const void* dy, const float* gamma, const float* mean_rstd, void* dx,
                      float* dgamma, float* dbeta, int in_dtype, int out_dtype, int rows, int D,
                      void* workspace, size_t workspace_bytes, void* stream);

/* mas_layernorm_bwd_add: mas_layernorm_bwd with dx = (LayerNorm input gradient) + dx_add -- dx_add [rows][D] in in_dtype is the
 *   gradient that reached x along its skip connection (x feeds the pre-LayerNorm AND the residual of the block's sandwich
 *   LayerNorm, reference models/transformer.py:197-210): one pass instead of the backward plus a separate add.  dx_add NULL =
 *   mas_layernorm_bwd; dx_add may alias dx.                                                                                  */
int mas_layernorm_bwd_add(const void* x, const void* dy, const float* gamma, const float* mean_rstd, const void* dx_add, void* dx,
                          float* dgamma, float* dbeta, int in_dtype, int out_dtype, int rows, int D,
                          void* workspace, size_t workspace_bytes, void* stream);

/* mas_layernorm_bwd_colsum: mas_layernorm_bwd_add that ALSO returns dx_colsum[D] (fp32) = the column sums of dx as stored (rounded to
 *   in_dtype; fixed summation order) -- the bias gradient of the Linear layer whose output this LayerNorm normalises (out_proj / lin2
 *   in front of the sandwich LayerNorms, reference models/transformer.py:201-203,207-209: grad_bias = grad_output.sum(0) with
 *   grad_output = this dx), which otherwise costs a mas_colsum pass over dx.  dx_colsum NULL = mas_layernorm_bwd_add.             */
int mas_layernorm_bwd_colsum(const void* x, const void* dy, const float* gamma, const float* mean_rstd, const void* dx_add, void* dx,
                             float* dgamma, float* dbeta, float* dx_colsum, int in_dtype, int out_dtype, int rows, int D,
                             void* workspace, size_t workspace_bytes, void* stream);

/* mas_layernorm_pair_* (ABI v9): the sandwich LayerNorm + residual of one sub-block and the pre-LayerNorm of the next as ONE pass,
 *   xnew = residual + LN1(h),  y2 = LN2(xnew)      (reference models/transformer.py:201-203 + :205, and :207-209 + :197 of the next layer
 *   or the final LayerNorm :264) -- the row stays in registers between the two: 12 B per element instead of 16, bit for bit the values
 *   of the two separate launches.  h / dh in h_dtype, residual / xnew / dres / dskip in x_dtype, y2 / dy2 in y_dtype:
 *   (bf16, fp32, bf16) -- the autocast transformer with its fp32 residual stream -- or all fp32.  D % 4 == 0, D <= 1024.
 *   The backward is mas_layernorm_bwd_add on (xnew, dy2, dskip) followed by mas_layernorm_bwd_colsum on (h, its result): a fused form was
 *   built and lost (two waves per SIMD at 190 VGPRs: 80 us against 34 + 25).                                                         */
int mas_layernorm_pair_fwd(const void* h, const float* gamma1, const float* beta1, const void* residual, const float* gamma2,
                           const float* beta2, void* xnew, void* y2, float* mean_rstd1, float* mean_rstd2, int h_dtype, int x_dtype,
                           int y_dtype, int rows, int D, float eps1, float eps2, void* stream);

/* mas_colsum: out[c] = sum over rows of x[r][c] (fp32 accumulation, fixed summation order: bitwise run-to-run deterministic).
 *   The bias gradient of the transformer's Linear layers (torch.nn.Linear in reference models/transformer.py:31,34,125,126:
 *   grad_bias = grad_output.sum(0)) over [B*S, N] activations.  x bf16 or fp32, row stride = cols, cols % 8 == 0 (bf16) / % 4 (fp32).
 *   workspace: mas_colsum_workspace(rows, cols) bytes, caller-owned.                                                          */
size_t mas_colsum_workspace(int rows, int cols);
int mas_colsum(const void* x, int dtype, int rows, int cols, float* out, void* workspace, size_t workspace_bytes, void* stream);

/* -------------------------------------------------------------------------------------------
 * BatchNorm over [M][C] fp32 NHWC activations (ABI v8): the nn.SyncBatchNorm(embed_dim) behind quant_conv, reference models/vqvae.py:15-16
 * (torch.nn.SyncBatchNorm: batch statistics exchanged across the process group in training, running statistics in evaluation).
 * The library computes per-rank sums in a fixed order and takes GLOBAL sums back; the exchange itself (torch.distributed.all_reduce of
 * the fp64 vector) is the host side's, exactly where torch's own SyncBatchNorm has it.  C % 4 == 0, C <= 1024.
 *   mas_bn_partial_sums: sums[2 C + 1] (fp64) = { per-channel S1[C], S2[C], (double) M }:
 *       dy == NULL : S1 = sum x,  S2 = sum x^2                      (forward statistics)
 *       dy != NULL : S1 = sum dy, S2 = sum dy * (x - mean) * rstd   (backward; = dbeta, dgamma of this rank); mean_rstd [C][2] required
 *   mas_bn_finalize   : from (global) sums: mean_rstd [C][2], scale_shift [C][2] (y = x * scale + shift with gamma / beta folded in;
 *       gamma / beta NULL = 1 / 0) and the running statistics updated in place with `momentum` (unbiased variance, torch's convention;
 *       running_* may be NULL); sums == NULL: evaluation mode, the pair comes from running_mean / running_var.
 *   mas_bn_apply      : y = x * scale + shift.
 *   mas_bn_bwd_apply  : dx = gamma * rstd * (dy - S1 / n - xhat * S2 / n) with the GLOBAL backward sums and n = sums[2 C].        */
size_t mas_bn_workspace(int M, int C);
int mas_bn_partial_sums(const float* x, const float* dy, const float* mean_rstd, int M, int C, double* sums, void* workspace,
                        size_t workspace_bytes, void* stream);
int mas_bn_finalize(const double* sums, const float* gamma, const float* beta, float eps, float momentum, float* running_mean,
                    float* running_var, float* mean_rstd, float* scale_shift, int C, void* stream);
int mas_bn_apply(const float* x, const float* scale_shift, float* y, int M, int C, void* stream);
int mas_bn_bwd_apply(const float* x, const float* dy, const float* mean_rstd, const float* gamma, const double* sums, float* dx, int M,
                     int C, void* stream);
/* ABI v9: the same three passes on bf16 or fp32 storage (dtype = MAS_BF16 / MAS_F32 for x, dy, y, dx alike) with the LeakyReLU(slope) that
 * follows the normalisation fused in -- the nn.BatchNorm2d + nn.LeakyReLU(0.2) pairs of the PatchGAN discriminator, reference
 * losses/discriminator.py:26-33 (per-rank batch statistics there: no exchange).  slope == 1: no activation (the v8 entry points above
 * are these with MAS_F32 and slope 1).
 *   mas_bn_apply_act        : y = lrelu(x * scale + shift).
 *   mas_bn_partial_sums_act : backward sums of g = dy * lrelu'(u), u = x * scale + shift recomputed (scale_shift required when slope != 1).
 *   mas_bn_bwd_apply_act    : dx = gamma * rstd * (g - S1 / n - xhat * S2 / n) with the same g.                                      */
int mas_bn_partial_sums_act(const void* x, const void* dy, const float* mean_rstd, const float* scale_shift, float slope, int dtype, int M, int C,
                            double* sums, void* workspace, size_t workspace_bytes, void* stream);
int mas_bn_apply_act(const void* x, const float* scale_shift, void* y, float slope, int dtype, int M, int C, void* stream);
int mas_bn_bwd_apply_act(const void* x, const void* dy, const float* mean_rstd, const float* gamma, const float* scale_shift, float slope,
                         const double* sums, void* dx, int dtype, int M, int C, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MAS_HIP_H */
