"""MI355X-native VQGAN building blocks behind the reference's class surface.

Same import path, constructor kwargs, attribute names and ``state_dict`` keys as the
reference's ``models/modules.py`` (Normalize :40, Upsample :44, Downsample :62, ResnetBlock :84,
AttnBlock :139, Swish :194, Encoder :199, Decoder :337, Codebook :451) so ``conf/*.yaml`` and
``train.py`` instantiate these classes unchanged.  The arithmetic runs in hand-written HIP
kernels (libmas_hip.so) on NHWC bf16 (or fp32) activations:

* GroupNorm: the statistics come out of the PRODUCING convolution's epilogue (or one launch on small maps), one
  streaming pass writes silu(gn(x)) as a tensor and the 3x3 convolution AND its weight gradient run prologue-free on
  it (measured faster on MI355X than the fused loader, which is kept: ``MAS_GN_MATERIALIZE=0``);
* the residual add rides the conv epilogue; ``Upsample`` + conv runs in its sub-pixel form (four 2x2 phase
  convolutions on the low-resolution map, conv_up2.hip) and the one-sided Downsample padding is address
  arithmetic inside the conv kernel;
* the codebook lookup never materialises the [B*h*w, n_embed] distance matrix.

Parameters stay plain fp32 OIHW ``nn.Parameter``s (``change_requires_grad`` in the reference's
``utils.py:27-29`` flips ``requires_grad`` every step; DDP hooks see ordinary ``.grad``).
"""
from __future__ import annotations

import os

import torch
import torch.nn as nn
import torch.nn.functional as F
import torch.distributed as dist

from mas_hip import ACT_AFFINE, ACT_AFFINE_SILU, ACT_NONE
from mas_hip import ops

# SyncBatchNorm: 2 (default since round 6) = batchnorm.hip everywhere, WITH the cross-rank exchange when the group has more than one rank (one
# all_reduce of 2 C + 1 fp64 sums each way: tested at two ranks over gloo against the full batch, and over RCCL itself at world size 1 with the
# exchange forced -- tests/test_gpu_bn.py; no multi-GPU node exists here, so RCCL has not carried it between two devices yet);
# 1 = batchnorm.hip only where no exchange is needed, torch's own module for the exchanging case (the default of round 5);
# 0 = torch's implementation everywhere.  MAS_SYNCBN_EXCHANGE_AT_WORLD_1=1 (tests): a one-rank group exchanges too.
_MAS_SYNCBN = int(os.environ.get("MAS_SYNCBN", "2"))
_EXCHANGE_AT_WORLD_1 = os.environ.get("MAS_SYNCBN_EXCHANGE_AT_WORLD_1", "0") == "1"
_RESERVOIR_ASYNC = os.environ.get("MAS_RESERVOIR_ASYNC", "1") == "1"          # 0: the codebook's reservoir permutations reach the GPU by a blocking copy


def nonlinearity(x):
    """swish (reference modules.py:35-37); stand-alone use only -- on the hot path it is fused."""
    return x * torch.sigmoid(x)


class _GroupNorm(nn.GroupNorm):
    """GroupNorm(32, C, eps=1e-6).  Inside Encoder/Decoder/ResnetBlock/AttnBlock its parameters are
    consumed by the fused conv kernels; calling the module on its own (off the hot path) uses ATen."""


def Normalize(in_channels):
    return _GroupNorm(num_groups=32, num_channels=in_channels, eps=1e-6, affine=True)


class SyncBatchNorm(nn.SyncBatchNorm):
    """``nn.SyncBatchNorm`` parameters, buffers and ``state_dict`` keys (reference models/vqvae.py:16: the normalisation behind
    ``quant_conv``) on ``libmas_hip.so``'s BatchNorm kernels (``batchnorm.hip``) for a 4-D fp32 CUDA input: per-rank sums in a fixed order,
    running statistics as torch keeps them.  With more than one rank in ``process_group`` the training forward exchanges the per-rank
    sums: ONE all_reduce of 2 C + 1 fp64 values each way around the same kernels (the default since round 6; ``MAS_SYNCBN=1`` hands that
    case to torch's own implementation, as round 5 did).  ``MAS_SYNCBN=0`` (or any other input) falls through to torch's
    implementation everywhere."""

    def forward(self, x):
        if not (_MAS_SYNCBN and x.is_cuda and x.dim() == 4 and x.dtype == torch.float32 and x.shape[1] % 4 == 0 and x.shape[1] <= 1024
                and (self.training or self.track_running_stats)):
            return super().forward(x)
        training = self.training or not self.track_running_stats
        group = None
        if training and torch.distributed.is_available() and torch.distributed.is_initialized():
            pg = self.process_group if self.process_group is not None else torch.distributed.group.WORLD
            if torch.distributed.get_world_size(pg) > 1 or _EXCHANGE_AT_WORLD_1:
                if _MAS_SYNCBN < 2:
                    return super().forward(x)               # the exchanging case: torch's own module unless MAS_SYNCBN=2 (see above)
                group = pg
        if training and group is None and x.numel() // x.shape[1] == 1:
            raise ValueError(f"Expected more than 1 value per channel when training, got input size {x.shape}")      # (torch's own check)
        momentum = 0.0 if self.momentum is None else self.momentum
        if training and self.track_running_stats and self.num_batches_tracked is not None:
            self.num_batches_tracked.add_(1)
            if self.momentum is None:                       # cumulative moving average (torch's convention; costs a host read)
                momentum = 1.0 / float(self.num_batches_tracked)
        rm = self.running_mean if self.track_running_stats else None
        rv = self.running_var if self.track_running_stats else None
        return ops.sync_batch_norm(x, self.weight, self.bias, rm, rv, self.eps, momentum, training, group)


class Conv2d(nn.Conv2d):
    """nn.Conv2d parameters (same default init, same state_dict keys) + the HIP implicit-GEMM kernel.
    ``in_dtype`` / ``out_dtype`` (None = the global compute dtype) pin the precision of one layer:
    the latent tail (encoder's last conv, quant_conv, post_quant_conv) runs with fp32 storage so
    codebook indices are computed from fp32 latents."""

    in_dtype = None
    out_dtype = None
    pad4 = None          # (top, bottom, left, right); None -> symmetric self.padding

    def _pad4(self):
        if self.pad4 is not None:
            return self.pad4
        ph, pw = self.padding
        return (ph, ph, pw, pw)

    # The packed bf16/NHWC weight images are cached on ops._param_stamp (Parameter._version, data_ptr, optimizer steps -- fused
    # optimizers do not bump _version, a global optimizer post-step hook covers them); writes THROUGH ``.data`` change
    # neither.  The two places such writes normally surround -- loading a checkpoint and switching train()/eval() (EMA
    # swap-in) -- drop this layer's images; any other ``.data`` write needs ``mas_hip.ops.invalidate_weight_cache()``.
    def _load_from_state_dict(self, *args, **kwargs):
        super()._load_from_state_dict(*args, **kwargs)
        ops.drop_weight_cache_of(self.weight, self.bias)

    def train(self, mode: bool = True):
        if mode != self.training:
            ops.drop_weight_cache_of(self.weight, self.bias)       # this layer's images only (not the process-wide cache per child)
        return super().train(mode)

    def forward(self, x, residual=None, upsample=False):
        if self.kernel_size[0] != self.kernel_size[1] or self.kernel_size[0] not in (1, 3, 4) or self.stride[0] not in (1, 2) \
                or self.dilation != (1, 1) or self.groups != 1:
            raise NotImplementedError("libmas_hip conv supports 1x1 / 3x3 (and, bf16, 4x4), stride 1 / 2, dense, undilated")
        return ops.norm_act_conv(x, self.weight, self.bias, None, None, residual, stride=self.stride[0], padding=self._pad4(),
                                 act=ACT_NONE, upsample=upsample, in_dtype=self.in_dtype, out_dtype=self.out_dtype)

    def fused(self, x, norm, act, residual=None):
        """conv(act(norm(x))) (+residual) with the norm/act folded into this conv's loader."""
        return ops.norm_act_conv(x, self.weight, self.bias, norm.weight, norm.bias, residual, stride=self.stride[0],
                                 padding=self._pad4(), act=act, groups=norm.num_groups, eps=norm.eps,
                                 in_dtype=self.in_dtype, out_dtype=self.out_dtype)


class Upsample(nn.Module):
    """reference modules.py:44-59: nearest x2, then 3x3 conv.  The 4x tensor is never written: maps at least 32
    pixels wide take the sub-pixel form (four 2x2 phase convolutions on the low-resolution map, 2.25x fewer
    FLOPs: conv_up2.hip), narrower ones fold the x2 into the 3x3 kernel's address arithmetic."""

    def __init__(self, in_channels, with_conv):
        super().__init__()
        self.with_conv = with_conv
        if self.with_conv:
            self.conv = Conv2d(in_channels, in_channels, kernel_size=3, stride=1, padding=1)

    def forward(self, x):
        if self.with_conv:
            return self.conv(x, upsample=True)
        return F.interpolate(x, scale_factor=2.0, mode="nearest")   # never taken by the reference's configs


class Downsample(nn.Module):
    """reference modules.py:62-81: zero-pad right/bottom by one, 3x3 stride-2 conv; the pad is a
    bounds check in the kernel, no padded copy."""

    def __init__(self, in_channels, with_conv):
        super().__init__()
        self.with_conv = with_conv
        if self.with_conv:
            self.conv = Conv2d(in_channels, in_channels, kernel_size=3, stride=2, padding=0)
            self.conv.pad4 = (0, 1, 0, 1)

    def forward(self, x):
        if self.with_conv:
            return self.conv(x)
        return F.avg_pool2d(x, kernel_size=2, stride=2)             # never taken by the reference's configs


class ResnetBlock(nn.Module):
    """reference modules.py:84-136:  x + conv2(silu(gn2(conv1(silu(gn1(x))))))  (1x1 shortcut when
    the channel count changes) as ONE autograd node (``ops.resblock``): GroupNorm statistics from the
    producer's epilogue, silu(gn(.)) written once per convolution, the residual add in conv2's epilogue,
    the skip gradient added inside norm1's backward pass."""

    def __init__(self, *, in_channels, out_channels=None, conv_shortcut=False, dropout):
        super().__init__()
        self.in_channels = in_channels
        out_channels = in_channels if out_channels is None else out_channels
        self.out_channels = out_channels
        self.use_conv_shortcut = conv_shortcut
        self.norm1 = Normalize(in_channels)
        self.conv1 = Conv2d(in_channels, out_channels, kernel_size=3, stride=1, padding=1)
        self.norm2 = Normalize(out_channels)
        self.dropout = torch.nn.Dropout(dropout)
        self.conv2 = Conv2d(out_channels, out_channels, kernel_size=3, stride=1, padding=1)
        if self.in_channels != self.out_channels:
            if self.use_conv_shortcut:
                self.conv_shortcut = Conv2d(in_channels, out_channels, kernel_size=3, stride=1, padding=1)
            else:
                self.nin_shortcut = Conv2d(in_channels, out_channels, kernel_size=1, stride=1, padding=0)

    def forward(self, x):
        if self.training and self.dropout.p > 0:
            raise NotImplementedError("dropout > 0 is off the hot path (the reference always builds blocks with dropout=0.0)")
        plain = self.conv1.in_dtype is None and self.conv2.out_dtype is None and self.conv1.bias is not None and self.conv2.bias is not None
        if plain and self.in_channels == self.out_channels:
            return ops.resblock(x, self.norm1, self.conv1, self.norm2, self.conv2)     # one autograd node (fused skip gradient)
        if plain and not self.use_conv_shortcut and self.nin_shortcut.bias is not None and self.nin_shortcut.in_dtype is None \
                and self.nin_shortcut.out_dtype is None and self.in_channels % 64 == 0 and self.out_channels % 128 == 0:
            # (round 4) the 1x1 shortcut inside the same node: its data gradient reaches norm1's backward as `dres` instead of
            # through a separate add over the block's input (4 blocks of VQ-IMG, 0.25 ms of element-wise adds per step)
            return ops.resblock(x, self.norm1, self.conv1, self.norm2, self.conv2, self.nin_shortcut)
        h = self.conv1.fused(x, self.norm1, ACT_AFFINE_SILU)
        if self.in_channels != self.out_channels:
            x = self.conv_shortcut(x) if self.use_conv_shortcut else self.nin_shortcut(x)
        return self.conv2.fused(h, self.norm2, ACT_AFFINE_SILU, residual=x)


class AttnBlock(nn.Module):
    """reference modules.py:139-191: single-head spatial self-attention over h*w tokens.
    GN is folded into ONE fused q|k|v 1x1 conv (Cout = 3C); proj_out carries the residual add."""

    def __init__(self, in_channels):
        super().__init__()
        self.in_channels = in_channels
        self.norm = Normalize(in_channels)
        self.q = Conv2d(in_channels, in_channels, kernel_size=1, stride=1, padding=0)
        self.k = Conv2d(in_channels, in_channels, kernel_size=1, stride=1, padding=0)
        self.v = Conv2d(in_channels, in_channels, kernel_size=1, stride=1, padding=0)
        self.proj_out = Conv2d(in_channels, in_channels, kernel_size=1, stride=1, padding=0)

    def forward(self, x):
        c = self.in_channels
        w_qkv = torch.cat([self.q.weight, self.k.weight, self.v.weight], dim=0)
        w_qkv._mas_sources = (self.q.weight, self.k.weight, self.v.weight)     # its packed images are cached on these three stamps
        b_qkv = torch.cat([self.q.bias, self.k.bias, self.v.bias], dim=0)
        qkv = ops.norm_act_conv(x, w_qkv, b_qkv, self.norm.weight, self.norm.bias, None, stride=1, padding=(0, 0, 0, 0),
                                act=ACT_AFFINE, groups=self.norm.num_groups, eps=self.norm.eps)
        h_ = ops.spatial_attention(qkv, c)
        return self.proj_out(h_, residual=x)


class Swish(nn.Module):
    def forward(self, x):
        return x * torch.sigmoid(x)


def _run_layers(layers, x):
    """nn.Sequential semantics, except that a trailing [GroupNorm, Swish, Conv2d] triple
    (reference modules.py:234-236, 362-364) is executed as one fused kernel."""
    layers = list(layers)
    i = 0
    while i < len(layers):
        m = layers[i]
        if isinstance(m, _GroupNorm) and i + 2 < len(layers) + 0 and isinstance(layers[i + 1], Swish) and isinstance(layers[i + 2], Conv2d):
            x = layers[i + 2].fused(x, m, ACT_AFFINE_SILU)
            i += 3
        else:
            x = m(x)
            i += 1
    return x


class Encoder(nn.Module):
    """reference modules.py:199-240.  [B,in_channels,H,W] fp32 -> [B,z_channels,H/2^k,W/2^k] fp32
    (the last conv stores fp32 so the latent tail is not rounded to bf16)."""

    def __init__(self, in_channels=3, channels=[128, 128, 128, 256, 512, 512], attn_resolutions=[32], resolution=512, dropout=0.0,
                 num_res_blocks=2, z_channels=256, **kwargs):
        super(Encoder, self).__init__()
        layers = [Conv2d(in_channels, channels[0], 3, 1, 1)]
        for i in range(len(channels) - 1):
            in_channels = channels[i]
            out_channels = channels[i + 1]
            for j in range(num_res_blocks):
                layers.append(ResnetBlock(in_channels=in_channels, out_channels=out_channels, dropout=0.0))
                in_channels = out_channels
                if resolution in attn_resolutions:
                    layers.append(AttnBlock(in_channels))
            if i < len(channels) - 2:
                layers.append(Downsample(channels[i + 1], with_conv=True))
                resolution //= 2
        layers.append(ResnetBlock(in_channels=channels[-1], out_channels=channels[-1], dropout=0.0))
        layers.append(AttnBlock(channels[-1]))
        layers.append(ResnetBlock(in_channels=channels[-1], out_channels=channels[-1], dropout=0.0))
        layers.append(Normalize(channels[-1]))
        layers.append(Swish())
        layers.append(Conv2d(channels[-1], z_channels, 3, 1, 1))
        layers[-1].out_dtype = torch.float32
        self.model = nn.Sequential(*layers)

    def forward(self, x):
        return _run_layers(self.model, x)


class Decoder(nn.Module):
    """reference modules.py:337-369.  The output conv (model[-1], whose ``.weight`` the VQGAN loss
    differentiates against, train.py:96) stores fp32."""

    def __init__(self, out_channels=3, channels=[128, 128, 128, 256, 512, 512], attn_resolutions=[32], resolution=512, dropout=0.0,
                 num_res_blocks=2, z_channels=256, **kwargs):
        super(Decoder, self).__init__()
        ch_mult = channels[1:]
        num_resolutions = len(ch_mult)
        block_in = ch_mult[num_resolutions - 1]
        curr_res = resolution // 2 ** (num_resolutions - 1)
        layers = [Conv2d(z_channels, block_in, kernel_size=3, stride=1, padding=1),
                  ResnetBlock(in_channels=block_in, out_channels=block_in, dropout=0.0),
                  AttnBlock(block_in),
                  ResnetBlock(in_channels=block_in, out_channels=block_in, dropout=0.0)]
        for i in reversed(range(num_resolutions)):
            block_out = ch_mult[i]
            for i_block in range(num_res_blocks + 1):
                layers.append(ResnetBlock(in_channels=block_in, out_channels=block_out, dropout=0.0))
                block_in = block_out
                if curr_res in attn_resolutions:
                    layers.append(AttnBlock(block_in))
            if i > 0:
                layers.append(Upsample(block_in, with_conv=True))
            curr_res = curr_res * 2
        layers.append(Normalize(block_in))
        layers.append(Swish())
        layers.append(Conv2d(block_in, out_channels, kernel_size=3, stride=1, padding=1))
        layers[-1].out_dtype = torch.float32
        self.model = nn.Sequential(*layers)

    def forward(self, x):
        return _run_layers(self.model, x)


class Codebook(nn.Module):
    """reference modules.py:451-528.  Same warm-up schedule (q_counter / reservoir are plain
    attributes, not in the state_dict); the steady-state lookup is the fused HIP kernel."""

    def __init__(self, codebook_size, codebook_dim, beta, init_steps=2000, reservoir_size=2e5):
        super().__init__()
        self.codebook_size = codebook_size
        self.codebook_dim = codebook_dim
        self.beta = beta
        self.embedding = nn.Embedding(self.codebook_size, self.codebook_dim)
        self.embedding.weight.data.uniform_(-1.0 / self.codebook_size, 1.0 / self.codebook_size)
        self.q_start_collect, self.q_init, self.q_re_end, self.q_re_step = init_steps, init_steps * 3, init_steps * 30, init_steps // 2
        self.q_counter = 0
        self.reservoir_size = int(reservoir_size)
        self.reservoir = None

    @staticmethod
    def _randperm(n, device, keep=None):
        """``torch.randperm(n)[:keep]`` as the reference draws it -- on the CPU, from the default generator (modules.py:479,481) -- delivered to
        ``device`` WITHOUT a host synchronisation: through a pinned staging tensor and a non-blocking copy (the caching host allocator keeps
        the block until the copy has run).  The reference indexes a CUDA tensor with the CPU permutation, ``torch.randperm(n, device="cuda")``
        copies its small-n CPU result with a blocking copy: either way the host waits for the whole encoder forward it has queued, twice per
        step, and the short kernels that follow (the decoder's 16 x 16 layers) run host-bound (profiles/r06_host_sync.txt)."""
        p = torch.randperm(n)
        if keep is not None:
            p = p[:keep]
        if device.type != "cuda" or not _RESERVOIR_ASYNC:
            return p.to(device)
        pinned = torch.empty(p.numel(), dtype=p.dtype, pin_memory=True)
        pinned.copy_(p)
        return pinned.to(device, non_blocking=True)

    def _collect(self, z_flat, batch_size):
        # reservoir sampling of 10 latents / image (reference modules.py:477-481)
        z_new = z_flat.detach().reshape(batch_size, -1, self.codebook_dim)
        z_new = z_new[:, self._randperm(z_new.size(1), z_new.device)][:, :10].reshape(-1, self.codebook_dim)
        self.reservoir = z_new if self.reservoir is None else torch.cat([self.reservoir, z_new], dim=0)
        keep = self._randperm(self.reservoir.size(0), self.reservoir.device, self.reservoir_size)
        self.reservoir = self.reservoir[keep].detach()

    def _reinit_from_reservoir(self):
        # reference modules.py:487-499.  Two departures, both fixing reference defects that would break training here:
        # (i) no process group (single-GPU run) -> world size 1 instead of the reference's crash in dist.get_world_size();
        # (ii) every rank clusters the same gathered pool, but from its OWN random initial centroids, so the replicas'
        #      codebooks drift apart and nothing re-synchronises them (DDP broadcasts parameters only at construction):
        #      rank 0's centroids are broadcast instead.
        from .kmeans import kmeans_fit
        distributed = dist.is_available() and dist.is_initialized()
        world_size = dist.get_world_size() if distributed else 1
        print("Updating codebook from reservoir.")
        if world_size > 1:
            gathered = [torch.zeros_like(self.reservoir) for _ in range(world_size)]
            dist.all_gather(gathered, self.reservoir.clone())
            pool = torch.cat(gathered, dim=0)
        else:
            pool = self.reservoir
        cent = kmeans_fit(pool, self.codebook_size).detach().contiguous()
        if world_size > 1:
            dist.broadcast(cent, 0)
        self.embedding.weight.data = cent

    def forward(self, z):
        z = ops.nhwc(z, torch.float32)                    # b c h w logical, NHWC memory: the flatten is a view
        batch_size = z.size(0)
        if self.training:
            self.q_counter += 1
            if self.q_counter > self.q_start_collect:
                self._collect(z.permute(0, 2, 3, 1).reshape(-1, self.codebook_dim), batch_size)
            if self.q_counter < self.q_init:
                return z, z.new_tensor(0), None           # warm-up: unquantised, loss 0, indices None
            if self.q_init <= self.q_counter < self.q_re_end:
                if (self.q_counter - self.q_init) % self.q_re_step == 0 or self.q_counter == self.q_init + self.q_re_end - 1:
                    self._reinit_from_reservoir()
        z_q, loss, min_encoding_indices = ops.vq_lookup(z, self.embedding.weight, self.beta)
        return z_q, loss, min_encoding_indices

    def get_codebook_entry(self, indices, shape):
        z_q = self.embedding(indices)
        if shape is not None:
            z_q = z_q.view(shape)
            z_q = z_q.permute(0, 3, 1, 2).contiguous()
        return z_q
