"""PatchGAN discriminator (reference losses/discriminator.py:17-38) on the MI355X convolution kernels.

Same constructor, same ``nn.Sequential`` layout and therefore the same ``state_dict`` keys (``model.0.weight`` ...
``model.11.bias``), same ``weights_init``.  The five 4x4 convolutions (stride 2, 2, 2, 1, 1; pad 1) run on libmas_hip's
implicit-GEMM kernels in bf16 with fp32 accumulation: forward and data gradient on ``conv_fwd_kernel<4x4>`` (the stride-2 data
gradient as a zero-stuffed stride-1 convolution), the weight gradient on the transpose-read kernel (stride 2 through a
space-to-depth image, ``mas_space_to_depth2x``).  Each ``BatchNorm2d`` + ``LeakyReLU(0.2)`` pair is one pass of ``batchnorm.hip`` on the
bf16 map (round 6: ``ops.batch_norm_leaky_relu`` -- fp64 fixed-order batch statistics, running statistics as torch keeps them, the
activation fused into the apply pass and its derivative into the backward sums; until round 5 torch's module on an fp32 round trip);
the modules stay in the ``nn.Sequential`` (parameters, buffers, ``state_dict`` keys).  Channel counts the kernels do not take
(C % 4 != 0, C > 1024) and CPU tensors keep the torch modules.  The first layer's bare ``LeakyReLU`` is torch's."""
import torch
import torch.nn as nn

from mas_hip import ops
from models.modules import Conv2d


def weights_init(m):
    classname = m.__class__.__name__
    if classname.find('Conv') != -1:
        nn.init.normal_(m.weight.data, 0.0, 0.02)
    elif classname.find('BatchNorm') != -1:
        nn.init.normal_(m.weight.data, 1.0, 0.02)
        nn.init.constant_(m.bias.data, 0)


class _DiscConv(Conv2d):
    """bf16 storage whatever the global compute dtype (the 4x4 geometries exist in bf16 only); ``out_dtype`` fp32 for the logits."""
    in_dtype = torch.bfloat16
    out_dtype = torch.bfloat16


class Discriminator(nn.Module):
    def __init__(self, in_channels=3, num_filters_last=64, n_layers=3):
        super(Discriminator, self).__init__()
        layers = [_DiscConv(in_channels, num_filters_last, 4, 2, 1), nn.LeakyReLU(0.2)]
        num_filters_mult = 1
        for i in range(1, n_layers + 1):
            num_filters_mult_last = num_filters_mult
            num_filters_mult = min(2 ** i, 8)
            layers += [
                _DiscConv(num_filters_last * num_filters_mult_last, num_filters_last * num_filters_mult, 4,
                          2 if i < n_layers else 1, 1, bias=False),
                nn.BatchNorm2d(num_filters_last * num_filters_mult),
                nn.LeakyReLU(0.2, True)
            ]
        layers.append(_DiscConv(num_filters_last * num_filters_mult, 1, 4, 1, 1))
        layers[-1].out_dtype = torch.float32          # patch logits in fp32: the hinge / adaptive-weight arithmetic reads them
        self.model = nn.Sequential(*layers)

    def forward(self, x):
        h = x
        mods = list(self.model)
        i = 0
        while i < len(mods):
            m = mods[i]
            if isinstance(m, nn.BatchNorm2d):
                nxt = mods[i + 1] if i + 1 < len(mods) else None
                fused = isinstance(nxt, nn.LeakyReLU)
                if h.is_cuda and h.dim() == 4 and h.shape[1] % 4 == 0 and h.shape[1] <= 1024 and h.dtype in (torch.bfloat16, torch.float32):
                    h = ops.batch_norm_leaky_relu(h, m, nxt.negative_slope if fused else 1.0)
                    i += 2 if fused else 1
                    continue
                h = m(h.float()).to(torch.bfloat16)   # statistics and affine in fp32 on the bf16 map (channels_last preserved)
            else:
                h = m(h)
            i += 1
        return h
