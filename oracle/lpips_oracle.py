"""TEST INFRASTRUCTURE (only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import this).

CPU restatement of the reference's LPIPS-VGG16 perceptual distance (reference losses/lpips.py:41-144) in plain fp32 torch
functional calls.  The arithmetic lives in third-party code the reference does not vendor: ``torchvision.models.vgg16`` (version
unpinned; the published configuration 'D': 13 3x3/pad-1 convolutions of 64,64 | 128,128 | 256,256,256 | 512,512,512 | 512,512,512
channels, ReLU after each, 2x2/stride-2 max-pool between the groups) -- restated here, anchored on the reference's own slicing of
``features[0:4] / [4:9] / [9:16] / [16:23] / [23:30]`` (lpips.py:103-108).  Pinned by tests/golden/lpips_tiny.npz, produced by the
reference's own LPIPS class (tests/golden/make_lpips_golden.py) on synthetic weights (``synth_lpips_state_dict``): the pretrained
files (torchvision's VGG16, the linear heads of ``vgg.pth``, lpips.py:10-15) are not available offline, so PARITY IS PINNED ON THE
ARITHMETIC AND THE state_dict LAYOUT, NOT ON THE PRETRAINED VALUES."""
import zlib

import torch
import torch.nn.functional as F

VGG_CFG = [64, 64, "M", 128, 128, "M", 256, 256, 256, "M", 512, 512, 512, "M", 512, 512, 512]
CHANNELS = [64, 128, 256, 512, 512]                    # lpips.py:45
# conv index inside each slice's nn.Sequential (lpips.py:104-108: slices of torchvision's features keep their order)
SLICE_CONVS = {"slice1": [(0, 3, 64), (2, 64, 64)],
               "slice2": [(1, 64, 128), (3, 128, 128)],
               "slice3": [(1, 128, 256), (3, 256, 256), (5, 256, 256)],
               "slice4": [(1, 256, 512), (3, 512, 512), (5, 512, 512)],
               "slice5": [(1, 512, 512), (3, 512, 512), (5, 512, 512)]}
SHIFT = (-.030, -.088, -.188)                           # lpips.py:81
SCALE = (.458, .448, .450)                              # lpips.py:82


def expected_keys():
    """state_dict keys of the reference's LPIPS module (buffers included), in its registration order."""
    keys = ["scaling_layer.shift", "scaling_layer.scale"]
    for s, convs in SLICE_CONVS.items():
        for i, _, _ in convs:
            keys += [f"vgg.{s}.{i}.weight", f"vgg.{s}.{i}.bias"]
    keys += [f"lin{i}.model.1.weight" for i in range(5)]
    return keys


def synth_lpips_state_dict(seed=3):
    """deterministic stand-in weights keyed like the reference's module: He-scaled convolutions (activations stay O(1) through 13
    layers), small biases, non-negative linear heads (the published heads are non-negative)."""
    sd = {"scaling_layer.shift": torch.tensor(SHIFT)[None, :, None, None], "scaling_layer.scale": torch.tensor(SCALE)[None, :, None, None]}

    def gen(name):
        return torch.Generator().manual_seed(seed * 1000003 + zlib.crc32(name.encode()))
    for s, convs in SLICE_CONVS.items():
        for i, cin, cout in convs:
            kw, kb = f"vgg.{s}.{i}.weight", f"vgg.{s}.{i}.bias"
            sd[kw] = torch.randn(cout, cin, 3, 3, generator=gen(kw)) * (2.0 / (cin * 9)) ** 0.5
            sd[kb] = torch.randn(cout, generator=gen(kb)) * 0.05
    for i, c in enumerate(CHANNELS):
        k = f"lin{i}.model.1.weight"
        sd[k] = torch.rand(1, c, 1, 1, generator=gen(k)) * (2.0 / c)
    return sd


class _RoundBf16(torch.autograd.Function):
    """value AND gradient pass through a bf16 store (what a kernel with bf16 activations / bf16 data gradients does)"""

    @staticmethod
    def forward(ctx, x):
        return x.bfloat16().float()

    @staticmethod
    def backward(ctx, g):
        return g.bfloat16().float()


def vgg_features(sd, x, bf16_storage=False):
    """lpips.py:111-124: the five ReLU taps relu1_2, relu2_2, relu3_3, relu4_3, relu5_3.  ``bf16_storage``: the PRECISION MODEL of
    the HIP path (bf16 weights, every activation and every data gradient stored in bf16, fp32 accumulation) -- separates what the
    storage format costs from what a kernel gets wrong."""
    rnd = _RoundBf16.apply if bf16_storage else (lambda t: t)
    feats = []
    h = rnd(x)
    for s, convs in SLICE_CONVS.items():
        if s != "slice1":
            h = F.max_pool2d(h, 2, 2)
        for i, _, _ in convs:
            w = sd[f"vgg.{s}.{i}.weight"]
            w = w.bfloat16().float() if bf16_storage else w
            h = F.relu(rnd(F.conv2d(h, w, sd[f"vgg.{s}.{i}.bias"], padding=1)))
        feats.append(h)
    return feats


def norm_tensor(x):
    """lpips.py:127-134"""
    return x / (torch.sqrt(torch.sum(x ** 2, dim=1, keepdim=True)) + 1e-10)


def lpips(sd, real_x, fake_x, bf16_storage=False):
    """lpips.py:67-76 (eval mode: the Dropout in front of every linear head is the identity) -> [B,1,1,1]"""
    shift, scale = sd["scaling_layer.shift"], sd["scaling_layer.scale"]
    fr = vgg_features(sd, (real_x - shift) / scale, bf16_storage)
    ff = vgg_features(sd, (fake_x - shift) / scale, bf16_storage)
    total = 0
    for i in range(5):
        d = (norm_tensor(fr[i]) - norm_tensor(ff[i])) ** 2
        total = total + F.conv2d(d, sd[f"lin{i}.model.1.weight"]).mean([2, 3], keepdim=True)
    return total
