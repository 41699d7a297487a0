"""LPIPS-VGG16 perceptual distance (reference losses/lpips.py:41-144) on the MI355X convolution kernels.

Same class names, constructor-free surface and ``state_dict`` layout as the reference (``scaling_layer.{shift,scale}``,
``vgg.slice<k>.<i>.{weight,bias}`` with the indices torchvision's ``features[a:b]`` slices give, ``lin<k>.model.1.weight``), so a
state_dict saved from the reference's module loads with ``strict=True``.  The thirteen 3x3 convolutions of VGG16 -- all of the
FLOPs: ~20 GFLOP per 256x256 image and pass -- run on libmas_hip's implicit-GEMM kernels in bf16 with fp32 accumulation, forward
and data gradient (the network is frozen: no weight gradients); real and reconstructed images go through the stack as ONE batch.
ReLU, the four 2x2 max-pools and the channel normalisation / squared difference / 1x1 linear heads / spatial mean of the distance
are the reference's own torch expressions (elementwise and reduction passes).

What the reference does at construction and this module cannot: it downloads ``vgg.pth`` and takes torchvision's pretrained VGG16
(lpips.py:10-15,55,101).  There is no network here: weights come from ``MAS_LPIPS_CKPT`` (a state_dict with the keys above, e.g.
``LPIPS().state_dict()`` saved from the reference), or from the reference's path if that file exists, plus ``MAS_VGG16_CKPT``
(torchvision's VGG16 state_dict) for the backbone when the first file holds the heads only -- as the reference's ``vgg.pth``
does.  Every tensor still at its random initialisation afterwards is named in a warning (``LPIPS.unloaded``;
an error by default, as in the reference, whose constructor cannot run without its weight files; ``MAS_LPIPS_STRICT=0`` makes it a
warning) -- the loss is then a perceptual distance in name only."""
import os
import warnings

import torch
import torch.nn as nn
import torch.nn.functional as F

from models.modules import Conv2d

CKPT_PATHS = (os.environ.get("MAS_LPIPS_CKPT", ""), "/home/ubuntu/Make-A-Scene/weights/vgg.pth")      # lpips.py:15


class ScalingLayer(nn.Module):
    def __init__(self):
        super().__init__()
        self.register_buffer("shift", torch.Tensor([-.030, -.088, -.188])[None, :, None, None])
        self.register_buffer("scale", torch.Tensor([.458, .448, .450])[None, :, None, None])

    def forward(self, x):
        return (x - self.shift) / self.scale


class NetLinLayer(nn.Module):
    """Dropout + bias-free 1x1 convolution C -> 1 (lpips.py:89-95); a plain nn.Conv2d holder: the head is applied as a weighted
    channel sum inside ``LPIPS.forward``."""

    def __init__(self, in_channels, out_channels=1):
        super().__init__()
        self.model = nn.Sequential(nn.Dropout(), nn.Conv2d(in_channels, out_channels, 1, 1, 0, bias=False))


class _VggConv(Conv2d):
    in_dtype = torch.bfloat16
    out_dtype = torch.bfloat16


class VGG16(nn.Module):
    """torchvision's VGG16 ``features[0:30]`` in the reference's five slices (lpips.py:98-124)."""

    def __init__(self):
        super().__init__()

        def block(cin, cout, n, pool):
            layers = [nn.MaxPool2d(kernel_size=2, stride=2)] if pool else []
            for k in range(n):
                layers += [_VggConv(cin if k == 0 else cout, cout, 3, 1, 1), nn.ReLU(inplace=True)]
            return nn.Sequential(*layers)
        self.slice1 = block(3, 64, 2, False)
        self.slice2 = block(64, 128, 2, True)
        self.slice3 = block(128, 256, 3, True)
        self.slice4 = block(256, 512, 3, True)
        self.slice5 = block(512, 512, 3, True)
        for param in self.parameters():
            param.requires_grad = False

    def forward(self, x):
        feats = []
        h = x
        for s in (self.slice1, self.slice2, self.slice3, self.slice4, self.slice5):
            h = s(h)
            feats.append(h)
        return tuple(feats)


def norm_tensor(x):
    norm_factor = torch.sqrt(torch.sum(x ** 2, dim=1, keepdim=True))
    return x / (norm_factor + 1e-10)


def spatial_average(x):
    return x.mean([2, 3], keepdim=True)


class LPIPS(nn.Module):
    _warned = False

    def __init__(self):
        super().__init__()
        self.scaling_layer = ScalingLayer()
        self.channels = [64, 128, 256, 512, 512]
        self.vgg = VGG16()
        self.lin0 = NetLinLayer(self.channels[0])
        self.lin1 = NetLinLayer(self.channels[1])
        self.lin2 = NetLinLayer(self.channels[2])
        self.lin3 = NetLinLayer(self.channels[3])
        self.lin4 = NetLinLayer(self.channels[4])
        self.load_from_pretrained()
        self.lins = [self.lin0, self.lin1, self.lin2, self.lin3, self.lin4]
        for param in self.parameters():
            param.requires_grad = False

    # torchvision vgg16().features index of each convolution -> (slice, index inside the slice) (lpips.py:103-108: features[0:4],
    # [4:9], [9:16], [16:23], [23:30])
    _TV_CONVS = {0: ("slice1", 0), 2: ("slice1", 2), 5: ("slice2", 1), 7: ("slice2", 3), 10: ("slice3", 1), 12: ("slice3", 3),
                 14: ("slice3", 5), 17: ("slice4", 1), 19: ("slice4", 3), 21: ("slice4", 5), 24: ("slice5", 1), 26: ("slice5", 3),
                 28: ("slice5", 5)}

    def load_from_pretrained(self, name="vgg_lpips"):
        """The reference fills this module from TWO sources: torchvision's pretrained VGG16 for the backbone (lpips.py:101) and
        ``vgg.pth`` for the five ``lin<k>`` heads (lpips.py:55-57, ``strict=False``) -- that file alone leaves the 13 backbone
        convolutions untouched.  Here: ``MAS_LPIPS_CKPT`` (or the reference's ``vgg.pth`` path) may hold either the whole module's
        state_dict or the heads only; ``MAS_VGG16_CKPT`` may hold torchvision's ``vgg16().state_dict()`` / ``vgg16().features
        .state_dict()`` for the backbone.  Whatever is still at its random initialisation afterwards is an ERROR naming the tensors
        (``self.unloaded``); ``MAS_LPIPS_STRICT=0`` turns it into a warning."""
        want = set(self.state_dict().keys()) - {"scaling_layer.shift", "scaling_layer.scale"}
        loaded = set()
        for path in CKPT_PATHS:
            if path and os.path.exists(path):
                sd = torch.load(path, map_location=torch.device("cpu"))
                res = self.load_state_dict(sd, strict=False)
                loaded |= want - set(res.missing_keys)
                break
        tv_path = os.environ.get("MAS_VGG16_CKPT", "")
        if tv_path and os.path.exists(tv_path):
            tv = torch.load(tv_path, map_location=torch.device("cpu"))
            mine = self.state_dict()
            for idx, (sl, j) in self._TV_CONVS.items():
                for leaf in ("weight", "bias"):
                    src = tv.get(f"features.{idx}.{leaf}", tv.get(f"{idx}.{leaf}"))
                    key = f"vgg.{sl}.{j}.{leaf}"
                    if src is not None and tuple(src.shape) == tuple(mine[key].shape):
                        mine[key].copy_(src)
                        loaded.add(key)
        self.unloaded = sorted(want - loaded)
        if not self.unloaded:
            return
        backbone = [k for k in self.unloaded if k.startswith("vgg.")]
        heads = [k for k in self.unloaded if k.startswith("lin")]
        msg = ("LPIPS: %d of %d tensors keep their RANDOM initialisation (backbone: %d of 26 -- set MAS_VGG16_CKPT to torchvision's "
               "vgg16 state_dict, or MAS_LPIPS_CKPT to a full state_dict of the reference's LPIPS module; heads: %d of 5 -- "
               "MAS_LPIPS_CKPT / vgg.pth); the perceptual term is NOT a perceptual distance until they are loaded.  Missing: %s"
               % (len(self.unloaded), len(want), len(backbone), len(heads), ", ".join(self.unloaded[:6]) + (" ..." if len(self.unloaded) > 6 else "")))
        if os.environ.get("MAS_LPIPS_STRICT", "1") == "1":      # the default: the reference fails hard on missing weights too (lpips.py:15,55)
            raise RuntimeError(msg + "  (MAS_LPIPS_STRICT=0 downgrades this to a warning: tests, arithmetic checks with synthetic weights)")
        if loaded or not LPIPS._warned:       # a PARTIAL load (e.g. vgg.pth alone: heads without backbone) is reported every time
            LPIPS._warned = True
            warnings.warn(msg)

    def forward(self, real_x, fake_x):
        b = real_x.shape[0]
        feats = self.vgg(self.scaling_layer(torch.cat([real_x, fake_x], dim=0)))            # one pass for both images
        total = 0
        for i, f in enumerate(feats):
            f = f.float()
            d = (norm_tensor(f[:b]) - norm_tensor(f[b:])) ** 2
            w = self.lins[i].model[1].weight.float()                                          # [1, C, 1, 1], eval-mode Dropout = identity
            if self.training and self.lins[i].model[0].p > 0 and self.lins[i].training:
                d = self.lins[i].model[0](d)
            total = total + spatial_average((d * w).sum(dim=1, keepdim=True))
        return total
