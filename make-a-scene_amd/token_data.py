"""Token dataset between stage 1 (VQ-SEG / VQ-IMG) and stage 2 (the autoregressive transformer) -- SURVEY 8(f) rank 4.

The reference's transformer loop consumes ``img_token, seg_token, _, _, text_token = data`` (train.py:141-145) but nothing in
the reference produces such tuples (its webdataset preprocessors emit pixels).  This module is the missing producer / format:

* ``tokenize_batch``: frozen ``VQBASE.encode_to_indices`` on images and segmentation maps (the MI355X encoder + VQ kernels);
* ``TokenShardWriter`` / ``TokenShard``: a flat little-endian shard file, fixed-length records, memory-mapped on read;
* ``TokenDataset``: a ``torch.utils.data.Dataset`` over shards yielding exactly the 5-tuple train.py unpacks
  (positions 2 and 3, ignored there, are zero scalars).

Shard layout (version 1), all integers little-endian:
    0   8 bytes  magic  b"MASTOK01"
    8   u32      n_samples
    12  u32      img_len      (e.g. 1024 = 32x32 latent grid)
    16  u32      seg_len      (e.g. 256)
    20  u32      text_len     (e.g. 256; zero padded, train.py:147 / transformer.py:350-353)
    24  u32      img_vocab, 28 u32 seg_vocab, 32 u32 text_vocab
    36  u8 x 3   bytes per token of the img / seg / text arrays (2 = uint16 when vocab <= 65536, else 4 = uint32)
    39  25 bytes zero padding (header = 64 bytes)
    64  img  [n_samples][img_len], then seg [n_samples][seg_len], then text [n_samples][text_len]
"""
from __future__ import annotations

import os
import struct
from typing import Iterable, List, Sequence

import numpy as np
import torch

MAGIC = b"MASTOK01"
HEADER_BYTES = 64
_HDR = struct.Struct("<8s7I3B25x")
assert _HDR.size == HEADER_BYTES


def _width(vocab: int) -> int:
    return 2 if vocab <= 65536 else 4


def _np_dtype(width: int):
    return np.dtype("<u2") if width == 2 else np.dtype("<u4")


class TokenShardWriter:
    """Appends (img, seg, text) token batches to one shard file; ``close()`` (or the context manager) finalises the header."""

    def __init__(self, path: str, img_len: int, seg_len: int, text_len: int, img_vocab: int, seg_vocab: int, text_vocab: int):
        self.path = path
        self.lens = (int(img_len), int(seg_len), int(text_len))
        self.vocabs = (int(img_vocab), int(seg_vocab), int(text_vocab))
        self.widths = tuple(_width(v) for v in self.vocabs)
        self.n = 0
        self._parts: List[List[np.ndarray]] = [[], [], []]

    def append(self, img_tokens, seg_tokens, text_tokens) -> None:
        arrs = []
        for t, ln, vocab, w in zip((img_tokens, seg_tokens, text_tokens), self.lens, self.vocabs, self.widths):
            a = t.detach().cpu().numpy() if isinstance(t, torch.Tensor) else np.asarray(t)
            if a.ndim != 2 or a.shape[1] != ln:
                raise ValueError(f"token batch of shape {a.shape}, expected [batch, {ln}]")
            if a.size and (a.min() < 0 or a.max() >= vocab):
                raise ValueError(f"token out of range [0, {vocab})")
            arrs.append(a.astype(_np_dtype(w)))
        if not (arrs[0].shape[0] == arrs[1].shape[0] == arrs[2].shape[0]):
            raise ValueError("img / seg / text batches differ in length")
        for k in range(3):
            self._parts[k].append(arrs[k])
        self.n += arrs[0].shape[0]

    def close(self) -> None:
        tmp = self.path + ".tmp"
        with open(tmp, "wb") as f:
            f.write(_HDR.pack(MAGIC, self.n, *self.lens, *self.vocabs, *self.widths))
            for k in range(3):
                for a in self._parts[k]:
                    f.write(np.ascontiguousarray(a).tobytes())
                if not self._parts[k]:
                    pass
        os.replace(tmp, self.path)          # a reader never sees a half-written shard
        self._parts = [[], [], []]

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        if exc[0] is None:
            self.close()


class TokenShard:
    """Memory-mapped view of one shard."""

    def __init__(self, path: str):
        self.path = path
        with open(path, "rb") as f:
            hdr = f.read(HEADER_BYTES)
        if len(hdr) != HEADER_BYTES:
            raise ValueError(f"{path}: truncated header")
        magic, n, il, sl, tl, iv, sv, tv, iw, sw, tw = _HDR.unpack(hdr)
        if magic != MAGIC:
            raise ValueError(f"{path}: not a Make-A-Scene token shard (magic {magic!r})")
        self.n, self.lens, self.vocabs, self.widths = n, (il, sl, tl), (iv, sv, tv), (iw, sw, tw)
        if any(w not in (2, 4) for w in self.widths):
            raise ValueError(f"{path}: bad token width {self.widths}")
        need = HEADER_BYTES + n * sum(l * w for l, w in zip(self.lens, self.widths))
        if os.path.getsize(path) != need:
            raise ValueError(f"{path}: size {os.path.getsize(path)} != {need} implied by the header")
        self._arr = None

    def arrays(self):
        if self._arr is None:            # opened lazily: DataLoader workers map the file themselves after the fork / pickle
            off, out = HEADER_BYTES, []
            for ln, w in zip(self.lens, self.widths):
                out.append(np.memmap(self.path, dtype=_np_dtype(w), mode="r", offset=off, shape=(self.n, ln)) if self.n else
                           np.zeros((0, ln), dtype=_np_dtype(w)))
                off += self.n * ln * w
            self._arr = tuple(out)
        return self._arr

    def __getstate__(self):
        d = dict(self.__dict__)
        d["_arr"] = None
        return d

    def __len__(self):
        return self.n


class TokenDataset(torch.utils.data.Dataset):
    """``img_token, seg_token, _, _, text_token`` records over one or more shards (reference train.py:141-145)."""

    def __init__(self, paths: Sequence[str]):
        self.shards = [TokenShard(p) for p in paths]
        if not self.shards:
            raise ValueError("TokenDataset: no shards")
        for s in self.shards[1:]:
            if s.lens != self.shards[0].lens or s.vocabs != self.shards[0].vocabs:
                raise ValueError(f"{s.path}: token lengths / vocabularies differ from {self.shards[0].path}")
        self.offsets = np.cumsum([0] + [len(s) for s in self.shards])

    def __len__(self):
        return int(self.offsets[-1])

    def __getitem__(self, i: int):
        if i < 0:
            i += len(self)
        if not 0 <= i < len(self):
            raise IndexError(i)
        k = int(np.searchsorted(self.offsets, i, side="right") - 1)
        img, seg, text = self.shards[k].arrays()
        j = i - int(self.offsets[k])
        as_t = lambda a: torch.from_numpy(np.asarray(a[j]).astype(np.int64))
        zero = torch.zeros((), dtype=torch.int64)
        return as_t(img), as_t(seg), zero, zero, as_t(text)


@torch.no_grad()
def tokenize_batch(vq_img, vq_seg, images: torch.Tensor, segmentations: torch.Tensor):
    """Frozen-VQ encode of one batch: images [B,3,H,W] and segmentation maps [B,C_seg,h,w] -> (img_tokens [B, (H/16)^2],
    seg_tokens [B, (h/16)^2]) on the MI355X encoder / VQ kernels (BASELINE config 5's first stage)."""
    return vq_img.encode_to_indices(images), vq_seg.encode_to_indices(segmentations)


def write_token_shards(out_dir: str, batches: Iterable, img_vocab: int, seg_vocab: int, text_vocab: int, samples_per_shard: int = 65536,
                       prefix: str = "tokens") -> List[str]:
    """``batches`` yields (img_tokens [b, Li], seg_tokens [b, Ls], text_tokens [b, Lt]); returns the shard paths written."""
    os.makedirs(out_dir, exist_ok=True)
    paths: List[str] = []
    w = None
    for img, seg, text in batches:
        if w is None or w.n >= samples_per_shard:
            if w is not None:
                w.close()
            paths.append(os.path.join(out_dir, f"{prefix}-{len(paths):05d}.mastok"))
            w = TokenShardWriter(paths[-1], img.shape[1], seg.shape[1], text.shape[1], img_vocab, seg_vocab, text_vocab)
        w.append(img, seg, text)
    if w is not None:
        w.close()
    return paths
