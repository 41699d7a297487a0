"""Codebook re-initialisation by k-means over the latent reservoir (reference models/modules.py:487-499, where it is
delegated to the third-party ``fast_pytorch_kmeans.KMeans`` -- absent from the reference tree, unpinned -- with its
defaults: random distinct data points as initial centroids, euclidean, max_iter=100, tol=1e-4, an empty cluster's
centroid becomes the zero vector).  Lloyd iterations here use the same fused nearest-codebook HIP kernel as the
forward pass for the assignment step; oracle/kmeans_oracle.py restates the algorithm for the tests."""
import torch

from mas_hip import ops


@torch.no_grad()
def kmeans_fit(points: torch.Tensor, n_clusters: int, max_iter: int = 100, tol: float = 1e-4, init_idx=None,
               return_info: bool = False):
    pts = points.detach().float().contiguous()
    m, d = pts.shape
    if init_idx is None:
        if m < n_clusters:
            raise RuntimeError(f"kmeans_fit: {m} points for {n_clusters} clusters (the library samples without replacement)")
        init_idx = torch.randperm(m, device=pts.device)[:n_clusters]
    cent = pts[init_idx].clone()
    z = pts.t().reshape(1, d, m, 1)                       # [1,D,M,1] logical == rows of pts in NHWC memory
    ones = torch.ones(m, device=pts.device)
    idx, it = None, 0
    for it in range(1, max_iter + 1):
        _, _, idx = ops.vq_lookup(z, cent, 0.0)           # nearest centroid, lowest index on ties (the VQ kernel)
        sums = torch.zeros_like(cent).index_add_(0, idx, pts)
        cnt = torch.zeros(n_clusters, device=pts.device).index_add_(0, idx, ones)
        new = torch.where(cnt[:, None] > 0, sums / cnt.clamp(min=1)[:, None], torch.zeros_like(cent))
        err = (new - cent).pow(2).sum()
        cent = new
        if float(err) <= tol:
            break
    return (cent, idx, it) if return_info else cent
