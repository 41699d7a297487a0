"""Codebook re-initialisation by k-means over the latent reservoir (reference
models/modules.py:487-499, where it is delegated to the third-party ``fast_pytorch_kmeans.KMeans``
with its defaults: random-point init, euclidean, max_iter=100, tol=1e-4).  Lloyd iterations here
use the same fused nearest-codebook HIP kernel as the forward pass for the assignment step."""
import torch

from mas_hip import ops


@torch.no_grad()
def kmeans_fit(points: torch.Tensor, n_clusters: int, max_iter: int = 100, tol: float = 1e-4) -> torch.Tensor:
    pts = points.detach().float().contiguous()
    m, d = pts.shape
    pick = torch.randperm(m, device=pts.device)[:n_clusters]
    cent = pts[pick].clone()
    if cent.shape[0] < n_clusters:                       # fewer points than clusters: pad with jittered copies
        extra = pts[torch.randint(0, m, (n_clusters - cent.shape[0],), device=pts.device)]
        cent = torch.cat([cent, extra + 1e-4 * torch.randn_like(extra)], dim=0)
    z = pts.t().reshape(1, d, m, 1)                       # [1,D,M,1] logical == rows of pts in NHWC memory
    for _ in range(max_iter):
        _, _, idx = ops.vq_lookup(z, cent, 0.0)
        sums = torch.zeros_like(cent).index_add_(0, idx, pts)
        cnt = torch.zeros(n_clusters, device=pts.device).index_add_(0, idx, torch.ones(m, device=pts.device))
        new = torch.where(cnt[:, None] > 0, sums / cnt.clamp(min=1)[:, None], cent)
        err = (new - cent).pow(2).sum()
        cent = new
        if float(err) <= tol:
            break
    return cent
