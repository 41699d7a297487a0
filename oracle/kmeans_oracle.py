"""TEST INFRASTRUCTURE ONLY (imported by tests/ alone): CPU restatement of the codebook re-initialisation's k-means.

The reference delegates it to the third-party ``fast_pytorch_kmeans.KMeans`` (models/modules.py:8,489-499), which is
NOT vendored in /root/reference and NOT installed here; the reference pins no version (no requirements file).  PARITY
UNPINNED for this piece: what follows restates the library's published algorithm (fast_pytorch_kmeans 0.1.x,
``KMeans(n_clusters, max_iter=100, tol=1e-4, mode='euclidean')``, ``fit_predict``):
  * initial centroids = ``n_clusters`` distinct data points drawn at random (here: passed in, so tests are deterministic);
  * each iteration: assign every point to its nearest centroid (argmax of -(|a|^2 - 2ab + |b|^2), lowest index on ties),
    new centroid = mean of its points, an EMPTY cluster's centroid becomes the zero vector (the library's NaN -> 0);
  * stop when sum((new - old)^2) <= tol or after max_iter iterations.
"""
import numpy as np


def kmeans_lloyd(points: np.ndarray, init_idx: np.ndarray, max_iter: int = 100, tol: float = 1e-4):
    x = points.astype(np.float64)
    cent = x[init_idx].copy()
    k = cent.shape[0]
    assign = None
    for it in range(max_iter):
        d = (x * x).sum(1)[:, None] + (cent * cent).sum(1)[None, :] - 2.0 * x @ cent.T
        assign = d.argmin(1)
        new = np.zeros_like(cent)
        np.add.at(new, assign, x)
        cnt = np.bincount(assign, minlength=k).astype(np.float64)
        new = np.where(cnt[:, None] > 0, new / np.maximum(cnt, 1)[:, None], 0.0)
        err = float(((new - cent) ** 2).sum())
        cent = new
        if err <= tol:
            break
    return cent.astype(np.float32), assign, it + 1
