"""A DETERMINISTIC stand-in for ``fast_pytorch_kmeans.KMeans`` (the third-party class the reference's codebook re-initialisation calls,
reference models/modules.py:489-499; absent from /root/reference and not installable offline).  Same surface as the reference uses:
``KMeans(n_clusters=K)``, ``.fit_predict(x)``, ``.centroids``.  It is injected into BOTH sides -- as the ``fast_pytorch_kmeans`` module the
reference imports (tests/golden/make_reinit_golden.py) and as ``models.kmeans.kmeans_fit`` of the product (tests/test_dp_gloo.py) -- so that
everything AROUND the clustering (reservoir sampling :477-481, all_gather :490-495, embedding overwrite :499, the schedule :483-488) can
be compared bit for bit.  One Lloyd step in fp64 from evenly strided seeds: a pure function of the pool."""
import torch


class KMeans:
    def __init__(self, n_clusters, **_kw):
        self.n_clusters = n_clusters
        self.centroids = None

    def fit_predict(self, x):
        pts = x.detach().double()
        seeds = torch.linspace(0, pts.shape[0] - 1, self.n_clusters).round().long()
        cent = pts[seeds].clone()
        lab = torch.cdist(pts, cent).argmin(1)
        acc = torch.zeros_like(cent).index_add_(0, lab, pts)
        cnt = torch.bincount(lab, minlength=self.n_clusters).double().unsqueeze(1)
        self.centroids = torch.where(cnt > 0, acc / cnt.clamp_min(1.0), cent).float()
        return lab


def kmeans_fit(points, n_clusters, **_kw):
    """the product's ``models.kmeans.kmeans_fit(points, n_clusters)`` signature"""
    m = KMeans(n_clusters)
    m.fit_predict(points)
    return m.centroids


# the schedule both sides run (small init_steps so that collection, warm-up, two re-initialisations and steady-state lookups all happen)
CFG = dict(codebook_size=16, codebook_dim=8, beta=0.25, init_steps=4, reservoir_size=30)
STEPS = 15
BATCH, HW = 3, 4


def latents(rank, step):
    """the z of (rank, step): numpy stream, independent of torch's RNG (which both sides consume for the reservoir's randperm only)"""
    import numpy as np
    rs = np.random.RandomState(1000 * rank + step)
    return torch.from_numpy(rs.randn(BATCH, CFG["codebook_dim"], HW, HW).astype(np.float32) + 0.5 * rank)
