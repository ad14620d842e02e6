"""TEST INFRASTRUCTURE ONLY (never imported by the product): CPU restatement of simple_knn.distCUDA2 as the reference uses it at
/root/reference/MVs_Algorithms/GaussianSplatting/main_3DGS_renderer.py:420 -- per point, the mean of the squared distances to its three
nearest OTHER points.  The wheel's source is not in the reference tree (parity unpinned); the statistic itself is unambiguous and is computed
here exactly with a k-d tree (scipy) in float64 on the float32 coordinates."""
import numpy as np
from scipy.spatial import cKDTree


def dist2_mean3(points):
    p = np.asarray(points, dtype=np.float32).astype(np.float64)
    n = p.shape[0]
    if n == 0:
        return np.zeros((0,), np.float32)
    k = min(n, 8)
    d, idx = cKDTree(p).query(p, k=k)
    d, idx = d.reshape(n, k), idx.reshape(n, k)
    out = np.zeros(n)
    for i in range(n):
        others = d[i][idx[i] != i][:3]            # self can sit anywhere among duplicates: drop it by index, keep the three nearest others
        if others.shape[0] < 3 and n > k:         # more than k-1 duplicates of this position: brute force this one
            dd = np.sort(((p - p[i]) ** 2).sum(1))
            others = np.sqrt(np.delete(dd, 0)[:3])
        out[i] = (others ** 2).sum() / 3.0
    return out.astype(np.float32)
