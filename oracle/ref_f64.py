"""Independent float64 truths (test infrastructure only; see oracle/__init__.py).

These do not follow any reference file: they are the mathematical definitions the
oracle and the HIP kernels are bounded against.
"""
import numpy as np


def chamfer_f64(p1, p2):
    """p1 [B,N,3], p2 [B,M,3] -> (chamfer [B], d1 [B,N], i1, d2 [B,M], i2) in float64.
    chamfer = mean_i min_j |p1_i - p2_j|^2 + mean_j min_i |p2_j - p1_i|^2
    (the definition behind kaolin.metrics.pointcloud.chamfer_distance, w1 = w2 = 1)."""
    p1 = np.asarray(p1, np.float64)
    p2 = np.asarray(p2, np.float64)
    diff = p1[:, :, None, :] - p2[:, None, :, :]
    d = (diff * diff).sum(-1)
    d1, i1 = d.min(2), d.argmin(2)
    d2, i2 = d.min(1), d.argmin(1)
    return d1.mean(1) + d2.mean(1), d1, i1, d2, i2


def exact_emd(p1, p2):
    """Exact earth mover's distance for equal-size uniform clouds: the minimum over
    permutations of sum_i |p1_i - p2_perm(i)| (Euclidean, not squared), by the Hungarian
    method.  p1, p2 [N,3] -> (total cost, assignment)."""
    from scipy.optimize import linear_sum_assignment
    p1 = np.asarray(p1, np.float64)
    p2 = np.asarray(p2, np.float64)
    c = np.sqrt(((p1[:, None, :] - p2[None, :, :]) ** 2).sum(-1))
    r, col = linear_sum_assignment(c)
    return c[r, col].sum(), col
