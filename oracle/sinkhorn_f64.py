"""CPU ORACLE -- test infrastructure, NOT product code.

Independent float64 restatement of the Sinkhorn divergence the reference's ``emd_wrapper`` computes
(``/root/reference/src/models/utils.py:9,12-13``: ``neuralnet_pytorch.metrics.emd_loss(sinkhorn=True)``
-> ``geomloss.SamplesLoss()`` with its defaults).  Both packages are absent from the reference tree
and unpinned, so this follows geomloss' PUBLISHED algorithm -- as of the 0.2.x line,
``geomloss/sinkhorn_samples.py::sinkhorn_tensorized`` + ``sinkhorn_divergence.py::{epsilon_schedule,
sinkhorn_loop, sinkhorn_cost}``, restated from memory of that source -- and parity stays UNPINNED:

  SamplesLoss defaults: loss="sinkhorn", p=2, blur=0.05, reach=None (balanced), diameter=None,
  scaling=0.5, debias=True, potentials=False, backend "tensorized" for small clouds.
  C(x,y) = |x - y|^2 / 2;  uniform weights;  diameter = |max - min| over ALL points of the batch;
  eps schedule = [diam^2] + [exp(e) for e in arange(2 ln diam, 2 ln blur, 2 ln scaling)] + [blur^2];
  softmin(eps, C, h)_i = -eps * logsumexp_j(h_j - C_ij / eps);
  loop: symmetric updates from the previous duals, averaged (new = (old + softmin)/2), one final
  un-averaged "extrapolation" step at the last eps;  cost = <a, f_ba - f_aa> + <b, g_ab - g_bb>.

Unlike ``oracle.sinkhorn_divergence`` (float32 soft-mins from the C file, O(N+M) memory, written
to mirror the HIP kernel's data flow) this version keeps explicit [N,M] cost matrices in float64
and uses scipy's logsumexp: a different route to the same number, used as the cross-check.
"""
from __future__ import annotations

import numpy as np
from scipy.special import logsumexp


def epsilon_schedule(p: float, diameter: float, blur: float, scaling: float):
    return ([diameter ** p]
            + [float(np.exp(e)) for e in np.arange(p * np.log(diameter), p * np.log(blur), p * np.log(scaling))]
            + [blur ** p])


def _softmin(eps, C, h):
    """C [N,M], h [M] -> [N]"""
    return -eps * logsumexp(h[None, :] - C / eps, axis=1)


def sinkhorn_divergence_f64(x, y, blur: float = 0.05, scaling: float = 0.5):
    """x [B,N,3], y [B,M,3] -> [B] float64."""
    x = np.asarray(x, np.float64)
    y = np.asarray(y, np.float64)
    B, N, _ = x.shape
    M = y.shape[1]
    pts = np.concatenate([x.reshape(-1, 3), y.reshape(-1, 3)])
    diameter = float(np.linalg.norm(pts.max(0) - pts.min(0)))
    eps_list = epsilon_schedule(2, diameter, blur, scaling)
    out = np.empty(B)
    a_log = np.full(N, -np.log(N))
    b_log = np.full(M, -np.log(M))
    for b in range(B):
        sq = lambda u, v: 0.5 * ((u[:, None, :] - v[None, :, :]) ** 2).sum(-1)
        C_xx, C_yy, C_xy, C_yx = sq(x[b], x[b]), sq(y[b], y[b]), sq(x[b], y[b]), sq(y[b], x[b])
        eps = eps_list[0]
        g_ab = _softmin(eps, C_yx, a_log)          # on y, from a
        f_ba = _softmin(eps, C_xy, b_log)          # on x, from b
        f_aa = _softmin(eps, C_xx, a_log)
        g_bb = _softmin(eps, C_yy, b_log)
        for eps in eps_list:
            ft_ba = _softmin(eps, C_xy, b_log + g_ab / eps)
            gt_ab = _softmin(eps, C_yx, a_log + f_ba / eps)
            ft_aa = _softmin(eps, C_xx, a_log + f_aa / eps)
            gt_bb = _softmin(eps, C_yy, b_log + g_bb / eps)
            f_ba, g_ab = 0.5 * (f_ba + ft_ba), 0.5 * (g_ab + gt_ab)
            f_aa, g_bb = 0.5 * (f_aa + ft_aa), 0.5 * (g_bb + gt_bb)
        f_ba, g_ab = _softmin(eps, C_xy, b_log + g_ab / eps), _softmin(eps, C_yx, a_log + f_ba / eps)
        f_aa, g_bb = _softmin(eps, C_xx, a_log + f_aa / eps), _softmin(eps, C_yy, b_log + g_bb / eps)
        out[b] = (f_ba - f_aa).mean() + (g_ab - g_bb).mean()
    return out
