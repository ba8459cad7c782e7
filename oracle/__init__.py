"""CPU ORACLE -- test infrastructure, NOT product code.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this package; nothing under ``fpsg_amd/`` does.  It wraps ``oracle/fpsg_oracle.c``
(plain-C restatement of the hot-path arithmetic, see that file's header for what each
function follows in the reference and for the parity status of each op) with numpy, and
adds torch ``autograd.Function`` shims over the same C code so that the reference's
model code can be driven end to end on CPU as the checker / CPU baseline.

Parity status in one line each (details: oracle/fpsg_oracle.c):
  Chamfer  -- Kaolin 0.9.0 algorithm restated; pinned by the Kaolin docstring KAT and a
              float64 brute force (the reference itself holds no test for it).
  kNN/edge -- follow src/dgcnn/model.py:13-42; pinned by goldens generated from the
              reference's own functions (tests/golden/make_golden.py).
  EMD      -- PARITY UNPINNED (neuralnet_pytorch absent, unpinned); bounded against
              exact assignment (scipy) in tests.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libfpsg_oracle.so")

_f32p = ctypes.POINTER(ctypes.c_float)
_i32p = ctypes.POINTER(ctypes.c_int32)


def build(force: bool = False) -> str:
    """Compile oracle/fpsg_oracle.c with gcc (idempotent)."""
    src = os.path.join(_HERE, "fpsg_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.run(["make", "-C", _HERE] + (["-B"] if force else []), check=True,
                       stdout=subprocess.DEVNULL)
    return _SO


_lib = None


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build())
    return _lib


def _f32(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float32)


def _i32(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.int32)


def _p(a: np.ndarray):
    if a.dtype == np.float32:
        return a.ctypes.data_as(_f32p)
    if a.dtype == np.int32:
        return a.ctypes.data_as(_i32p)
    raise TypeError(a.dtype)


# ----------------------------------------------------------------------------- Chamfer
def sided_distance(a, b):
    """a [B,N,3], b [B,M,3] -> (dist [B,N] f32, idx [B,N] i32)."""
    a, b = _f32(a), _f32(b)
    B, N, _ = a.shape
    M = b.shape[1]
    dist = np.empty((B, N), np.float32)
    idx = np.empty((B, N), np.int32)
    lib().oracle_sided_distance(_p(a), _p(b), B, N, M, _p(dist), _p(idx))
    return dist, idx


def chamfer_fwd(xyz1, xyz2):
    """-> dist1 [B,N], idx1 [B,N], dist2 [B,M], idx2 [B,M]."""
    xyz1, xyz2 = _f32(xyz1), _f32(xyz2)
    B, N, _ = xyz1.shape
    M = xyz2.shape[1]
    d1 = np.empty((B, N), np.float32)
    i1 = np.empty((B, N), np.int32)
    d2 = np.empty((B, M), np.float32)
    i2 = np.empty((B, M), np.int32)
    lib().oracle_chamfer_fwd(_p(xyz1), _p(xyz2), B, N, M, _p(d1), _p(i1), _p(d2), _p(i2))
    return d1, i1, d2, i2


def chamfer_bwd(xyz1, xyz2, idx1, idx2, g1, g2):
    xyz1, xyz2, g1, g2 = _f32(xyz1), _f32(xyz2), _f32(g1), _f32(g2)
    idx1, idx2 = _i32(idx1), _i32(idx2)
    B, N, _ = xyz1.shape
    M = xyz2.shape[1]
    gx1 = np.empty_like(xyz1)
    gx2 = np.empty_like(xyz2)
    lib().oracle_chamfer_bwd(_p(xyz1), _p(xyz2), _p(idx1), _p(idx2), _p(g1), _p(g2),
                             B, N, M, _p(gx1), _p(gx2))
    return gx1, gx2


def chamfer_losses(dist1, dist2, n_first, w_first, w_rest):
    """K1l in the order include/fpsg_hip.h pins: -> float32 [3] = (sum over the first n_first pairs of cd_b, sum over
    the rest, w_first * first + w_rest * rest), cd_b = mean_i dist1[b] + mean_j dist2[b]."""
    dist1, dist2 = _f32(dist1), _f32(dist2)
    B, N = dist1.shape
    M = dist2.shape[1]
    out = np.empty((3,), np.float32)
    fn = lib().oracle_chamfer_losses
    fn.argtypes = [_f32p, _f32p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_float,
                   ctypes.c_float, _f32p]
    fn(_p(dist1), _p(dist2), B, N, M, int(n_first), float(w_first), float(w_rest), _p(out))
    return out


def chamfer_distance_np(p1, p2, w1=1.0, w2=1.0):
    """Kaolin 0.9.0 chamfer_distance: w1*mean_i(dist1) + w2*mean_j(dist2), shape [B].
    The means are taken in float32 by numpy (pairwise summation)."""
    d1, _, d2, _ = chamfer_fwd(p1, p2)
    return (np.float32(w1) * d1.mean(axis=1, dtype=np.float32)
            + np.float32(w2) * d2.mean(axis=1, dtype=np.float32))


# -------------------------------------------------------------- torch shims (CPU only)
def _torch():
    import torch
    return torch


def make_torch_chamfer():
    """Returns a CPU, differentiable ``chamfer_distance(p1, p2, w1=1., w2=1.) -> [B]``
    backed by the C oracle (used to run the model code on CPU as checker/baseline)."""
    torch = _torch()

    class _SidedPair(torch.autograd.Function):
        @staticmethod
        def forward(ctx, p1, p2):
            assert p1.device.type == "cpu" and p2.device.type == "cpu"
            d1, i1, d2, i2 = chamfer_fwd(p1.detach().numpy(), p2.detach().numpy())
            ctx.save_for_backward(p1, p2, torch.from_numpy(i1), torch.from_numpy(i2))
            return torch.from_numpy(d1), torch.from_numpy(d2)

        @staticmethod
        def backward(ctx, g1, g2):
            p1, p2, i1, i2 = ctx.saved_tensors
            gx1, gx2 = chamfer_bwd(p1.detach().numpy(), p2.detach().numpy(), i1.numpy(),
                                   i2.numpy(), g1.contiguous().numpy(),
                                   g2.contiguous().numpy())
            return torch.from_numpy(gx1), torch.from_numpy(gx2)

    def chamfer_distance(p1, p2, w1=1.0, w2=1.0):
        d1, d2 = _SidedPair.apply(p1.contiguous().float(), p2.contiguous().float())
        return w1 * d1.mean(dim=-1) + w2 * d2.mean(dim=-1)

    return chamfer_distance


# ----------------------------------------------------------------------- DGCNN graph ops
def knn(x, k):
    """x [B,C,N] -> int32 [B,N,k] (src/dgcnn/model.py:13-20; arithmetic: fpsg_oracle.c)."""
    x = _f32(x)
    B, C, N = x.shape
    idx = np.empty((B, N, k), np.int32)
    lib().oracle_knn(_p(x), B, C, N, k, _p(idx))
    return idx


def edge_feature(x, idx):
    """x [B,C,N], idx [B,N,k] -> [B,2C,N,k] (src/dgcnn/model.py:23-42)."""
    x, idx = _f32(x), _i32(idx)
    B, C, N = x.shape
    k = idx.shape[2]
    out = np.empty((B, 2 * C, N, k), np.float32)
    lib().oracle_edge_feature(_p(x), _p(idx), B, C, N, k, _p(out))
    return out


def edge_feature_bwd(gout, idx):
    gout, idx = _f32(gout), _i32(idx)
    B, C2, N, k = gout.shape
    gx = np.empty((B, C2 // 2, N), np.float32)
    lib().oracle_edge_feature_bwd(_p(gout), _p(idx), B, C2 // 2, N, k, _p(gx))
    return gx


def in_edge_lists(idx):
    """The order in which oracle_edge_feature_bwd's scatter loop (fpsg_oracle.c, ascending (n, j)) reaches every
    destination, as lists: idx [N,k] of one cloud -> (rev, off) with rev[off[d]:off[d+1]] = the edges e = n*k + j
    whose idx[n,j] == d, ascending e.  Entries outside [0,N) belong to no list.  (The reference reaches the same
    sums through autograd's index backward of src/dgcnn/model.py:30-56, whose atomics fix no order.)"""
    idx = np.asarray(idx)
    N, k = idx.shape
    lists = [[] for _ in range(N)]
    for n in range(N):
        for j in range(k):
            d = int(idx[n, j])
            if 0 <= d < N:
                lists[d].append(n * k + j)
    off = np.zeros(N + 1, np.int32)
    off[1:] = np.cumsum([len(l) for l in lists])
    rev = np.array([e for l in lists for e in l], np.int32)
    return rev, off


# --------------------------------------------------------------------------------- EMD
def emd_approx(xyz1, xyz2, want_grad=False):
    """Approximate-assignment EMD (PARITY UNPINNED, see fpsg_oracle.c): cost [B] and, if
    asked, the gradients w.r.t. both clouds with the soft assignment held constant."""
    xyz1, xyz2 = _f32(xyz1), _f32(xyz2)
    B, N, _ = xyz1.shape
    M = xyz2.shape[1]
    cost = np.empty((B,), np.float32)
    if want_grad:
        g1, g2 = np.empty_like(xyz1), np.empty_like(xyz2)
        lib().oracle_emd_approx(_p(xyz1), _p(xyz2), B, N, M, _p(cost), _p(g1), _p(g2))
        return cost, g1, g2
    lib().oracle_emd_approx(_p(xyz1), _p(xyz2), B, N, M, _p(cost), None, None)
    return cost


def softmin(x, y, h, eps):
    """out[b,i] = -eps * logsumexp_j(h[b,j] - |x_i - y_j|^2 / (2 eps)); x [B,N,3], y [B,M,3], h [B,M]."""
    x, y, h = _f32(x), _f32(y), _f32(h)
    B, N, _ = x.shape
    M = y.shape[1]
    out = np.empty((B, N), np.float32)
    lib().oracle_softmin.argtypes = [_f32p, _f32p, _f32p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                     ctypes.c_float, _f32p]
    lib().oracle_softmin(_p(x), _p(y), _p(h), B, N, M, ctypes.c_float(eps), _p(out))
    return out


def sinkhorn_epsilons(x, y, p=2, blur=0.05, scaling=0.5):
    """geomloss' epsilon schedule: diameter^p, then a geometric descent by scaling^p, then blur^p."""
    pts = np.concatenate([np.asarray(x, np.float64).reshape(-1, 3), np.asarray(y, np.float64).reshape(-1, 3)])
    diameter = float(np.linalg.norm(pts.max(0) - pts.min(0)))
    eps = [diameter ** p]
    eps += [float(np.exp(e)) for e in np.arange(p * np.log(diameter), p * np.log(blur), p * np.log(scaling))]
    eps += [blur ** p]
    return eps


def sinkhorn_divergence(x, y, blur=0.05, scaling=0.5, softmin_fn=None):
    """Debiased Sinkhorn divergence between uniform clouds x [B,N,3], y [B,M,3] -> [B]
    (the symmetric, annealed loop of geomloss.SamplesLoss('sinkhorn', p=2, blur, scaling),
    restated from its published algorithm; PARITY UNPINNED)."""
    sm = softmin_fn or softmin
    x, y = _f32(x), _f32(y)
    B, N, _ = x.shape
    M = y.shape[1]
    a_log = np.full((B, N), -np.log(N), np.float32)
    b_log = np.full((B, M), -np.log(M), np.float32)
    eps_s = sinkhorn_epsilons(x, y, 2, blur, scaling)
    e = eps_s[0]
    a_x, b_y = sm(x, x, a_log, e), sm(y, y, b_log, e)
    a_y, b_x = sm(y, x, a_log, e), sm(x, y, b_log, e)
    for e in eps_s:
        at_y = sm(y, x, a_log + b_x / e, e)
        bt_x = sm(x, y, b_log + a_y / e, e)
        at_x = sm(x, x, a_log + a_x / e, e)
        bt_y = sm(y, y, b_log + b_y / e, e)
        a_y, b_x = 0.5 * (a_y + at_y), 0.5 * (b_x + bt_x)
        a_x, b_y = 0.5 * (a_x + at_x), 0.5 * (b_y + bt_y)
    a_y, b_x = sm(y, x, a_log + b_x / e, e), sm(x, y, b_log + a_y / e, e)
    a_x, b_y = sm(x, x, a_log + a_x / e, e), sm(y, y, b_log + b_y / e, e)
    return ((b_x - a_x).mean(1) + (a_y - b_y).mean(1)).astype(np.float32)


def adam_step(p, g, m, v, t, lr=1e-3, beta1=0.9, beta2=0.999, eps=1e-8):
    """One step of ``torch.optim.Adam`` as the reference configures it (``src/trainNetwork.py:118-123``:
    ``optim.Adam(model.parameters(), lr=opt.lr, betas=(.9, .999))``, no weight decay / amsgrad), in
    float64 numpy, following the algorithm in the class's documentation:
    ``m = b1 m + (1-b1) g;  v = b2 v + (1-b2) g^2;  p -= lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps)``.
    ``t`` counts from 1.  Returns the new ``(p, m, v)``."""
    p, g, m, v = (np.asarray(a, dtype=np.float64) for a in (p, g, m, v))
    m = beta1 * m + (1.0 - beta1) * g
    v = beta2 * v + (1.0 - beta2) * g * g
    step_size = lr / (1.0 - beta1 ** t)
    denom = np.sqrt(v) / np.sqrt(1.0 - beta2 ** t) + eps
    return p - step_size * m / denom, m, v
