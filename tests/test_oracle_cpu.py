"""The oracle itself, pinned: Kaolin docstring known-answer, float64 brute force, autograd."""
import json
import os

import numpy as np
import torch

from conftest import GOLDEN, unit_ball_clouds


def test_kaolin_known_answer(oracle):
    kat = json.load(open(os.path.join(GOLDEN, "kaolin_chamfer_kat.json")))
    out = oracle.chamfer_distance_np(np.array(kat["p1"], np.float32), np.array(kat["p2"], np.float32))
    np.testing.assert_allclose(out, kat["expected"], rtol=2e-6)


def test_sided_distance_vs_float64(oracle):
    from oracle.ref_f64 import chamfer_f64
    rng = np.random.default_rng(3)
    p1 = unit_ball_clouds(rng, 3, 700)
    p2 = np.tanh(rng.standard_normal((3, 513, 3))).astype(np.float32)
    d1, i1, d2, i2 = oracle.chamfer_fwd(p1, p2)
    cd, fd1, fi1, fd2, fi2 = chamfer_f64(p1, p2)
    assert (i1 == fi1).mean() > 0.999 and (i2 == fi2).mean() > 0.999  # fp32 near-ties only
    np.testing.assert_allclose(d1, fd1, rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(d2, fd2, rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(oracle.chamfer_distance_np(p1, p2), cd, rtol=1e-5)


def test_tie_rule_first_minimum(oracle):
    p1 = np.zeros((1, 3, 3), np.float32)
    p2 = np.array([[[1, 0, 0], [0, 1, 0], [0, 0, 1], [1, 0, 0]]], np.float32)
    d, i = oracle.sided_distance(p1, p2)
    assert (i == 0).all() and (d == 1).all()


def test_backward_vs_float64_autograd(oracle):
    rng = np.random.default_rng(5)
    a = unit_ball_clouds(rng, 2, 300)
    b = np.tanh(rng.standard_normal((2, 411, 3))).astype(np.float32)
    cd = oracle.make_torch_chamfer()
    ta = torch.tensor(a, requires_grad=True)
    tb = torch.tensor(b, requires_grad=True)
    w = torch.tensor([0.3, 1.7])
    (cd(ta, tb) * w).sum().backward()
    a64 = torch.tensor(a, dtype=torch.float64, requires_grad=True)
    b64 = torch.tensor(b, dtype=torch.float64, requires_grad=True)
    D = ((a64[:, :, None] - b64[:, None]) ** 2).sum(-1)
    ((D.min(2)[0].mean(1) + D.min(1)[0].mean(1)) * w.double()).sum().backward()
    np.testing.assert_allclose(ta.grad.numpy(), a64.grad.numpy(), rtol=1e-4, atol=1e-9)
    np.testing.assert_allclose(tb.grad.numpy(), b64.grad.numpy(), rtol=1e-4, atol=1e-9)


def test_oracle_properties_hypothesis(oracle):
    """Size-independent properties of the Chamfer restatement on random ragged shapes:
    permutation equivariance, exact power-of-two scaling, argument symmetry, and that every
    reported distance is the distance to the reported index."""
    from hypothesis import given, settings, strategies as st

    @settings(max_examples=25, deadline=None)
    @given(st.integers(1, 3), st.integers(1, 70), st.integers(1, 90), st.integers(0, 10**6))
    def check(B, N, M, seed):
        rng = np.random.default_rng(seed)
        a = rng.standard_normal((B, N, 3)).astype(np.float32)
        b = rng.standard_normal((B, M, 3)).astype(np.float32)
        d1, i1, d2, i2 = oracle.chamfer_fwd(a, b)
        # symmetry of the two directions
        e1, j1, e2, j2 = oracle.chamfer_fwd(b, a)
        assert np.array_equal(d1, e2) and np.array_equal(d2, e1) and np.array_equal(i1, j2)
        # scaling by 2 is exact in fp32: distances x4, same indices
        s1, k1, s2, k2 = oracle.chamfer_fwd(2 * a, 2 * b)
        assert np.array_equal(s1, 4 * d1) and np.array_equal(k1, i1) and np.array_equal(k2, i2)
        # permuting the candidates permutes the indices
        perm = rng.permutation(M)
        p1, q1, _, _ = oracle.chamfer_fwd(a, b[:, perm])
        assert np.array_equal(p1, d1)
        assert np.array_equal(np.take_along_axis(b[:, perm], q1[..., None].repeat(3, -1), 1),
                              np.take_along_axis(b, i1[..., None].repeat(3, -1), 1))
        # distance to the reported index
        nb = np.take_along_axis(b, i1[..., None].repeat(3, -1), 1)
        np.testing.assert_allclose(((a - nb) ** 2).sum(-1), d1, rtol=1e-5, atol=1e-7)

    check()


def test_sinkhorn_restatement_vs_independent_float64(oracle):
    """The float32 / O(N+M) restatement used as the HIP kernels' checker against an independent float64
    restatement of the same published geomloss loop with explicit [N,M] matrices (oracle/sinkhorn_f64.py)."""
    from oracle.sinkhorn_f64 import epsilon_schedule, sinkhorn_divergence_f64
    rng = np.random.default_rng(8)
    x = unit_ball_clouds(rng, 2, 300)
    y = (unit_ball_clouds(rng, 2, 400) * 0.7 + 0.2).astype(np.float32)
    a = oracle.sinkhorn_divergence(x, y)
    b = sinkhorn_divergence_f64(x, y)
    np.testing.assert_allclose(a, b, rtol=1e-5)
    assert np.allclose(oracle.sinkhorn_epsilons(x, y), epsilon_schedule(2, np.linalg.norm(
        np.concatenate([x.reshape(-1, 3), y.reshape(-1, 3)]).max(0) - np.concatenate([x.reshape(-1, 3), y.reshape(-1, 3)]).min(0)), 0.05, 0.5))
    # the divergence of a cloud with itself is zero, and it is symmetric
    assert abs(sinkhorn_divergence_f64(x, x)).max() < 1e-12
    np.testing.assert_allclose(sinkhorn_divergence_f64(x, y), sinkhorn_divergence_f64(y, x), rtol=1e-10)


def test_in_edge_lists_are_the_stable_grouping_of_the_edges():
    """oracle.in_edge_lists (the visiting order of the oracle's sequential scatter) = a stable sort of the edge
    numbers by destination; entries outside [0, N) belong to no list."""
    import oracle
    rng = np.random.default_rng(3)
    idx = rng.integers(0, 50, size=(50, 7)).astype(np.int32)
    idx[4, 2] = -1
    idx[9, 0] = 50
    rev, off = oracle.in_edge_lists(idx)
    flat = idx.reshape(-1)
    valid = (flat >= 0) & (flat < 50)
    order = np.argsort(np.where(valid, flat, 50), kind="stable")[: int(valid.sum())]
    assert np.array_equal(rev, order.astype(np.int32))
    assert off[0] == 0 and off[-1] == valid.sum()
    assert np.array_equal(np.diff(off), np.bincount(flat[valid], minlength=50))
    # and it is the order in which oracle_edge_feature_bwd adds: one channel, gradient 1 on the "difference" half of
    # edge e only -> the destination's sum is 1 exactly for the edges in its list
    gx_lists = [list(rev[off[d]:off[d + 1]]) for d in range(50)]
    assert sorted(e for l in gx_lists for e in l) == sorted(np.nonzero(valid)[0].tolist())
    assert all(l == sorted(l) for l in gx_lists)


def test_chamfer_losses_order_is_a_valid_sum(oracle):
    """oracle_chamfer_losses (the summation order include/fpsg_hip.h pins for K1l) against float64 sums, ragged rows
    (a last block of 45 values, a row shorter than one group of 64) included."""
    rng = np.random.default_rng(3)
    for B, N, M, n_first in ((5, 2048, 2048, 2), (3, 301, 40, 0), (4, 1, 700, 4)):
        d1 = rng.random((B, N)).astype(np.float32)
        d2 = rng.random((B, M)).astype(np.float32)
        out = oracle.chamfer_losses(d1, d2, n_first, 1.0, 0.75)
        cd = d1.astype(np.float64).mean(1) + d2.astype(np.float64).mean(1)
        want = np.array([cd[:n_first].sum(), cd[n_first:].sum(), cd[:n_first].sum() + 0.75 * cd[n_first:].sum()])
        np.testing.assert_allclose(out, want, rtol=3e-6, atol=1e-7)
    ones = np.ones((2, 1024), np.float32)
    assert np.array_equal(oracle.chamfer_losses(ones, ones, 1, 2.0, 3.0), np.array([2.0, 2.0, 10.0], np.float32))


def test_graph_ops_match_reference_goldens(oracle):
    """The oracle's kNN graph and edge features against the REFERENCE functions' own outputs (``dgcnn.model.knn`` /
    ``get_graph_feature``, ``/root/reference/src/dgcnn/model.py:13-42``, captured by ``tests/golden/make_golden.py``):
    the pin that makes "HIP == oracle" on the GPU box mean "HIP == reference".  Equal at every position on all four
    kNN shapes and bit for bit on the edge features (measured; a change to ``oracle/fpsg_oracle.c`` that breaks either
    goes red here, on the CPU)."""
    gold = np.load(os.path.join(GOLDEN, "dgcnn_goldens.npz"))
    for tag in ("c3_n256", "c3_n2048", "c64_n256", "c64_n2048"):
        x, ref = gold[f"knn_x_{tag}"], gold[f"knn_idx_{tag}"]
        got = oracle.knn(x, ref.shape[2])
        assert got.shape == ref.shape and got.dtype == np.int32
        assert np.array_equal(got, ref), (tag, float((got == ref).mean()))
        assert np.array_equal(got[:, :, 0], np.broadcast_to(np.arange(x.shape[2]), got.shape[:2]))   # self first
    x, ref = gold["ggf_x"], gold["ggf_out"]
    out = oracle.edge_feature(x, oracle.knn(x, ref.shape[3]))
    assert out.shape == ref.shape and np.array_equal(out, ref)
