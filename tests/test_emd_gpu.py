"""K2 (approximate-assignment EMD) through the C ABI vs its CPU restatement, and bounded
against the exact Hungarian EMD.  Parity with the reference's third-party emd_loss is
UNPINNED (package absent and unversioned) -- these tests pin the solver to its
own specification and to the mathematical truth it approximates."""
import numpy as np
import pytest
import torch

from conftest import unit_ball_clouds

pytestmark = pytest.mark.gpu


def _clouds(seed, B, N, M):
    rng = np.random.default_rng(seed)
    return unit_ball_clouds(rng, B, N), np.tanh(rng.standard_normal((B, M, 3)) * 0.5).astype(np.float32)


@pytest.mark.parametrize("B,N,M", [(2, 256, 256), (1, 1024, 1024), (3, 100, 300), (2, 512, 128), (5, 2048, 2048)])
def test_cost_and_grads_vs_oracle(gpu, oracle, B, N, M):
    from fpsg_amd.metrics import emd_approx
    p1, p2 = _clouds(N + M, B, N, M)
    if N == 2048:
        B, p1, p2 = 1, p1[:1], p2[:1]          # keep the scalar CPU restatement to seconds
    t1 = torch.from_numpy(p1).to(gpu).requires_grad_()
    t2 = torch.from_numpy(p2).to(gpu).requires_grad_()
    cost = emd_approx(t1, t2)
    cost.sum().backward()
    ocost, og1, og2 = oracle.emd_approx(p1, p2, want_grad=True)
    # fp32 tolerance: v_exp_f32 vs libm expf, 16-way split sums vs sequential sums
    np.testing.assert_allclose(cost.detach().cpu().numpy(), ocost, rtol=2e-3)
    s1, s2 = np.abs(og1).max(), np.abs(og2).max()
    np.testing.assert_allclose(t1.grad.cpu().numpy(), og1, rtol=5e-3, atol=5e-3 * s1)
    np.testing.assert_allclose(t2.grad.cpu().numpy(), og2, rtol=5e-3, atol=5e-3 * s2)


@pytest.mark.parametrize("N", [128, 512])
def test_bounded_by_exact_emd(gpu, N):
    from fpsg_amd.metrics import emd_approx
    from oracle.ref_f64 import exact_emd
    p1, p2 = _clouds(N, 3, N, N)
    cost = emd_approx(torch.from_numpy(p1).to(gpu), torch.from_numpy(p2).to(gpu)).cpu().numpy()
    exact = np.array([exact_emd(a, b)[0] for a, b in zip(p1, p2)])
    assert (cost >= exact * 0.999).all() and (cost <= exact * 1.6).all(), (cost, exact)


def test_deterministic_and_self_distance(gpu):
    from fpsg_amd.metrics import emd_approx, emd_loss
    p1, p2 = _clouds(3, 5, 2048, 2048)
    a, b = torch.from_numpy(p1).to(gpu), torch.from_numpy(p2).to(gpu)
    c1, c2 = emd_approx(a, b), emd_approx(a, b)
    assert torch.equal(c1, c2)                                      # no float atomics
    same = emd_approx(a, a.clone())
    assert float(same.max()) < 1e-2 * float(c1.min())              # (near) zero to itself
    assert torch.equal(emd_loss(a, b, reduce="sum"), c1.sum()) and emd_loss(a, b, reduce="sum").dim() == 0


# ----------------------------------------------------------------- Sinkhorn form (sinkhorn=True)
@pytest.mark.parametrize("B,N,M,eps", [(2, 300, 500, 0.5), (1, 2048, 2048, 0.0025), (3, 64, 1000, 0.05),
                                       (2, 1, 7, 0.01), (1, 2500, 2100, 0.02)])
def test_softmin_vs_oracle(gpu, oracle, B, N, M, eps):
    from fpsg_amd.metrics import softmin
    rng = np.random.default_rng(N + M)
    x = unit_ball_clouds(rng, B, N)
    y = np.tanh(rng.standard_normal((B, M, 3)) * 0.5).astype(np.float32)
    h = (rng.standard_normal((B, M)) * 2 - np.log(M)).astype(np.float32)
    got = softmin(torch.from_numpy(x).to(gpu), torch.from_numpy(y).to(gpu), torch.from_numpy(h).to(gpu), eps)
    exp = oracle.softmin(x, y, h, eps)
    # values are eps * O(10): compare on the scale of eps (v_exp_f32/v_log_f32 vs libm)
    np.testing.assert_allclose(got.cpu().numpy(), exp, rtol=2e-5, atol=2e-4 * eps * 10)


def test_sinkhorn_divergence(gpu, oracle):
    """emd_wrapper == emd_loss(sinkhorn=True): the loop against its CPU restatement, against the
    exact optimal-transport cost it approximates (|x-y|^2/2, uniform weights), S(x,x) = 0."""
    from scipy.optimize import linear_sum_assignment
    from fpsg_amd.metrics import sinkhorn_divergence, emd_loss
    from fpsg_amd.utils import emd_wrapper
    rng = np.random.default_rng(8)
    x = unit_ball_clouds(rng, 3, 512)
    y = (unit_ball_clouds(rng, 3, 512) * 0.7 + 0.2).astype(np.float32)
    tx, ty = torch.from_numpy(x).to(gpu), torch.from_numpy(y).to(gpu)
    got = sinkhorn_divergence(tx, ty)
    exp = oracle.sinkhorn_divergence(x, y)
    np.testing.assert_allclose(got.cpu().numpy(), exp, rtol=2e-3)
    for b in range(3):
        C = 0.5 * ((x[b][:, None] - y[b][None]) ** 2).sum(-1)
        r, c = linear_sum_assignment(C)
        ot = C[r, c].mean()
        assert abs(float(got[b]) - ot) <= 0.08 * ot, (float(got[b]), ot)
    assert float(sinkhorn_divergence(tx, tx.clone()).abs().max()) < 1e-6
    assert torch.equal(sinkhorn_divergence(tx, ty), got)                      # deterministic
    assert torch.equal(emd_wrapper(tx, ty), got.sum()) and emd_wrapper(tx, ty).dim() == 0
    assert torch.equal(emd_loss(tx, ty, reduce="none", sinkhorn=True), got)
    big = sinkhorn_divergence(torch.rand(5, 2048, 3, device=gpu), torch.rand(5, 2048, 3, device=gpu) * 0.5)
    assert big.shape == (5,) and bool((big > 0).all())
