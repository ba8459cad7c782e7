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


@pytest.mark.parametrize("B,N,M", [(2, 256, 256), (1, 1024, 1024), (3, 100, 300), (2, 512, 128), (5, 2048, 2048),
                                   (2, 101, 67), (1, 3, 258)])      # rows padded to multiples of 4, clouds shorter than a wave
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
    # fp32 differences: v_exp_f32 vs libm expf, 64-lane tree sums vs sequential sums.  Measured on MI355X over these
    # shapes x 3 seeds (profiles/r03/emd_deviation.txt, tools/measure_emd_deviation.py): cost 5.4e-7 relative;
    # gradients 1.1e-3 of the largest gradient component (N = 2048; 1e-5 ... 1e-4 at N <= 512 -- the matching is soft,
    # a changed last bit of an exponent moves weight between near-equal candidates).  The cost bound is 1e-5 (~80 fp32
    # ulps, 10x under north_star's 1e-4): the last bits of v_exp_f32 and of the DPP tree sums follow the compiler's
    # scheduling, a bound of 3x one build's deviation would trip on a toolchain change without any regression.
    np.testing.assert_allclose(cost.detach().cpu().numpy(), ocost, rtol=1e-5)
    # same-build regression guard: 10x the largest deviation recorded for this build's kernels
    # (profiles/r04/emd_deviation.txt: 5.4e-7) -- the portable bound above stays what a toolchain change is held to
    dev_cost = float(np.max(np.abs(cost.detach().cpu().numpy() - ocost) / np.abs(ocost)))
    print(f"emd_approx B={B} N={N} M={M}: cost deviation {dev_cost:.3e} (recorded max 5.4e-7)")
    assert dev_cost <= 10 * 5.4e-7, dev_cost
    s1, s2 = np.abs(og1).max(), np.abs(og2).max()
    assert np.abs(t1.grad.cpu().numpy() - og1).max() <= 3.5e-3 * s1
    assert np.abs(t2.grad.cpu().numpy() - og2).max() <= 3.5e-3 * s2


@pytest.mark.parametrize("variant", [0, 1, 2, 3])
@pytest.mark.parametrize("B,N,M", [(2, 512, 512), (1, 2048, 2048), (3, 100, 300), (2, 101, 67), (1, 3, 258)])
def test_forward_only_variants_vs_oracle(gpu, oracle, B, N, M, variant):
    """The evaluation path (no gradients) in each of its forms -- the assignment sweep merged with the next level's
    row-normaliser sweep (its exponential the fourth power of the normaliser's), four owner points per wave -- against
    the CPU restatement, which calls expf once per level and pair; and the forms against each other."""
    from fpsg_amd import _hip
    lib = _hip.load()
    p1, p2 = _clouds(7 * N + M, B, N, M)
    t1, t2 = torch.from_numpy(p1).to(gpu), torch.from_numpy(p2).to(gpu)
    ws = torch.empty((lib.fpsg_emd_workspace_floats(B, N, M),), dtype=torch.float32, device=gpu)
    cost = torch.empty((B,), dtype=torch.float32, device=gpu)
    rc = lib.fpsg_emd_approx_variant(t1.data_ptr(), t2.data_ptr(), B, N, M, cost.data_ptr(), None, None, ws.data_ptr(),
                                     variant, torch.cuda.current_stream().cuda_stream)
    assert rc == 0, lib.fpsg_last_error()
    ocost = oracle.emd_approx(p1, p2, want_grad=False)
    # the separate sweeps evaluate the restatement's own expressions (1e-5, as test_cost_and_grads_vs_oracle); the merged
    # ones form exp(l d^2) as (exp(l/4 d^2))^4: 4e-7 per weight, which the auction's clamps amplify on SMALL clouds (measured
    # on MI355X: 2.6e-5 at N = 101 / M = 67, <= 5e-7 at 2048 points) -- half of north_star's 1e-4
    np.testing.assert_allclose(cost.cpu().numpy(), ocost, rtol=5e-5 if variant & 1 else 1e-5)
    base = torch.empty_like(cost)
    rc = lib.fpsg_emd_approx_variant(t1.data_ptr(), t2.data_ptr(), B, N, M, base.data_ptr(), None, None, ws.data_ptr(),
                                     0, torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    np.testing.assert_allclose(cost.cpu().numpy(), base.cpu().numpy(), rtol=5e-5 if variant & 1 else 1e-6)


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
    # largest deviation measured on MI355X: 8.3e-8 (profiles/r03/emd_deviation.txt); bound 1e-5 for the same reason as
    # test_cost_and_grads_vs_oracle's
    np.testing.assert_allclose(got.cpu().numpy(), exp, rtol=1e-5)
    dev_s = float(np.max(np.abs(got.cpu().numpy() - exp) / np.abs(exp)))
    print(f"sinkhorn_divergence: deviation {dev_s:.3e} (recorded max 8.3e-8, profiles/r04/emd_deviation.txt)")
    assert dev_s <= 10 * 8.3e-8, dev_s          # same-build regression guard; sinkhorn.hip is unchanged since round 3
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


def _unfused_loop(x, y, blur=0.05, scaling=0.5):
    """The same loop written out with one fpsg_softmin launch per soft-min and torch elementwise ops (the
    form geomloss itself has); the library's fused loop must reproduce it."""
    import math
    from fpsg_amd.metrics import sinkhorn_epsilons, softmin
    B, N, _ = x.shape
    M = y.size(1)
    pts = torch.cat([x.reshape(-1, 3), y.reshape(-1, 3)])
    eps_s = sinkhorn_epsilons(float((pts.amax(0) - pts.amin(0)).norm()), blur, scaling)
    a_log = torch.full((B, N), -math.log(N), dtype=torch.float32, device=x.device)
    b_log = torch.full((B, M), -math.log(M), dtype=torch.float32, device=x.device)
    e = eps_s[0]
    a_x, b_y = softmin(x, x, a_log, e), softmin(y, y, b_log, e)
    a_y, b_x = softmin(y, x, a_log, e), softmin(x, y, b_log, e)
    for e in eps_s:
        at_y = softmin(y, x, a_log + b_x / e, e)
        bt_x = softmin(x, y, b_log + a_y / e, e)
        at_x = softmin(x, x, a_log + a_x / e, e)
        bt_y = softmin(y, y, b_log + b_y / e, e)
        a_y, b_x = 0.5 * (a_y + at_y), 0.5 * (b_x + bt_x)
        a_x, b_y = 0.5 * (a_x + at_x), 0.5 * (b_y + bt_y)
    a_y, b_x = softmin(y, x, a_log + b_x / e, e), softmin(x, y, b_log + a_y / e, e)
    a_x, b_y = softmin(x, x, a_log + a_x / e, e), softmin(y, y, b_log + b_y / e, e)
    return (b_x - a_x).mean(dim=1) + (a_y - b_y).mean(dim=1)


@pytest.mark.parametrize("B,N,M", [(3, 512, 512), (2, 300, 700), (5, 2048, 2048), (1, 2500, 1000), (2, 1, 9)])
def test_fused_sinkhorn_loop_equals_unfused(gpu, B, N, M):
    from fpsg_amd.metrics import sinkhorn_divergence
    rng = np.random.default_rng(N * 3 + M)
    x = torch.from_numpy(unit_ball_clouds(rng, B, N)).to(gpu)
    y = torch.from_numpy((unit_ball_clouds(rng, B, M) * 0.6 + 0.25).astype(np.float32)).to(gpu)
    got = sinkhorn_divergence(x, y)
    exp = _unfused_loop(x, y)
    np.testing.assert_allclose(got.cpu().numpy(), exp.cpu().numpy(), rtol=5e-5, atol=1e-7)


def test_sinkhorn_vs_independent_float64(gpu):
    """Against the independent float64 restatement of geomloss' published loop (explicit [N,M] matrices,
    scipy logsumexp): fp32 soft-mins with v_exp_f32 / v_log_f32 stay within 1e-4 of it."""
    from oracle.sinkhorn_f64 import sinkhorn_divergence_f64
    from fpsg_amd.metrics import sinkhorn_divergence
    rng = np.random.default_rng(31)
    x = unit_ball_clouds(rng, 2, 700)
    y = np.tanh(rng.standard_normal((2, 600, 3)) * 0.4).astype(np.float32)
    got = sinkhorn_divergence(torch.from_numpy(x).to(gpu), torch.from_numpy(y).to(gpu)).cpu().numpy()
    exp = sinkhorn_divergence_f64(x, y)
    rel = np.abs(got - exp) / exp
    print("Sinkhorn divergence: HIP", got, "float64", exp, "relative deviation", rel)
    assert rel.max() <= 1e-4, rel
