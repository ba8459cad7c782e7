"""K1 parity: HIP Chamfer (through the C ABI) vs the CPU oracle -- bit-exact distances
and indices, gradients bit-exact; plus size-independent properties at full size."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, unit_ball_clouds

pytestmark = pytest.mark.gpu


def _run_fwd(p1, p2, dev):
    from fpsg_amd.metrics import sided_distances
    t1 = torch.from_numpy(p1).to(dev)
    t2 = torch.from_numpy(p2).to(dev)
    d1, i1, d2, i2 = sided_distances(t1, t2)
    return d1.cpu().numpy(), i1.cpu().numpy(), d2.cpu().numpy(), i2.cpu().numpy()


# (B, N, M): covers every (R, W) launch configuration, ragged sizes, N != M, tiny clouds,
# clouds larger than one LDS tile (4096) and sizes that are not multiples of 16.
SHAPES = [(1, 2048, 2048), (5, 2048, 2048), (32, 2048, 2048), (2, 1024, 1024), (3, 1, 1),
          (2, 7, 1), (2, 1, 50), (3, 333, 1000), (2, 4097, 5000), (1, 9000, 100),
          (70, 512, 300), (130, 256, 256)]


@pytest.mark.parametrize("B,N,M", SHAPES)
def test_fwd_bit_exact_vs_oracle(gpu, oracle, B, N, M):
    rng = np.random.default_rng(B * 1000003 + N * 131 + M)
    p1 = unit_ball_clouds(rng, B, N)
    p2 = np.tanh(rng.standard_normal((B, M, 3))).astype(np.float32)  # decoder-range stand-in
    d1, i1, d2, i2 = _run_fwd(p1, p2, gpu)
    od1, oi1, od2, oi2 = oracle.chamfer_fwd(p1, p2)
    assert np.array_equal(i1, oi1) and np.array_equal(i2, oi2)
    assert np.array_equal(d1.view(np.uint32), od1.view(np.uint32))
    assert np.array_equal(d2.view(np.uint32), od2.view(np.uint32))


def _call_fwd(lib, name, p1, p2, dev, *extra):
    B, N, M = p1.shape[0], p1.shape[1], p2.shape[1]
    t1, t2 = torch.from_numpy(p1).to(dev), torch.from_numpy(p2).to(dev)
    d1 = torch.empty((B, N), device=dev); d2 = torch.empty((B, M), device=dev)
    i1 = torch.empty((B, N), device=dev, dtype=torch.int32); i2 = torch.empty((B, M), device=dev, dtype=torch.int32)
    rc = getattr(lib, name)(t1.data_ptr(), t2.data_ptr(), B, N, M, d1.data_ptr(), i1.data_ptr(), d2.data_ptr(),
                            i2.data_ptr(), *extra, torch.cuda.current_stream().cuda_stream)
    assert rc == 0, lib.fpsg_last_error()
    return d1.cpu().numpy(), i1.cpu().numpy(), d2.cpu().numpy(), i2.cpu().numpy()


def _assert_fwd_equal(got, exp):
    d1, i1, d2, i2 = got
    od1, oi1, od2, oi2 = exp
    assert np.array_equal(i1, oi1) and np.array_equal(i2, oi2)
    assert np.array_equal(d1.view(np.uint32), od1.view(np.uint32))
    assert np.array_equal(d2.view(np.uint32), od2.view(np.uint32))


@pytest.mark.parametrize("cfg", range(7))
def test_every_two_pass_variant_is_bit_exact(gpu, oracle, cfg):
    """All (queries per lane, waves per workgroup) variants of the two-pass forward kernel, selected by
    explicit argument, on ragged sizes (tail lanes, partial chunks, N != M)."""
    from fpsg_amd import _hip
    rng = np.random.default_rng(100 + cfg)
    p1 = unit_ball_clouds(rng, 3, 1000)
    p2 = np.tanh(rng.standard_normal((3, 777, 3))).astype(np.float32)
    _assert_fwd_equal(_call_fwd(_hip.load(), "fpsg_chamfer_fwd_variant", p1, p2, gpu, cfg), oracle.chamfer_fwd(p1, p2))


# (R==8 ? 100 : 0) + 10 * W + cpw: every template instance, several tile widths
TILED_VARIANTS = [111, 121, 141, 144, 148, 122, 11, 21, 41, 44, 49, 14]


@pytest.mark.parametrize("variant", TILED_VARIANTS)
@pytest.mark.parametrize("B,N,M", [(3, 1000, 777), (2, 2048, 2048), (2, 1, 50), (2, 50, 1), (1, 4096, 3000), (2, 129, 4096)])
def test_every_tiled_variant_is_bit_exact(gpu, oracle, variant, B, N, M):
    """The one-pass tiled forward (each d(i,j) evaluated once, both directions served) for every tile
    shape: ragged rows / candidates, N != M, single points, the 4096-point limit."""
    from fpsg_amd import _hip
    lib = _hip.load()
    rng = np.random.default_rng(variant * 7 + N + M)
    p1 = unit_ball_clouds(rng, B, N)
    p2 = np.tanh(rng.standard_normal((B, M, 3))).astype(np.float32)
    nbytes = lib.fpsg_chamfer_workspace_bytes(B, N, M, variant)
    assert nbytes > 0
    ws = torch.empty((nbytes,), dtype=torch.uint8, device=gpu)
    got = _call_fwd(lib, "fpsg_chamfer_fwd_tiled", p1, p2, gpu, ws.data_ptr(), nbytes, variant)
    _assert_fwd_equal(got, oracle.chamfer_fwd(p1, p2))


def test_tiled_ties_and_duplicates(gpu, oracle):
    """Exact ties across lanes, segments, tiles and workgroups: lattice points and repeats
    (src/datasets/modelnet.py:61-64 pads short clouds with repeats) -- the lowest index must win on
    both sides in every tile shape."""
    from fpsg_amd import _hip
    lib = _hip.load()
    rng = np.random.default_rng(17)
    base = rng.integers(-3, 4, size=(3, 700, 3)).astype(np.float32)
    p1 = np.concatenate([base, base[:, :500], base[:, 100:300]], axis=1)      # 1400 with repeats
    p2 = np.concatenate([base[:, ::-1], base[:, :333]], axis=1)               # 1033
    exp = oracle.chamfer_fwd(p1, p2)
    for variant in (141, 144, 41, 11, 122):
        nbytes = lib.fpsg_chamfer_workspace_bytes(3, 1400, 1033, variant)
        ws = torch.empty((nbytes,), dtype=torch.uint8, device=gpu)
        _assert_fwd_equal(_call_fwd(lib, "fpsg_chamfer_fwd_tiled", p1, p2, gpu, ws.data_ptr(), nbytes, variant), exp)


def test_tiled_near_ties_across_tiles(gpu, oracle):
    """The one-pass forward's partial keys are a tile's minimum with its six lowest mantissa bits replaced by a position;
    the second launch must re-evaluate EVERY tile whose truncated minimum equals the smallest one.  Here every point has
    two nearest candidates in different tiles whose distances differ by a few ulps (either way round, or not at all)."""
    from fpsg_amd import _hip
    lib = _hip.load()
    rng = np.random.default_rng(23)
    n = 600
    # points on an exactly representable lattice, offsets of 14-bit multiples of 2^-23: p + delta is exact, so the two
    # candidates p + (a,b,c) and p + (c,a,b) are at the same true distance and their fp32 distances -- rounded in a
    # different order -- differ by an ulp or two, in either direction, for about a fifth of the points
    lat = np.stack(np.meshgrid(np.arange(8) * 0.25 - 1.0, np.arange(8) * 0.25 - 1.0, np.arange(10) * 0.125 - 0.5,
                               indexing="ij"), -1).reshape(-1, 3)[:n]
    p1 = np.stack([lat[rng.permutation(n)] for _ in range(3)]).astype(np.float32)
    delta = (rng.integers(-(1 << 14), 1 << 14, size=(3, n, 3)) * 2.0 ** -23).astype(np.float32)
    p2 = np.concatenate([p1 + delta, p1 + np.roll(delta, 1, axis=-1)], axis=1).astype(np.float32)   # 1200 candidates
    for a, b in ((p1, p2), (p2, p1)):
        exp = oracle.chamfer_fwd(a, b)
        for variant in (141, 144, 41, 11, 122, 148):
            nbytes = lib.fpsg_chamfer_workspace_bytes(3, a.shape[1], b.shape[1], variant)
            ws = torch.empty((nbytes,), dtype=torch.uint8, device=gpu)
            _assert_fwd_equal(_call_fwd(lib, "fpsg_chamfer_fwd_tiled", a, b, gpu, ws.data_ptr(), nbytes, variant), exp)
    # the construction does produce candidates in different tiles whose distances agree above the six lowest mantissa
    # bits and differ below them, both ways round
    da = oracle.chamfer_fwd(p1, p2[:, :n])[0].view(np.uint32)
    db = oracle.chamfer_fwd(p1, p2[:, n:])[0].view(np.uint32)
    assert ((da & ~np.uint32(63)) == (db & ~np.uint32(63))).mean() > 0.9
    assert (da < db).mean() > 0.05 and (da > db).mean() > 0.05


def test_tiled_refuses_what_it_cannot_do(gpu):
    from fpsg_amd import _hip
    lib = _hip.load()
    assert lib.fpsg_chamfer_workspace_bytes(2, 5000, 100, -1) == 0          # beyond 4096 points: two-pass kernel
    assert lib.fpsg_chamfer_workspace_bytes(2, 2048, 2048, -1) == 0         # few pairs: the two-pass kernel is faster
    assert lib.fpsg_chamfer_workspace_bytes(37, 2048, 2048, -1) > 0
    assert lib.fpsg_chamfer_workspace_bytes(2, 100, 100, 131) == 0           # W = 3 is not a variant
    p = torch.rand(2, 64, 3, device=gpu)
    d = torch.empty(2, 64, device=gpu); i = torch.empty(2, 64, device=gpu, dtype=torch.int32)
    ws = torch.empty(8, dtype=torch.uint8, device=gpu)
    rc = lib.fpsg_chamfer_fwd_tiled(p.data_ptr(), p.data_ptr(), 2, 64, 64, d.data_ptr(), i.data_ptr(), d.data_ptr(),
                                    i.data_ptr(), ws.data_ptr(), 8, 141, None)
    assert rc == -4 and b"workspace" in lib.fpsg_last_error()


def test_ties_pick_lowest_index(gpu, oracle):
    """Duplicated points (the reference pads short clouds with repeats,
    src/datasets/modelnet.py:61-64) and lattice points give exact ties."""
    rng = np.random.default_rng(7)
    base = rng.integers(-3, 4, size=(4, 300, 3)).astype(np.float32)  # many exact ties
    p1 = np.concatenate([base, base[:, :212]], axis=1)                # 512 with repeats
    p2 = np.concatenate([base[:, ::-1], base[:, :100]], axis=1)
    d1, i1, d2, i2 = _run_fwd(p1, p2, gpu)
    od1, oi1, od2, oi2 = oracle.chamfer_fwd(p1, p2)
    assert np.array_equal(i1, oi1) and np.array_equal(i2, oi2)
    assert np.array_equal(d1, od1) and np.array_equal(d2, od2)


def test_kaolin_docstring_known_answer(gpu):
    from fpsg_amd.metrics import chamfer_distance
    kat = json.load(open(os.path.join(GOLDEN, "kaolin_chamfer_kat.json")))
    p1 = torch.tensor(kat["p1"], dtype=torch.float32, device=gpu)
    p2 = torch.tensor(kat["p2"], dtype=torch.float32, device=gpu)
    out = chamfer_distance(p1, p2).cpu().numpy()
    np.testing.assert_allclose(out, np.array(kat["expected"]), rtol=1e-4)  # north_star tol


def _call_bwd(lib, name, p1, p2, i1, i2, g1, g2, dev):
    B, N, M = p1.shape[0], p1.shape[1], p2.shape[1]
    t = lambda a, dt=None: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    tp1, tp2, ti1, ti2, tg1, tg2 = t(p1), t(p2), t(i1.astype(np.int32)), t(i2.astype(np.int32)), t(g1), t(g2)
    gx1, gx2 = torch.empty_like(tp1), torch.empty_like(tp2)
    rc = getattr(lib, name)(tp1.data_ptr(), tp2.data_ptr(), ti1.data_ptr(), ti2.data_ptr(), tg1.data_ptr(),
                            tg2.data_ptr(), B, N, M, gx1.data_ptr(), gx2.data_ptr(),
                            torch.cuda.current_stream().cuda_stream)
    assert rc == 0, lib.fpsg_last_error()
    return gx1.cpu().numpy(), gx2.cpu().numpy()


@pytest.mark.parametrize("kernel", ["fpsg_chamfer_bwd", "fpsg_chamfer_bwd_sorted", "fpsg_chamfer_bwd_scan"])
@pytest.mark.parametrize("B,N,M", [(1, 2048, 2048), (5, 2048, 2048), (3, 333, 1000), (2, 4096, 3000),
                                   (2, 4097, 5000), (2, 1, 50), (3, 50, 1)])
def test_bwd_bit_exact_vs_oracle(gpu, oracle, kernel, B, N, M):
    """Both backward kernels (sorted inversion for clouds up to 4096 points, tile scan for any size) against
    the oracle's sequential order: own term, then the sources in ascending index."""
    from fpsg_amd import _hip
    if kernel == "fpsg_chamfer_bwd_sorted" and max(N, M) > 4096:
        pytest.skip("the sorted kernel serves clouds of at most 4096 points")
    rng = np.random.default_rng(B + N + M)
    p1 = unit_ball_clouds(rng, B, N)
    p2 = np.tanh(rng.standard_normal((B, M, 3))).astype(np.float32)
    g1 = rng.standard_normal((B, N)).astype(np.float32)
    g2 = rng.standard_normal((B, M)).astype(np.float32)
    _, oi1, _, oi2 = oracle.chamfer_fwd(p1, p2)
    gx1, gx2 = _call_bwd(_hip.load(), kernel, p1, p2, oi1, oi2, g1, g2, gpu)
    ogx1, ogx2 = oracle.chamfer_bwd(p1, p2, oi1, oi2, g1, g2)
    assert np.array_equal(gx1.view(np.uint32), ogx1.view(np.uint32))
    assert np.array_equal(gx2.view(np.uint32), ogx2.view(np.uint32))


def test_autograd_path_is_bit_exact(gpu, oracle):
    from fpsg_amd.metrics import _SidedPair
    rng = np.random.default_rng(9)
    p1 = unit_ball_clouds(rng, 5, 2048)
    p2 = np.tanh(rng.standard_normal((5, 2048, 3))).astype(np.float32)
    g1 = rng.standard_normal((5, 2048)).astype(np.float32)
    g2 = rng.standard_normal((5, 2048)).astype(np.float32)
    t1 = torch.from_numpy(p1).to(gpu).requires_grad_()
    t2 = torch.from_numpy(p2).to(gpu).requires_grad_()
    d1, d2, i1, i2 = _SidedPair.apply(t1, t2)
    torch.autograd.backward([d1, d2], [torch.from_numpy(g1).to(gpu), torch.from_numpy(g2).to(gpu)])
    od1, oi1, od2, oi2 = oracle.chamfer_fwd(p1, p2)
    assert np.array_equal(i1.cpu().numpy(), oi1) and np.array_equal(d2.detach().cpu().numpy(), od2)
    ogx1, ogx2 = oracle.chamfer_bwd(p1, p2, oi1, oi2, g1, g2)
    assert np.array_equal(t1.grad.cpu().numpy().view(np.uint32), ogx1.view(np.uint32))
    assert np.array_equal(t2.grad.cpu().numpy().view(np.uint32), ogx2.view(np.uint32))


@pytest.mark.parametrize("kernel", ["fpsg_chamfer_bwd_sorted", "fpsg_chamfer_bwd_scan"])
def test_bwd_many_sources_per_target(gpu, oracle, kernel):
    """Duplicate-heavy clouds: some points are the nearest neighbour of hundreds of others (routine early in
    training, when the generated cloud is a small blob)."""
    from fpsg_amd import _hip
    rng = np.random.default_rng(21)
    base = rng.integers(-2, 3, size=(3, 40, 3)).astype(np.float32)
    p1 = np.repeat(base, 30, axis=1)[:, :1100]                        # 30 copies of each point
    p2 = np.concatenate([base[:, :7], rng.standard_normal((3, 900, 3)).astype(np.float32) * 5], axis=1)
    g1 = rng.standard_normal(p1.shape[:2]).astype(np.float32)
    g2 = rng.standard_normal(p2.shape[:2]).astype(np.float32)
    _, oi1, _, oi2 = oracle.chamfer_fwd(p1, p2)
    assert np.bincount(oi1.reshape(-1)).max() > 100                    # really many-to-one
    gx1, gx2 = _call_bwd(_hip.load(), kernel, p1, p2, oi1, oi2, g1, g2, gpu)
    ogx1, ogx2 = oracle.chamfer_bwd(p1, p2, oi1, oi2, g1, g2)
    assert np.array_equal(gx1.view(np.uint32), ogx1.view(np.uint32))
    assert np.array_equal(gx2.view(np.uint32), ogx2.view(np.uint32))


def test_bwd_all_sources_on_one_target(gpu, oracle):
    """The extreme: every point of one cloud chooses the same target (a collapsed generated cloud)."""
    from fpsg_amd import _hip
    rng = np.random.default_rng(22)
    p1 = unit_ball_clouds(rng, 2, 2048)
    p2 = np.zeros((2, 2048, 3), np.float32) + 5.0
    p2[:, 77] = 0.0                                                    # the one point near cloud 1
    g1 = rng.standard_normal((2, 2048)).astype(np.float32)
    g2 = rng.standard_normal((2, 2048)).astype(np.float32)
    _, oi1, _, oi2 = oracle.chamfer_fwd(p1, p2)
    assert (oi1 == 77).all()
    ogx1, ogx2 = oracle.chamfer_bwd(p1, p2, oi1, oi2, g1, g2)
    for kernel in ("fpsg_chamfer_bwd_sorted", "fpsg_chamfer_bwd_scan"):
        gx1, gx2 = _call_bwd(_hip.load(), kernel, p1, p2, oi1, oi2, g1, g2, gpu)
        assert np.array_equal(gx1.view(np.uint32), ogx1.view(np.uint32))
        assert np.array_equal(gx2.view(np.uint32), ogx2.view(np.uint32))


def test_chamfer_value_and_grad_vs_float64(gpu):
    """Within north_star's 1e-4 relative fp32 tolerance of the float64 definition."""
    from fpsg_amd.metrics import chamfer_distance
    from oracle.ref_f64 import chamfer_f64
    rng = np.random.default_rng(11)
    p1 = unit_ball_clouds(rng, 4, 2048)
    p2 = np.tanh(rng.standard_normal((4, 2048, 3))).astype(np.float32)
    t1 = torch.from_numpy(p1).to(gpu).requires_grad_()
    t2 = torch.from_numpy(p2).to(gpu).requires_grad_()
    cd = chamfer_distance(t1, t2)
    cd.sum().backward()
    ref = chamfer_f64(p1, p2)[0]
    np.testing.assert_allclose(cd.detach().cpu().numpy(), ref, rtol=1e-4)
    a = torch.tensor(p1, dtype=torch.float64, requires_grad=True)
    b = torch.tensor(p2, dtype=torch.float64, requires_grad=True)
    D = ((a[:, :, None] - b[:, None]) ** 2).sum(-1)
    (D.min(2)[0].mean(1) + D.min(1)[0].mean(1)).sum().backward()
    np.testing.assert_allclose(t1.grad.cpu().numpy(), a.grad.numpy(), rtol=1e-4, atol=1e-9)
    np.testing.assert_allclose(t2.grad.cpu().numpy(), b.grad.numpy(), rtol=1e-4, atol=1e-9)


@pytest.mark.parametrize("B,N,M,n_first", [(37, 2048, 2048, 5), (7, 301, 258, 3), (3, 64, 128, 0), (4, 100, 100, 4),
                                           (64, 301, 258, 10), (9, 4096, 3000, 9), (2, 5000, 700, 1), (300, 64, 80, 120), (1100, 40, 40, 7)])
def test_episode_losses_equal_the_separate_operations(gpu, oracle, B, N, M, n_first):
    """K1l: the query sum, the support sum and the weighted total of few_shot.py:110-124 -- fused into the one-pass
    forward (fpsg_chamfer_fwd_tiled_losses: from ~7 pairs of 2048 points up) or one launch behind the two-pass forward
    (fpsg_chamfer_losses) -- bit for bit the oracle's sums in the order include/fpsg_hip.h pins, within fp32
    summation-order noise of the PyTorch operation chain and of float64 sums; the cloud gradients
    (fpsg_chamfer_bwd_losses: the per-pair constants formed inside the backward kernel) bit-identical to the chain's
    for the total, also when the gradient arrives through one of the partial sums."""
    from fpsg_amd.metrics import chamfer_distance, episode_chamfer_losses
    rng = np.random.default_rng(B * 1000 + N)
    p1 = unit_ball_clouds(rng, B, N)
    p2 = np.tanh(rng.standard_normal((B, M, 3))).astype(np.float32)
    wq, ws = 1.0, 0.75
    a1 = torch.from_numpy(p1).to(gpu).requires_grad_()
    a2 = torch.from_numpy(p2).to(gpu).requires_grad_()
    q, s, total = episode_chamfer_losses(a1, a2, n_first, wq, ws)
    assert q.dim() == 0 and s.dim() == 0 and total.dim() == 0
    b1 = torch.from_numpy(p1).to(gpu).requires_grad_()
    b2 = torch.from_numpy(p2).to(gpu).requires_grad_()
    cd = chamfer_distance(b1, b2)
    rq, rs = cd[:n_first].sum(), cd[n_first:].sum()
    rtotal = wq * rq + ws * rs
    for got, want in ((q, rq), (s, rs), (total, rtotal)):
        assert abs(got.item() - want.item()) <= 2e-6 * max(1.0, abs(want.item()))
    od1, _, od2, _ = oracle.chamfer_fwd(p1, p2)
    want3 = oracle.chamfer_losses(od1, od2, n_first, wq, ws)
    got3 = np.array([q.item(), s.item(), total.item()], np.float32)
    assert np.array_equal(got3.view(np.uint32), want3.view(np.uint32)), (got3, want3)
    ocd = od1.astype(np.float64).mean(1) + od2.astype(np.float64).mean(1)
    assert abs(q.item() - ocd[:n_first].sum()) <= 2e-6 * max(1.0, ocd[:n_first].sum())
    assert abs(total.item() - (wq * ocd[:n_first].sum() + ws * ocd[n_first:].sum())) <= 2e-6 * max(1.0, ocd.sum())
    total.backward()
    rtotal.backward()
    assert torch.equal(a1.grad, b1.grad) and torch.equal(a2.grad, b2.grad)
    # a gradient through the partial sums only
    a1.grad = a2.grad = b1.grad = b2.grad = None
    q, s, total = episode_chamfer_losses(a1, a2, n_first, wq, ws)
    (2.0 * q + 3.0 * s).backward()
    cd = chamfer_distance(b1, b2)
    (2.0 * cd[:n_first].sum() + 3.0 * cd[n_first:].sum()).backward()
    assert torch.equal(a1.grad, b1.grad) and torch.equal(a2.grad, b2.grad)


def test_full_size_properties(gpu):
    """BASELINE.json sizes (B=37 = 32-shot + 5-query clouds of 2048 points), checked
    through properties that need no oracle: self-distance is exactly zero with identity
    argmin, permutation of the candidate cloud permutes indices and keeps distances
    bit-identical, the two directions swap bit for bit when the arguments swap, and every
    reported distance is the distance to the reported index."""
    from fpsg_amd.metrics import sided_distances, chamfer_distance
    g = torch.Generator(device="cpu").manual_seed(5)
    B, N = 37, 2048
    p1 = (torch.rand(B, N, 3, generator=g) * 2 - 1).to(gpu)
    p2 = torch.tanh(torch.randn(B, N, 3, generator=g)).to(gpu)
    d1, i1, d2, i2 = sided_distances(p1, p1.clone())
    assert torch.count_nonzero(d1) == 0 and torch.count_nonzero(d2) == 0
    ar = torch.arange(N, device=gpu).expand(B, N)
    assert torch.equal(i1, ar) and torch.equal(i2, ar)
    # permutation equivariance (distinct random points => no ties)
    perm = torch.randperm(N, generator=g).to(gpu)
    a1, ai1, a2, ai2 = sided_distances(p1, p2)
    b1, bi1, b2, bi2 = sided_distances(p1, p2[:, perm].contiguous())
    assert torch.equal(a1, b1) and torch.equal(perm[bi1], ai1)
    assert torch.equal(a2[:, perm], b2) and torch.equal(ai2[:, perm], bi2)
    # symmetry: swapping arguments swaps the two sides bit for bit
    c1, ci1, c2, ci2 = sided_distances(p2, p1)
    assert torch.equal(c1, a2) and torch.equal(c2, a1) and torch.equal(ci1, ai2)
    assert torch.equal(chamfer_distance(p1, p2), chamfer_distance(p2, p1))
    # every reported distance is the distance to the reported index
    gathered = torch.gather(p2, 1, ai1.unsqueeze(-1).expand(-1, -1, 3))
    assert torch.allclose(((p1 - gathered) ** 2).sum(-1), a1, rtol=1e-5, atol=1e-7)


def test_rejects_bad_inputs(gpu):
    from fpsg_amd.metrics import chamfer_distance
    from fpsg_amd._hip import FpsgHipError
    a = torch.rand(2, 8, 3)
    with pytest.raises(FpsgHipError):
        chamfer_distance(a, a)  # CPU tensors: no fallback
    with pytest.raises(TypeError):
        chamfer_distance(a.double().to(gpu), a.double().to(gpu))
    with pytest.raises(ValueError):
        chamfer_distance(a.to(gpu).transpose(1, 2), a.to(gpu).transpose(1, 2))
    with pytest.raises(ValueError):
        chamfer_distance(a.to(gpu)[:, :, :2].contiguous(), a.to(gpu))
    with pytest.raises(ValueError):
        chamfer_distance(a.to(gpu)[:, :0], a.to(gpu))


def test_graph_replay_equals_eager_step(gpu):
    """TrainStep(graph=True) replays the captured episode: same gradients as the eager step
    (decoder grid injected so that both paths see the same random patch samples).  Library
    kernels are not bit-reproducible run to run and the tiny-batch BatchNorms amplify that, so
    the yardstick is a second EAGER run of the same step."""
    from fpsg_amd.engine import TrainStep, build_model, build_optimizer, default_options
    from fpsg_amd.episodes import synthetic_episode
    import copy
    torch.manual_seed(0)
    opt = default_options(device="cuda", intra_recon=True, lr=0.0)
    base = build_model(opt).to(gpu).train()
    eps = [synthetic_episode(4, 2, n_pts=2048, img_size=96, seed=s, device=gpu) for s in (1, 2)]
    grads = {}
    for mode in ("eager", "eager2", "graph"):
        m = copy.deepcopy(base)
        optimizer, _ = build_optimizer(m, opt)
        step = TrainStep(m, optimizer, graph=(mode == "graph"))
        fixed = {b: m.pc_decoder.sample_grids(b, gpu, torch.Generator(device=gpu).manual_seed(5 + b)) for b in (4, 2)}
        orig = m.pc_decoder.forward
        m.pc_decoder.forward = lambda h, grid=None, generator=None, pack=None, orig=orig, fixed=fixed: orig(h, grid=fixed[h.size(0)], pack=pack)
        pair = m.pc_decoder.forward_pair
        m.pc_decoder.forward_pair = lambda a, b, generator=None, pack=None, pair=pair, fixed=fixed: pair(
            a, b, pack=pack, grids=(fixed[a.size(0)], fixed[b.size(0)]))
        for _ in range(4):                       # graph mode: 2 eager uses, capture, replay
            out = step([eps[0], eps[1]])
        if mode == "graph":
            assert len(step._graphs) == 2        # same shape, first (copy) / later (add) episode of a step
        grads[mode] = (step.buckets.flat.clone(), float(out[-1]["ttl_loss"].sum()))
    cos = torch.nn.functional.cosine_similarity
    g_e, l_e = grads["eager"]
    g_2, l_2 = grads["eager2"]
    g_g, l_g = grads["graph"]
    noise = 1 - float(cos(g_e, g_2, dim=0))
    diff = 1 - float(cos(g_e, g_g, dim=0))
    assert abs(l_e - l_g) <= max(5e-3 * abs(l_e), 3 * abs(l_e - l_2)), (l_e, l_2, l_g)
    assert diff <= max(3 * noise, 2e-3), (noise, diff)
    assert abs(float(g_g.norm() / g_e.norm()) - 1) < max(3 * abs(float(g_2.norm() / g_e.norm()) - 1), 1e-2)
