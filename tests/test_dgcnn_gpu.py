"""K3 (kNN) and K4a (edge features) through the C ABI vs the oracle (bit-exact indices /
features) and vs goldens produced by the reference's own dgcnn/model.py functions."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(GOLDEN, "dgcnn_goldens.npz"))


KNN_SHAPES = [(2, 3, 256, 20), (2, 3, 2048, 20), (1, 64, 1024, 20), (2, 5, 100, 7), (1, 128, 512, 20),
              (3, 3, 33, 33), (1, 6, 17, 1), (2, 64, 2048, 20), (1, 3, 64, 64), (1, 130, 300, 16),
              (1, 3, 2049, 20), (2, 3, 5000, 20), (1, 64, 4100, 32), (1, 20, 2500, 5)]   # > one LDS tile


@pytest.mark.parametrize("B,C,N,k", KNN_SHAPES)
def test_knn_bit_exact_vs_oracle(gpu, oracle, B, C, N, k):
    from fpsg_amd.dgcnn import knn
    rng = np.random.default_rng(B * 7 + C * 13 + N)
    x = rng.standard_normal((B, C, N)).astype(np.float32)
    got = knn(torch.from_numpy(x).to(gpu), k)
    assert got.dtype == torch.int64 and got.shape == (B, N, k)
    exp = oracle.knn(x, k)
    assert np.array_equal(got.cpu().numpy(), exp)


def test_knn_ties_and_duplicates(gpu, oracle):
    from fpsg_amd.dgcnn import knn
    rng = np.random.default_rng(1)
    x = rng.integers(-2, 3, size=(2, 3, 200)).astype(np.float32)     # lattice: many exact ties
    x[:, :, 100:] = x[:, :, :100]                                     # exact duplicates
    got = knn(torch.from_numpy(x).to(gpu), 20).cpu().numpy()
    assert np.array_equal(got, oracle.knn(x, 20))


# the streaming kernel's range (C <= 128, k <= 24), sizes around its tiles: 16-candidate tiles, 32-/64-/512-candidate
# stages, 128-query workgroups, more than 8 clouds (the cloud -> XCD mapping), k odd / 1 / 24
STREAM_SHAPES = [(1, 3, 16, 5), (1, 3, 31, 20), (2, 3, 129, 20), (9, 3, 300, 20), (1, 3, 513, 7), (3, 3, 1024, 24),
                 (2, 4, 700, 24), (1, 1, 257, 3), (2, 2, 2048, 20), (17, 3, 255, 20), (1, 3, 24, 24),     # C = 1..4, k = N
                 (1, 64, 64, 20), (2, 64, 65, 1), (1, 64, 1000, 20), (1, 48, 257, 9), (1, 128, 33, 20), (2, 128, 200, 13),
                 (1, 100, 777, 20), (1, 17, 2048, 20), (10, 64, 256, 20)]


@pytest.mark.parametrize("B,C,N,k", STREAM_SHAPES)
def test_knn_kernels_and_layouts_agree_with_oracle(gpu, oracle, B, C, N, k):
    """The streaming kernel (default in this range), its slow exact path, the score-tile kernel and the point-major
    entry all give the oracle's lists."""
    from fpsg_amd.dgcnn import KNN_FORCE_SLOW, KNN_FORCE_TILE, knn_int32
    rng = np.random.default_rng(B * 11 + C * 5 + N + k)
    x = rng.standard_normal((B, C, N)).astype(np.float32)
    exp = oracle.knn(x, k)
    xt = torch.from_numpy(x).to(gpu)
    assert np.array_equal(knn_int32(xt, k).cpu().numpy(), exp)
    assert np.array_equal(knn_int32(xt, k, flags=KNN_FORCE_SLOW).cpu().numpy(), exp)
    assert np.array_equal(knn_int32(xt, k, flags=KNN_FORCE_TILE).cpu().numpy(), exp)
    assert np.array_equal(knn_int32(xt.transpose(1, 2).contiguous(), k, point_major=True).cpu().numpy(), exp)


def test_knn_many_equal_scores(gpu, oracle):
    """Hundreds of coincident points: the threshold filter cannot bring a row's buffer under its watermark, so the row
    is compacted to its exact k best -- same lists as the oracle (lowest index first among equal scores)."""
    from fpsg_amd.dgcnn import knn_int32
    rng = np.random.default_rng(5)
    x = rng.standard_normal((2, 64, 640)).astype(np.float32)
    x[:, :, 100:400] = x[:, :, 100:101]                 # 300 copies of one point
    x[1, :, :] = 0.0                                    # a cloud collapsed to one point: every score equal
    got = knn_int32(torch.from_numpy(x).to(gpu), 20).cpu().numpy()
    assert np.array_equal(got, oracle.knn(x, 20))


@pytest.mark.parametrize("order", ["ascending", "descending", "interleaved"])
def test_knn_adversarial_candidate_orders(gpu, oracle, order):
    """Points sorted along a line: for most queries every new candidate beats all earlier ones (or none does), the
    worst case for a streaming threshold -- the buffers hit their watermark every few tiles and the events do the work.
    Same lists as the oracle."""
    from fpsg_amd.dgcnn import knn_int32
    rng = np.random.default_rng(17)
    N = 1500
    t = np.sort(rng.random(N).astype(np.float32))
    if order == "descending":
        t = t[::-1].copy()
    elif order == "interleaved":
        t = np.concatenate([t[::2], t[1::2][::-1]])
    x = np.stack([t, 0.25 * t, -0.5 * t])[None].astype(np.float32)            # [1,3,N]
    x = np.concatenate([x, x + rng.standard_normal(x.shape).astype(np.float32) * 1e-3])
    got = knn_int32(torch.from_numpy(x).to(gpu), 20).cpu().numpy()
    assert np.array_equal(got, oracle.knn(x, 20))
    x64 = np.concatenate([x, np.zeros((2, 61, N), np.float32)], axis=1)         # the same through the C = 64 kernel
    assert np.array_equal(knn_int32(torch.from_numpy(x64).to(gpu), 20).cpu().numpy(), oracle.knn(x64, 20))


def test_knn_three_channels_many_equal_scores_and_signed_zeros(gpu, oracle):
    """C = 3 (DGCNN's first layer) where the tie rule matters: hundreds of coincident points (every later candidate ties
    with the k-th kept one and must lose to the lower indices), a cloud collapsed to the origin (scores +0 / -0),
    coordinates with both signs of zero."""
    from fpsg_amd.dgcnn import knn_int32
    rng = np.random.default_rng(6)
    x = rng.standard_normal((3, 3, 900)).astype(np.float32)
    x[0, :, 100:500] = x[0, :, 100:101]                 # 400 copies of one point
    x[1] = 0.0
    x[1, :, ::2] = -0.0                                 # signed zeros: every score is +-0
    x[2, 1] = 0.0                                       # a flat cloud
    got = knn_int32(torch.from_numpy(x).to(gpu), 20).cpu().numpy()
    assert np.array_equal(got, oracle.knn(x, 20))
    got = knn_int32(torch.from_numpy(x[:, :, :300].copy()).to(gpu), 24).cpu().numpy()
    assert np.array_equal(got, oracle.knn(x[:, :, :300].copy(), 24))


def test_knn_point_major_is_refused_outside_the_streaming_range(gpu):
    from fpsg_amd.dgcnn import knn_int32
    from fpsg_amd._hip import FpsgHipError
    with pytest.raises(FpsgHipError):
        knn_int32(torch.rand(1, 64, 200, device=gpu), 20, point_major=True)      # C = 200
    with pytest.raises(FpsgHipError):
        knn_int32(torch.rand(1, 64, 16, device=gpu), 32, point_major=True)       # k = 32


@pytest.mark.parametrize("tag", ["c3_n256", "c3_n2048", "c64_n256", "c64_n2048"])
def test_knn_matches_reference_function(gpu, gold, tag):
    """Against reference knn() outputs: identical neighbour sets except fp32 near-ties
    (torch.matmul rounds the inner products in a different order)."""
    from fpsg_amd.dgcnn import knn
    x = gold[f"knn_x_{tag}"]
    ref = gold[f"knn_idx_{tag}"]
    got = knn(torch.from_numpy(x).to(gpu), 20).cpu().numpy()
    assert (got[:, :, 0] == np.arange(x.shape[2])[None]).mean() > 0.999      # self first
    same_pos = (got == ref).mean()
    same_set = np.mean([len(set(a) & set(b)) / 20.0 for a, b in zip(got.reshape(-1, 20), ref.reshape(-1, 20))])
    assert same_set > 0.999 and same_pos > 0.99, (same_set, same_pos)


@pytest.mark.parametrize("B,C,N,k", [(2, 3, 256, 20), (1, 64, 512, 20), (2, 5, 96, 20), (1, 128, 300, 9)])
def test_edge_feature_fwd_bwd_vs_oracle(gpu, oracle, B, C, N, k):
    from fpsg_amd.dgcnn import get_graph_feature
    rng = np.random.default_rng(C + N)
    x = rng.standard_normal((B, C, N)).astype(np.float32)
    idx = rng.integers(0, N, size=(B, N, k)).astype(np.int32)
    xt = torch.from_numpy(x).to(gpu).requires_grad_()
    out = get_graph_feature(xt, k=k, idx=torch.from_numpy(idx).to(gpu).long())
    exp = oracle.edge_feature(x, idx)
    assert out.shape == (B, 2 * C, N, k) and np.array_equal(out.detach().cpu().numpy(), exp)
    g = rng.standard_normal(exp.shape).astype(np.float32)
    out.backward(torch.from_numpy(g).to(gpu))
    gexp = oracle.edge_feature_bwd(g, idx)
    np.testing.assert_allclose(xt.grad.cpu().numpy(), gexp, rtol=1e-4, atol=1e-4)   # fp32 atomics


def test_get_graph_feature_matches_reference_function(gpu, gold):
    from fpsg_amd.dgcnn import get_graph_feature
    out = get_graph_feature(torch.from_numpy(gold["ggf_x"]).to(gpu), k=20).cpu().numpy()
    ref = gold["ggf_out"]
    assert out.shape == ref.shape
    assert (out == ref).mean() > 0.995       # only fp32 near-tie neighbour swaps may differ
    assert np.array_equal(out[:, 5:], ref[:, 5:])   # centre half is independent of the graph


@pytest.mark.parametrize("mode", ["eval", "train"])
def test_dgcnn_encoder_matches_reference_module(gpu, gold, mode):
    from fpsg_amd.dgcnn import DGCNNfeat
    net = DGCNNfeat()
    net.load_state_dict(torch.load(os.path.join(GOLDEN, "dgcnn_state.pt"), weights_only=True), strict=True)
    net = net.to(gpu).train(mode == "train")
    assert sum(p.numel() for p in net.parameters()) == 354688
    with torch.no_grad():
        feat = net(torch.from_numpy(gold["dgcnn_x"]).to(gpu)).cpu().numpy()
    ref = gold[f"dgcnn_feat_{mode}"]
    assert feat.shape == ref.shape == (3, 1024)
    close = np.isclose(feat, ref, rtol=2e-3, atol=2e-4)
    assert close.mean() > 0.995, close.mean()          # a near-tie neighbour swap moves few features
    if mode == "train":
        np.testing.assert_allclose(net.conv4[1].running_mean.cpu().numpy(), gold["dgcnn_conv4_bn_mean_train"],
                                   rtol=1e-3, atol=1e-4)
        np.testing.assert_allclose(net.conv4[1].running_var.cpu().numpy(), gold["dgcnn_conv4_bn_var_train"],
                                   rtol=1e-3, atol=1e-4)


def test_knn_rejects_bad_inputs(gpu):
    from fpsg_amd.dgcnn import knn
    from fpsg_amd._hip import FpsgHipError
    with pytest.raises(FpsgHipError):
        knn(torch.rand(1, 3, 64), 20)
    with pytest.raises(ValueError):
        knn(torch.rand(1, 3, 10, device=gpu), 20)          # k > N
    with pytest.raises(FpsgHipError):
        knn(torch.rand(1, 500, 64, device=gpu), 20)        # C beyond the LDS query-tile limit
    assert knn(torch.rand(1, 3, 4096, device=gpu), 20).shape == (1, 4096, 20)   # any N


@pytest.mark.parametrize("mode", ["train", "eval"])
def test_fused_edgeconv_equals_reference_chain(gpu, mode):
    """K4b (gather + BN statistics + max fused, nothing materialised) against the literal
    reference chain get_graph_feature -> Conv2d -> BatchNorm2d -> LeakyReLU -> max (itself
    checked against the reference goldens above): forward, running statistics, and the
    gradients of the input and of every parameter."""
    import copy
    from fpsg_amd.dgcnn import DGCNNfeat
    torch.manual_seed(3)
    a = DGCNNfeat(fused=True)
    a.load_state_dict(torch.load(os.path.join(GOLDEN, "dgcnn_state.pt"), weights_only=True))  # gammas of both signs
    a = a.to(gpu)
    b = copy.deepcopy(a)
    b.fused = False
    a.train(mode == "train"); b.train(mode == "train")
    g = torch.Generator().manual_seed(9)
    pts = torch.randn(4, 3, 320, generator=g)
    pts = (pts / pts.norm(dim=1, keepdim=True).amax(dim=2, keepdim=True)).to(gpu)
    xa = pts.clone().requires_grad_()
    xb = pts.clone().requires_grad_()
    fa, fb = a(xa), b(xb)
    assert fa.shape == fb.shape == (4, 1024)
    scale = fb.abs().max()
    assert (fa - fb).abs().max() <= 2e-4 * scale, (fa - fb).abs().max() / scale
    w = torch.randn(fa.shape, generator=g).to(gpu)
    (fa * w).sum().backward()
    (fb * w).sum().backward()
    gs = xb.grad.abs().max()
    assert (xa.grad - xb.grad).abs().max() <= 5e-3 * gs, (xa.grad - xb.grad).abs().max() / gs
    for (n1, p1), (_, p2) in zip(a.named_parameters(), b.named_parameters()):
        s = p2.grad.abs().max() + 1e-12
        assert (p1.grad - p2.grad).abs().max() <= 5e-3 * s, (n1, float((p1.grad - p2.grad).abs().max() / s))
    for (n1, b1), (_, b2) in zip(a.named_buffers(), b.named_buffers()):
        assert torch.allclose(b1.float(), b2.float(), rtol=1e-4, atol=1e-5), n1


def test_fused_edgeconv_is_deterministic(gpu):
    from fpsg_amd.dgcnn import DGCNNfeat
    torch.manual_seed(0)
    net = DGCNNfeat().to(gpu).train()
    x = torch.randn(3, 3, 512, device=gpu)
    outs = []
    for _ in range(2):
        net.zero_grad()
        xi = x.clone().requires_grad_()
        net(xi).square().sum().backward()
        outs.append((xi.grad.clone(), net.conv3[0].weight.grad.clone()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])


def test_max_mean_over_points_matches_autograd(gpu):
    """dgcnn._MaxMeanOverPoints against the plain ``cat(max, mean)`` graph it replaces: value bit for bit, gradient to
    fp32 round-off (one add of the two shares in a different order); ties in the max go to the same index."""
    import torch
    from fpsg_amd.dgcnn import _MaxMeanOverPoints
    torch.manual_seed(2)
    h = torch.randn(5, 48, 300, device=gpu)
    h[0, 0, 7] = h[0, 0, 250] = 10.0                      # a tie: torch.max picks one index, both paths use it
    w = torch.randn(5, 96, device=gpu)
    a = h.clone().requires_grad_()
    out = _MaxMeanOverPoints.apply(a)
    (out * w).sum().backward()
    b = h.clone().requires_grad_()
    ref = torch.cat((b.max(dim=2)[0], b.mean(dim=2)), dim=1)
    (ref * w).sum().backward()
    assert torch.equal(out, ref)
    assert float((a.grad - b.grad).abs().max()) <= 1e-6 * float(b.grad.abs().max())


@pytest.mark.parametrize("B,N,k,kind", [(3, 2048, 20, "random"), (2, 2048, 20, "hubs"), (2, 512, 20, "same"),
                                        (2, 1, 1, "random"), (3, 77, 9, "random"), (2, 1025, 1, "random"),
                                        (1, 3276, 20, "random"), (2, 300, 64, "hubs"), (2, 2047, 32, "random")])
def test_reverse_graph_is_the_stable_grouping(gpu, oracle, B, N, k, kind):
    """fpsg_edgeconv_reverse_graph against the oracle (the order of its sequential scatter loop) and against the torch-sort form of the
    mirror: random lists, hub destinations (in-degree in the hundreds), every source choosing the same k points
    (in-degree N), odd N, single points, the 65535-edge limit -- bit for bit, twice (no dependence on timing)."""
    from fpsg_amd import _hip
    from fpsg_amd.dgcnn import _reverse_graph, _reverse_graph_sorted
    rng = np.random.default_rng(N * 31 + k)
    if kind == "random":
        idx = rng.integers(0, N, size=(B, N, k))
    elif kind == "hubs":
        idx = np.minimum(rng.geometric(0.02, size=(B, N, k)) - 1, N - 1)         # low indices are chosen by hundreds
    else:
        idx = np.broadcast_to(np.arange(k)[None, None, :] % N, (B, N, k)).copy()
    idx = idx.astype(np.int32)
    assert _hip.load().fpsg_edgeconv_reverse_graph_fits(N, k) == 1
    t = torch.from_numpy(idx).to(gpu)
    rev, off = _reverse_graph(t)
    rev2, off2 = _reverse_graph(t)
    srev, soff = _reverse_graph_sorted(t)
    assert torch.equal(rev, rev2) and torch.equal(off, off2)
    assert torch.equal(rev, srev) and torch.equal(off, soff)
    for b in range(B):
        erev, eoff = oracle.in_edge_lists(idx[b])
        assert np.array_equal(off[b].cpu().numpy(), eoff)
        assert np.array_equal(rev[b].cpu().numpy(), erev)


def test_reverse_graph_limits_and_bad_entries(gpu, oracle):
    from fpsg_amd import _hip
    from fpsg_amd.dgcnn import _reverse_graph, _reverse_graph_sorted
    lib = _hip.load()
    assert lib.fpsg_edgeconv_reverse_graph_fits(2048, 20) == 1
    assert lib.fpsg_edgeconv_reverse_graph_fits(4096, 20) == 0          # 81,920 edges: beyond the 16-bit ranks
    assert lib.fpsg_edgeconv_reverse_graph_fits(0, 20) == 0
    idx = torch.randint(0, 4096, (1, 4096, 20), dtype=torch.int32, device=gpu)
    rev = torch.empty(1, 4096 * 20, dtype=torch.int32, device=gpu)
    off = torch.empty(1, 4097, dtype=torch.int32, device=gpu)
    rc = lib.fpsg_edgeconv_reverse_graph(idx.data_ptr(), 1, 4096, 20, rev.data_ptr(), off.data_ptr(), None)
    assert rc == -4 and b"65535" in lib.fpsg_last_error()
    a, b = _reverse_graph(idx)                                          # the mirror sorts instead
    c, d = _reverse_graph_sorted(idx)
    assert torch.equal(a, c) and torch.equal(b, d)
    # entries outside [0, N) are skipped: the valid edges are grouped as before, off[N] counts them
    bad = torch.randint(0, 100, (2, 100, 8), dtype=torch.int32, device=gpu)
    bad[0, 3, 2] = -1
    bad[1, 50, 0] = 100
    rev = torch.full((2, 800), -7, dtype=torch.int32, device=gpu)
    off = torch.empty(2, 101, dtype=torch.int32, device=gpu)
    assert lib.fpsg_edgeconv_reverse_graph(bad.data_ptr(), 2, 100, 8, rev.data_ptr(), off.data_ptr(), None) == 0
    for bi in range(2):
        erev, eoff = oracle.in_edge_lists(bad[bi].cpu().numpy())
        assert eoff[-1] == 799
        assert np.array_equal(off[bi].cpu().numpy(), eoff)
        assert np.array_equal(rev[bi, :799].cpu().numpy(), erev) and int(rev[bi, 799]) == -7


@pytest.mark.parametrize("Co,stats", [(64, True), (128, True), (256, True), (64, False)])
def test_edgeconv_backward_forms_agree(gpu, monkeypatch, Co, stats):
    """fpsg_edgeconv_bwd: the form with several in-edges per load instruction (default) against the one-edge form of
    round 2 (FPSG_EDGECONV_BWD=one_edge) on a graph with hubs and isolated points -- equal up to the order of the fp32
    sums (a lane group sums every G-th in-edge, then the groups are added), and bit-identical from run to run."""
    from fpsg_amd import _hip
    from fpsg_amd.dgcnn import _reverse_graph
    lib = _hip.load()
    torch.manual_seed(Co)
    B, N, k = 3, 300, 20
    idx = torch.randint(0, N, (B, N, k), dtype=torch.int32, device=gpu)
    idx[:, :, :5] = torch.randint(0, 7, (B, N, 5), dtype=torch.int32, device=gpu)      # hubs: in-degree of hundreds
    idx[0, :, :] = 5                                                                       # one point takes every edge
    rev, off = _reverse_graph(idx)
    dzs = torch.randn(B, N, Co, device=gpu)
    jsel = torch.randint(0, k, (B, N, Co), dtype=torch.uint8, device=gpu)
    PQ = torch.randn(B, N, 2 * Co, device=gpu)
    s1 = torch.randn(B, N, Co, device=gpu) if stats else None
    coef = torch.randn(3, Co, device=gpu) * 0.1
    st = torch.cuda.current_stream().cuda_stream

    def run():
        out = torch.full_like(PQ, float("nan"))
        _hip.check(lib.fpsg_edgeconv_bwd(dzs.data_ptr(), jsel.data_ptr(), PQ.data_ptr(), s1.data_ptr() if stats else None,
                                         rev.data_ptr(), off.data_ptr(), coef.data_ptr(), B, N, k, Co, out.data_ptr(), st),
                   "fpsg_edgeconv_bwd")
        return out

    a, a2 = run(), run()
    monkeypatch.setenv("FPSG_EDGECONV_BWD", "one_edge")
    b = run()
    assert torch.equal(a, a2) and bool(torch.isfinite(a).all())
    scale = float(b.abs().max())
    assert float((a - b).abs().max()) <= 2e-5 * scale, float((a - b).abs().max()) / scale


@pytest.mark.parametrize("Co", [64, 128])
def test_edgeconv_forward_forms_agree(gpu, monkeypatch, Co):
    """fpsg_edgeconv_fwd: several neighbours per load instruction (default at Co = 64 / 128) against the one-neighbour
    form: the selected values and slots are identical (ties: lowest slot, duplicates in the lists included), the sums
    equal up to the order of the fp32 additions."""
    from fpsg_amd import _hip
    lib = _hip.load()
    torch.manual_seed(Co + 1)
    B, N, k = 3, 333, 20
    idx = torch.randint(0, N, (B, N, k), dtype=torch.int32, device=gpu)
    idx[:, :, 7] = idx[:, :, 2]                        # repeated neighbours: equal values in different slots
    idx[1, :, :] = idx[1, :, :1]                       # every slot the same point: all equal, slot 0 must win
    PQ = torch.randn(B, N, 2 * Co, device=gpu)
    sgn = torch.where(torch.randn(Co, device=gpu) < 0, -1.0, 1.0)
    st = torch.cuda.current_stream().cuda_stream

    def run():
        ysel = torch.empty(B, N, Co, device=gpu)
        jsel = torch.empty(B, N, Co, dtype=torch.uint8, device=gpu)
        s1 = torch.empty(B, N, Co, device=gpu)
        part = torch.empty(lib.fpsg_edgeconv_blocks(B, N, Co), 2, Co, device=gpu)
        _hip.check(lib.fpsg_edgeconv_fwd(PQ.data_ptr(), idx.data_ptr(), sgn.data_ptr(), B, N, k, Co, ysel.data_ptr(),
                                         jsel.data_ptr(), s1.data_ptr(), part.data_ptr(), st), "fpsg_edgeconv_fwd")
        return ysel, jsel, s1, part

    a = run()
    a2 = run()
    monkeypatch.setenv("FPSG_EDGECONV_BWD", "one_edge")
    b = run()
    assert all(torch.equal(x, y) for x, y in zip(a, a2))
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    assert bool((a[1][1] == 0).all())
    assert float((a[2] - b[2]).abs().max()) <= 1e-5 * float(b[2].abs().max())
    assert float((a[3] - b[3]).abs().max()) <= 1e-5 * float(b[3].abs().max())


@pytest.mark.parametrize("blocks,Co", [(8192, 64), (8192, 128), (16384, 256), (300, 64), (257, 256), (64, 128)])
def test_edgeconv_statistics_finalize_in_two_stages(gpu, blocks, Co):
    """fpsg_edgeconv_stats_finalize_ws (slice sums over coalesced row pieces, then one wave per channel) against the
    one-launch form and float64 sums of the same partial rows: scale / shift / mean / rstd and the running statistics."""
    from fpsg_amd import _hip
    lib = _hip.load()
    g = torch.Generator().manual_seed(blocks + Co)
    count = float(blocks * 256 * 20)
    n_b = 256 * 20                                                             # edge activations per block
    s0 = n_b * (0.3 + 0.05 * torch.randn(blocks, Co, generator=g))             # per-block sums of y (mean ~0.3, std ~1)
    s1 = n_b * (1.09 + 0.05 * torch.randn(blocks, Co, generator=g))            # ... and of y^2
    part = torch.stack([s0, s1], dim=1).contiguous().to(gpu)
    gamma = (torch.randn(Co, generator=g) * 0.5 + 1).to(gpu)
    beta = (torch.randn(Co, generator=g) * 0.1).to(gpu)
    st = torch.cuda.current_stream().cuda_stream
    res = []
    for two_stage in (False, True):
        rm, rv = torch.zeros(Co, device=gpu), torch.ones(Co, device=gpu)
        chan = torch.empty(4, Co, device=gpu)
        if two_stage:
            ws = torch.empty(lib.fpsg_edgeconv_stats_ws_floats(blocks, Co), device=gpu)
            rc = lib.fpsg_edgeconv_stats_finalize_ws(part.data_ptr(), blocks, gamma.data_ptr(), beta.data_ptr(), rm.data_ptr(),
                                                     rv.data_ptr(), 0.1, 1e-5, count, Co, 1, chan.data_ptr(), ws.data_ptr(), st)
        else:
            rc = lib.fpsg_edgeconv_stats_finalize(part.data_ptr(), blocks, gamma.data_ptr(), beta.data_ptr(), rm.data_ptr(),
                                                  rv.data_ptr(), 0.1, 1e-5, count, Co, 1, chan.data_ptr(), st)
        assert rc == 0, lib.fpsg_last_error()
        res.append((chan.cpu(), rm.cpu(), rv.cpu()))
    for a, b in zip(res[0], res[1]):
        assert torch.allclose(a, b, rtol=2e-6, atol=1e-7)
    p64 = part.double().cpu()
    mean = p64[:, 0].sum(0) / count
    var = (p64[:, 1].sum(0) / count - mean * mean).clamp_min(0)
    rstd = 1.0 / torch.sqrt(var + 1e-5)
    chan = res[1][0].double()
    assert torch.allclose(chan[2], mean, rtol=1e-6) and torch.allclose(chan[3], rstd, rtol=1e-5)
    assert torch.allclose(chan[0], gamma.double().cpu() * rstd, rtol=1e-5)
    assert torch.allclose(res[1][1].double(), 0.1 * mean, rtol=1e-6)
