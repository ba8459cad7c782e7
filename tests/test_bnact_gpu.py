"""K5 (BatchNorm fused with its activation) through the C ABI against PyTorch's own
BatchNorm + activation modules: outputs, input/parameter gradients, running statistics, in
training and eval mode.  fp32 tolerance: the two implementations sum in different orders."""
import numpy as np
import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu

CASES = [  # (shape, act)
    ((4, 64, 56, 56), "relu"), ((3, 512, 14, 14), "relu"), ((6, 1024, 2048), None), ((6, 128, 2048), "relu"),
    ((1, 16 * 67, 640), "relu"), ((2, 7, 4096 + 64), ("leaky", 0.2)), ((5, 3, 8, 8), "relu"), ((2, 64, 224, 224), "relu"),
    # more than 16384 values per channel: the multi-launch (sliced) path
    ((8, 64, 112, 112), "relu"), ((24, 32, 2048), None), ((5, 9, 4096 + 4100), ("leaky", 0.1)), ((37, 16, 56, 56), "relu"),
]


def _ref(bn, x, act):
    y = bn(x)
    if act == "relu":
        return torch.relu(y)
    if isinstance(act, tuple):
        return torch.nn.functional.leaky_relu(y, act[1])
    return y


@pytest.mark.parametrize("shape,act", CASES)
@pytest.mark.parametrize("mode", ["train", "eval"])
def test_matches_torch_modules(gpu, shape, act, mode):
    from fpsg_amd.fused_bn import bn_act, _eligible
    import copy
    torch.manual_seed(sum(shape))
    C = shape[1]
    bn = (nn.BatchNorm2d if len(shape) == 4 else nn.BatchNorm1d)(C).to(gpu)
    with torch.no_grad():
        bn.weight.copy_(torch.randn(C) * 0.5 + 1)
        bn.bias.copy_(torch.randn(C) * 0.3)
        bn.running_mean.copy_(torch.randn(C) * 0.2)
        bn.running_var.copy_(torch.rand(C) + 0.5)
    ref = copy.deepcopy(bn)
    bn.train(mode == "train"); ref.train(mode == "train")
    x1 = (torch.randn(*shape, device=gpu) * 1.7 + 0.4).requires_grad_()
    x2 = x1.detach().clone().requires_grad_()
    assert _eligible(x1)
    y1 = bn_act(bn, x1, act)
    y2 = _ref(ref, x2, act)
    assert torch.allclose(y1, y2, rtol=1e-4, atol=1e-5), float((y1 - y2).abs().max())
    g = torch.randn_like(y1)
    y1.backward(g); y2.backward(g)
    s = float(x2.grad.abs().max())
    # a pre-activation within fp32 round-off of 0 may fall on either side of the ReLU kink in the
    # two implementations (z = x*scale+shift here, (x-mean)*rstd*gamma+beta there): such isolated
    # elements get the other one-sided derivative; everything else must agree tightly
    bad = (x1.grad - x2.grad).abs() > 2e-4 * s + 1e-6
    assert int(bad.sum()) <= max(2, int(2e-6 * bad.numel())), int(bad.sum())
    if bad.any():
        assert float(y2.detach()[bad].abs().max()) < 1e-5
    # channels that own such a kink element get its whole one-sided contribution in dgamma/dbeta
    kink_channels = bad.transpose(0, 1).reshape(C, -1).any(dim=1)
    for a, b in ((bn.weight.grad, ref.weight.grad), (bn.bias.grad, ref.bias.grad)):
        scale = float(b.abs().max())
        diff = (a - b).abs()
        assert float(diff[~kink_channels].max()) <= 3e-4 * scale + 1e-5
        if kink_channels.any():
            assert float(diff[kink_channels].max()) <= 30.0
    assert torch.allclose(bn.running_mean, ref.running_mean, rtol=1e-5, atol=1e-6)
    assert torch.allclose(bn.running_var, ref.running_var, rtol=1e-4, atol=1e-6)
    assert int(bn.num_batches_tracked) == int(ref.num_batches_tracked)


def test_dispatch_rules(gpu):
    from fpsg_amd.fused_bn import bn_act, _eligible
    bn = nn.BatchNorm1d(8).to(gpu)
    assert not _eligible(torch.randn(4, 8, device=gpu))            # nn.Linear output: plain modules
    assert not _eligible(torch.randn(4, 8, 30, device=gpu))        # short / unaligned rows
    assert not _eligible(torch.randn(4, 8, 256))                   # CPU tensor of the CPU port
    y = bn_act(bn, torch.randn(4, 8, 30, device=gpu), "relu")
    assert y.shape == (4, 8, 30) and float(y.detach().min()) >= 0
    cpu_bn = nn.BatchNorm1d(8)
    assert bn_act(cpu_bn, torch.randn(4, 8, 256), "relu").shape == (4, 8, 256)


def test_deterministic(gpu):
    from fpsg_amd.fused_bn import bn_act
    bn = nn.BatchNorm2d(64).to(gpu).train()
    x = torch.randn(8, 64, 56, 56, device=gpu)
    outs = []
    for _ in range(2):
        xi = x.clone().requires_grad_()
        y = bn_act(bn, xi, "relu")
        y.square().sum().backward()
        outs.append((y.detach().clone(), xi.grad.clone()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])


@pytest.mark.parametrize("shape,act", [((4, 64, 56, 56), "relu"), ((37, 16, 56, 56), "relu"), ((6, 128, 2048), None),
                                       ((24, 32, 2048), "relu"), ((2, 7, 4096 + 64), ("leaky", 0.2))])
@pytest.mark.parametrize("mode", ["train", "eval"])
def test_pre_bias_is_the_broadcast_add(gpu, shape, act, mode):
    """pre_bias: act(BN(x + b[c])) with the add done in registers -- the forward is BIT-equal to
    feeding the materialised x + b, dx likewise, and dpre_bias is the sum of dx (fp64 yardstick)."""
    from fpsg_amd.fused_bn import bn_act
    import copy
    torch.manual_seed(sum(shape) + 1)
    C = shape[1]
    bn = (nn.BatchNorm2d if len(shape) == 4 else nn.BatchNorm1d)(C).to(gpu)
    with torch.no_grad():
        bn.weight.copy_(torch.randn(C) * 0.5 + 1)
        bn.bias.copy_(torch.randn(C) * 0.3)
    bn2 = copy.deepcopy(bn)
    bn.train(mode == "train"); bn2.train(mode == "train")
    x = torch.randn(*shape, device=gpu) * 1.3
    b = (torch.randn(C, device=gpu) * 0.5).requires_grad_()
    x1 = x.clone().requires_grad_()
    x2 = (x + b.detach().view(1, C, *([1] * (len(shape) - 2)))).requires_grad_()
    y1 = bn_act(bn, x1, act, pre_bias=b)
    y2 = bn_act(bn2, x2, act)
    assert torch.equal(y1, y2)
    g = torch.randn_like(y1)
    y1.backward(g); y2.backward(g)
    assert torch.equal(x1.grad, x2.grad)
    assert torch.equal(bn.weight.grad, bn2.weight.grad) and torch.equal(bn.bias.grad, bn2.bias.grad)
    assert torch.equal(bn.running_mean, bn2.running_mean) and torch.equal(bn.running_var, bn2.running_var)
    dims = [0] + list(range(2, len(shape)))
    want = x2.grad.double().sum(dim=dims)
    mass = x2.grad.double().abs().sum(dim=dims)
    assert float(((b.grad.double() - want).abs() / (mass * 1e-6 + 1e-12)).max()) <= 1.0
    if mode == "eval":       # a real gradient there (in training mode BatchNorm cancels the bias: sum(dx) ~ 0)
        assert float(want.abs().max()) > 1e-3


@pytest.mark.parametrize("kind", ["conv2d", "conv1d"])
def test_conv_bn_act_matches_module_chain(gpu, kind):
    from fpsg_amd.fused_bn import conv_bn_act
    import copy
    torch.manual_seed(11)
    if kind == "conv2d":
        conv, bn, x = nn.Conv2d(16, 32, 3, padding=1), nn.BatchNorm2d(32), torch.randn(5, 16, 40, 40)
    else:
        conv, bn, x = nn.Conv1d(64, 128, 1), nn.BatchNorm1d(128), torch.randn(6, 64, 2048)
    conv, bn, x = conv.to(gpu), bn.to(gpu).train(), x.to(gpu)
    with torch.no_grad():
        conv.bias.copy_(torch.randn_like(conv.bias) * 0.3)
    conv2, bn2 = copy.deepcopy(conv), copy.deepcopy(bn)
    x1, x2 = x.clone().requires_grad_(), x.clone().requires_grad_()
    y1 = conv_bn_act(conv, bn, x1, "relu")
    y2 = torch.relu(bn2(conv2(x2)))
    assert torch.allclose(y1, y2, rtol=1e-4, atol=1e-5)
    g = torch.randn_like(y1)
    y1.backward(g); y2.backward(g)
    # an output within fp32 round-off of the ReLU kink may fall on either side in the two chains; each
    # such element perturbs one column of the input gradient and one row of the weight gradient
    flips = int(((y1 > 0) != (y2 > 0)).sum())
    assert flips <= 4
    fan = conv.weight[0].numel()
    bad = (x1.grad - x2.grad).abs() > 3e-4 * float(x2.grad.abs().max()) + 1e-5
    assert int(bad.sum()) <= flips * fan
    slack = flips * 8.0 * float(g.abs().max()) * float(x.abs().max())
    for a, r in ((conv.weight.grad, conv2.weight.grad), (bn.weight.grad, bn2.weight.grad),
                 (bn.bias.grad, bn2.bias.grad)):
        assert float((a - r).abs().max()) <= 3e-4 * float(r.abs().max()) + 1e-5 + slack
    # the convolution bias in front of a training-mode BatchNorm has a mathematically zero gradient;
    # both implementations return round-off of the size of eps * sum|dx|
    assert conv.bias.grad is not None and conv.bias.grad.shape == conv2.bias.grad.shape
    floor = 1e-5 * float(g.abs().sum() / g.shape[1])
    assert float(conv.bias.grad.abs().max()) <= floor and float(conv2.bias.grad.abs().max()) <= floor
    assert torch.allclose(bn.running_mean, bn2.running_mean, rtol=1e-5, atol=1e-6)
    assert torch.allclose(bn.running_var, bn2.running_var, rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("shape", [(8, 64, 56, 56), (37, 16, 28, 28), (3, 8, 112, 112), (20, 24, 30, 30), (2, 3, 224, 224),
                                   (48, 5, 64, 6)])
@pytest.mark.parametrize("mode", ["train", "eval"])
def test_pooled_variant_matches_unfused_chain(gpu, shape, mode):
    """conv_bn_act_pool == max_pool2d(K5(conv(x))) : pooled output BIT-equal (same arithmetic per
    element, max is exact); gradients agree to summation order (the pooled backward partitions its
    partial sums differently)."""
    from fpsg_amd.fused_bn import conv_bn_act, conv_bn_act_pool, _pool_eligible
    import copy
    torch.manual_seed(sum(shape))
    N, C, H, W = shape
    conv = nn.Conv2d(C, C, 3, padding=1).to(gpu)
    bn = nn.BatchNorm2d(C).to(gpu)
    pool = nn.MaxPool2d(2, 2)
    with torch.no_grad():
        conv.bias.copy_(torch.randn(C) * 0.3)
        bn.weight.copy_(torch.randn(C) * 0.5 + 1)
        bn.bias.copy_(torch.randn(C) * 0.3)
    conv2, bn2 = copy.deepcopy(conv), copy.deepcopy(bn)
    bn.train(mode == "train"); bn2.train(mode == "train")
    x = torch.randn(*shape, device=gpu)
    x1, x2 = x.clone().requires_grad_(), x.clone().requires_grad_()
    assert _pool_eligible(conv._conv_forward(x, conv.weight, None), pool)
    y1 = conv_bn_act_pool(conv, bn, pool, x1, "relu")
    y2 = pool(conv_bn_act(conv2, bn2, x2, "relu"))
    assert y1.shape == (N, C, H // 2, W // 2) and torch.equal(y1, y2)
    g = torch.randn_like(y1)
    y1.backward(g); y2.backward(g)
    for a, r in ((x1.grad, x2.grad), (conv.weight.grad, conv2.weight.grad), (bn.weight.grad, bn2.weight.grad),
                 (bn.bias.grad, bn2.bias.grad)):
        assert float((a - r).abs().max()) <= 2e-4 * float(r.abs().max()) + 1e-6
    assert torch.equal(bn.running_mean, bn2.running_mean) and torch.equal(bn.running_var, bn2.running_var)
    assert int(bn.num_batches_tracked) == int(bn2.num_batches_tracked)
    if mode == "eval":
        assert float((conv.bias.grad - conv2.bias.grad).abs().max()) <= 2e-4 * float(conv2.bias.grad.abs().max()) + 1e-6


def test_pooled_variant_ties_and_torch_pool(gpu):
    """Ties inside a window (all four activations clipped to 0 by the ReLU, or equal positives) go to
    the first element in (h, w) scan order, as torch's max_pool2d does: checked against the plain
    torch chain BatchNorm2d -> ReLU -> MaxPool2d on inputs built from a few repeated values."""
    from fpsg_amd.fused_bn import conv_bn_act_pool
    import copy
    torch.manual_seed(3)
    N, C, H, W = 6, 8, 64, 64
    conv = nn.Conv2d(C, C, 1).to(gpu)
    with torch.no_grad():                                   # identity 1x1 convolution, zero bias
        conv.weight.copy_(torch.eye(C).view(C, C, 1, 1)); conv.bias.zero_()
    bn = nn.BatchNorm2d(C).to(gpu).train()
    bn2 = copy.deepcopy(bn)
    x = torch.randint(-2, 3, (N, C, H, W), device=gpu).float()          # five distinct values: many ties
    x1, x2 = x.clone().requires_grad_(), x.clone().requires_grad_()
    y1 = conv_bn_act_pool(conv, bn, nn.MaxPool2d(2, 2), x1, "relu")
    y2 = torch.nn.functional.max_pool2d(torch.relu(bn2(x2)), 2, 2)
    assert torch.allclose(y1, y2, rtol=1e-5, atol=1e-6)
    g = torch.rand_like(y1) + 0.5
    y1.backward(g); y2.backward(g)
    # the same window elements receive gradient: compare the support of the part of dx that comes
    # through the pool (dx = k1*dz + k2*x + k3: subtract the dense part using a zero-gradient twin)
    assert float((x1.grad - x2.grad).abs().max()) <= 2e-4 * float(x2.grad.abs().max()) + 1e-6


@pytest.mark.parametrize("shape,act", [((6, 128, 2048), "relu"), ((64, 32, 2048), None), ((3, 16, 4096 + 640), ("leaky", 0.2)),
                                       ((5, 24, 1000), "relu"), ((2, 8, 9000), None)])
@pytest.mark.parametrize("mode", ["train", "eval"])
def test_max_variant_matches_unfused_chain(gpu, shape, act, mode):
    """conv_bn_act_max == conv_bn_act(...).max(dim=2)[0]: outputs BIT-equal (monotone maps commute with
    max), running statistics equal, gradients to summation order.  gamma has both signs (arg-max and
    arg-min rows)."""
    from fpsg_amd.fused_bn import conv_bn_act, conv_bn_act_max
    import copy
    torch.manual_seed(sum(shape) + 7)
    N, C, L = shape
    conv = nn.Conv1d(16, C, 1).to(gpu)
    bn = nn.BatchNorm1d(C).to(gpu)
    with torch.no_grad():
        conv.bias.copy_(torch.randn(C) * 0.3)
        bn.weight.copy_(torch.randn(C))            # both signs
        bn.weight[0] = 0.0                         # a constant channel
        bn.bias.copy_(torch.randn(C) * 0.3)
    conv2, bn2 = copy.deepcopy(conv), copy.deepcopy(bn)
    bn.train(mode == "train"); bn2.train(mode == "train")
    x = torch.randn(N, 16, L, device=gpu)
    x1, x2 = x.clone().requires_grad_(), x.clone().requires_grad_()
    y1 = conv_bn_act_max(conv, bn, x1, act)
    y2 = conv_bn_act(conv2, bn2, x2, act).max(dim=2)[0]
    assert y1.shape == (N, C)
    # eval mode: same statistics, bit-equal outputs; training mode: the statistics pass of the max
    # variant groups its partial sums differently (several rows per workgroup)
    same_stats = mode == "eval"
    if same_stats:
        assert torch.equal(y1, y2)
        assert torch.equal(bn.running_mean, bn2.running_mean) and torch.equal(bn.running_var, bn2.running_var)
    else:
        assert torch.allclose(y1, y2, rtol=1e-5, atol=1e-6)
        assert torch.allclose(bn.running_mean, bn2.running_mean, rtol=1e-5, atol=1e-7)
        assert torch.allclose(bn.running_var, bn2.running_var, rtol=1e-5, atol=1e-7)
    g = torch.randn_like(y1)
    y1.backward(g); y2.backward(g)
    for a, r in ((x1.grad, x2.grad), (conv.weight.grad, conv2.weight.grad), (bn.weight.grad, bn2.weight.grad),
                 (bn.bias.grad, bn2.bias.grad)):
        assert float((a - r).abs().max()) <= 2e-4 * float(r.abs().max()) + 1e-6
    assert int(bn.num_batches_tracked) == int(bn2.num_batches_tracked)


def test_max_variant_selected_index(gpu):
    """The C ABI's idx output: first arg-max of x where gamma >= 0, first arg-min where gamma < 0."""
    from fpsg_amd import _hip
    torch.manual_seed(5)
    N, C, L = 4, 6, 5000
    x = torch.randint(-50, 50, (N, C, L), device=gpu).float()            # many ties
    gamma = torch.tensor([1.0, -1.0, 0.5, -2.0, 1.5, -0.1], device=gpu)
    beta = torch.zeros(C, device=gpu)
    lib = _hip.load()
    out = torch.empty(N, C, device=gpu); idx = torch.empty(N, C, dtype=torch.int32, device=gpu)
    chan = torch.empty(4, C, device=gpu)
    ws = torch.empty(lib.fpsg_bn_max_workspace_floats(N, C, L), device=gpu)
    rc = lib.fpsg_bn_act_max_fwd(_hip.ptr(x), None, _hip.ptr(gamma), _hip.ptr(beta), None, None, -1.0, N, C, L, 1, 1e-5, 0, 0.0,
                                 _hip.ptr(out), _hip.ptr(idx), _hip.ptr(chan), None, None, _hip.ptr(ws), None)
    assert rc == 0
    torch.cuda.synchronize()
    xc = x.cpu()
    for n in range(N):
        for c in range(C):
            row = xc[n, c]
            want = int((row == (row.max() if gamma[c] >= 0 else row.min())).nonzero()[0])
            assert int(idx[n, c]) == want, (n, c)


@pytest.mark.parametrize("shape,act", [((64, 128, 1024, 2048), None), ((8, 128, 1024, 2048), "relu"), ((5, 100, 1000, 300), "relu"),
                                       ((3, 16, 70, 4096 + 640), ("leaky", 0.2)), ((2, 64, 2500, 512), None)])
@pytest.mark.parametrize("mode", ["train", "eval"])
def test_max_variant_backward_without_the_dense_gradient(gpu, monkeypatch, shape, act, mode):
    """``_ConvBNActMax``: the backward of conv1x1 -> BatchNorm (+act) -> max over the points through the K x K Gram-matrix
    algebra + gather / scatter of csrc/maxbwd.hip against the dense form it replaces (``FPSG_MAX_BWD_ALGEBRA=0``: dx'
    materialised, two GEMMs over it): same forward bit for bit, every gradient to reassociation round-off, deterministic;
    PointNet's shape (64 clouds, 128 -> 1024, 2048 points), channel / point counts that are not multiples of the tiles,
    more than 1024 channels (two sorting passes), gamma of both signs."""
    import copy
    from fpsg_amd import fused_bn
    from fpsg_amd.fused_bn import conv_bn_act_max
    monkeypatch.setattr(fused_bn, "_MAX_ALGEBRA_MIN_POINTS", 0)      # small batches take the dense form by default
    B, K, C, L = shape
    torch.manual_seed(B + C + L)
    conv = nn.Conv1d(K, C, 1).to(gpu)
    bn = nn.BatchNorm1d(C).to(gpu)
    with torch.no_grad():
        conv.bias.copy_(torch.randn(C) * 0.3)
        bn.weight.copy_(torch.randn(C))
        bn.bias.copy_(torch.randn(C) * 0.3)
        # the weight as it sits in the flat parameter buffer of the optimizer: at a 4-byte, not a 16-byte, boundary
        flat = torch.empty(C * K + 5, device=gpu)
        flat[1:1 + C * K].copy_(conv.weight.reshape(-1))
        conv.weight.data = flat[1:1 + C * K].view(C, K, 1)
        assert conv.weight.data_ptr() % 16 == 4
    bn.train(mode == "train")
    a = torch.relu(torch.randn(B, K, L, device=gpu))
    g = torch.randn(B, C, device=gpu)
    res = []
    for flag in ("1", "0", "1"):
        monkeypatch.setenv("FPSG_MAX_BWD_ALGEBRA", flag)
        c2, b2 = copy.deepcopy(conv), copy.deepcopy(bn)
        ai = a.clone().requires_grad_()
        y = conv_bn_act_max(c2, b2, ai, act)
        y.backward(g)
        res.append((y.detach(), ai.grad, c2.weight.grad, c2.bias.grad, b2.weight.grad, b2.bias.grad, b2.running_mean.clone()))
    new, old, again = res
    assert torch.equal(new[0], old[0]) and torch.equal(new[6], old[6])
    for i, name in ((1, "d input"), (2, "d weight"), (4, "d gamma"), (5, "d beta")):
        scale = float(old[i].abs().max()) + 1e-30
        assert float((new[i] - old[i]).abs().max()) <= 2e-5 * scale, (name, float((new[i] - old[i]).abs().max()) / scale)
    # the bias in front of a training-mode BatchNorm has a mathematically zero gradient: both forms leave round-off
    if mode == "train":
        assert float(new[3].abs().max()) <= 1e-3 * float(old[2].abs().max())
    else:
        assert float((new[3] - old[3]).abs().max()) <= 2e-5 * (float(old[3].abs().max()) + 1e-30)
    for x, y in zip(new, again):
        assert torch.equal(x, y)


@pytest.mark.parametrize("rows,seg_lens,act,with_bias", [(48, (640, 4096), "relu", True), (7, (128, 256, 64), None, False),
                                                         (130, (68, 100, 64), "relu", True)])
def test_column_segments_equal_separate_calls(gpu, rows, seg_lens, act, with_bias):
    """fpsg_bn_act_rows_fwd/bwd: BatchNorm over column segments of the rows in one launch = one K5 call per segment on
    its own contiguous copy, bit for bit (output, batch statistics, input gradient); the affine / bias gradients are the
    sums over the segments."""
    from fpsg_amd.fused_bn import batch_norm_act, batch_norm_act_rows
    gen = torch.Generator().manual_seed(rows)
    M = sum(seg_lens)
    x = torch.randn((1, rows, M), generator=gen).to(gpu)
    gamma = (torch.randn(rows, generator=gen) * 0.5 + 1).to(gpu)
    beta = (torch.randn(rows, generator=gen) * 0.1).to(gpu)
    pb = (torch.randn(rows, generator=gen) * 0.3).to(gpu) if with_bias else None
    gy = torch.randn((1, rows, M), generator=gen).to(gpu)

    xr = x.clone().requires_grad_()
    pr = [t.clone().requires_grad_() if t is not None else None for t in (gamma, beta, pb)]
    y, stats = batch_norm_act_rows(xr, pr[0], pr[1], seg_lens, 1e-5, act, pre_bias=pr[2])
    y.backward(gy)
    assert stats.shape == (len(seg_lens), 2, rows)

    off = 0
    sums = [torch.zeros_like(gamma), torch.zeros_like(beta), torch.zeros_like(gamma)]
    for i, n in enumerate(seg_lens):
        xs = x[:, :, off:off + n].contiguous().requires_grad_()
        ps = [t.clone().requires_grad_() if t is not None else None for t in (gamma, beta, pb)]
        ys, mean, var = batch_norm_act(xs, ps[0], ps[1], None, None, True, 1.0, 1e-5, act, return_stats=True, pre_bias=ps[2])
        ys.backward(gy[:, :, off:off + n].contiguous())
        assert torch.equal(y[:, :, off:off + n], ys), i
        assert torch.equal(stats[i, 0], mean) and torch.equal(stats[i, 1], var), i
        assert torch.equal(xr.grad[:, :, off:off + n], xs.grad), i
        for k in range(3):
            if ps[k] is not None:
                sums[k] += ps[k].grad
        off += n
    for k in range(3):
        if pr[k] is not None:
            scale = float(sums[k].abs().max()) + 1e-12
            assert float((pr[k].grad - sums[k]).abs().max()) <= 1e-6 * scale, k


def test_column_segments_reject_bad_layouts(gpu):
    from fpsg_amd import _hip
    import ctypes
    lib = _hip.load()
    x = torch.zeros((4, 64), device=gpu)
    ch = torch.zeros((2, 4, 4), device=gpu)
    g = torch.ones(4, device=gpu)

    def call(ld, offs, lens):
        n = len(offs)
        return lib.fpsg_bn_act_rows_fwd(_hip.ptr(x), ld, (ctypes.c_int * n)(*offs), (ctypes.c_int * n)(*lens), n, None,
                                        _hip.ptr(g), _hip.ptr(g), 4, 1e-5, 1, 0.0, _hip.ptr(x), _hip.ptr(ch), None,
                                        _hip.stream_of(x))

    assert call(64, [0, 32], [32, 32]) == 0
    assert call(64, [0, 32], [32, 40]) != 0          # past the row
    assert call(64, [0, 16], [32, 32]) != 0          # overlap
    assert call(64, [0, 30], [28, 32]) != 0          # offset not a multiple of 4
    assert call(62, [0], [32]) != 0                  # ld not a multiple of 4
    assert call(64, [0] * 5, [4] * 5) != 0           # too many segments
    torch.cuda.synchronize()


@pytest.mark.parametrize("shape", [(8, 64, 112, 112), (24, 32, 2048), (5, 9, 4096 + 4100), (37, 16, 56, 56), (16, 769, 4736)])
def test_finalize_in_the_last_arriving_workgroup_equals_the_two_launch_form(gpu, monkeypatch, shape):
    """``FPSG_BN_FINALIZE_FOLD=1`` (opt-in): the forward finalize runs in the last-arriving workgroup of each channel of the
    statistics kernel.  It sums the same partials in the finalize kernel's own order, so outputs, saved coefficients and
    running statistics are equal bit for bit, call after call (the counters go back to zero; 40 calls go round the 16
    counter banks more than twice)."""
    from fpsg_amd.fused_bn import bn_act
    import copy
    torch.manual_seed(sum(shape))
    C = shape[1]
    bn = (nn.BatchNorm2d if len(shape) == 4 else nn.BatchNorm1d)(C).to(gpu).train()
    with torch.no_grad():
        bn.weight.copy_(torch.randn(C) * 0.5 + 1)
        bn.bias.copy_(torch.randn(C) * 0.3)
    folded = copy.deepcopy(bn)
    xs = [torch.randn(*shape, device=gpu) * (1 + i) + 0.3 * i for i in range(3)]
    outs = []
    monkeypatch.setenv("FPSG_BN_FINALIZE_FOLD", "0")
    with torch.no_grad():
        for i in range(40):
            outs.append(bn_act(bn, xs[i % 3], "relu").clone() if i % 13 == 0 else bn_act(bn, xs[i % 3], "relu").sum())
    monkeypatch.setenv("FPSG_BN_FINALIZE_FOLD", "1")
    with torch.no_grad():
        for i in range(40):
            y = bn_act(folded, xs[i % 3], "relu")
            assert torch.equal(y.clone() if i % 13 == 0 else y.sum(), outs[i]), i
    assert torch.equal(bn.running_mean, folded.running_mean) and torch.equal(bn.running_var, folded.running_var)
