"""K6 (Winograd F(m x m,3x3) transforms, m = 2 / 4, through the C ABI + fp32 batched GEMM) against the
reference's own operator, torch.nn.functional.conv2d (Conv2d 3x3 / padding 1 of vgg16_bn,
src/models/image_net.py:14): forward, data gradient, weight gradient.  Yardstick: the same
convolution in float64 on the CPU; tolerance 1e-4 of the tensor's scale (north_star), and the
Winograd result must not be worse than a few times the error of the library's own fp32 result."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

SHAPES = [  # N, C, K, H, W, m
    (2, 8, 5, 4, 6, 2), (1, 3, 7, 2, 2, 2), (3, 16, 16, 14, 14, 2), (2, 32, 24, 28, 30, 2), (5, 64, 48, 56, 56, 2),
    (37, 40, 32, 14, 14, 2),
    (2, 8, 5, 4, 8, 4), (1, 3, 7, 4, 4, 4), (3, 16, 16, 12, 16, 4), (2, 32, 24, 28, 32, 4), (5, 64, 48, 56, 56, 4),
    (7, 256, 128, 28, 28, 4),
    # m = 4 on sides that are even but not multiples of 4 (half-empty last tile row / column; VGG's 14 x 14 stage)
    (3, 16, 16, 14, 14, 4), (2, 32, 24, 18, 12, 4), (5, 40, 32, 14, 20, 4), (37, 24, 16, 14, 14, 4), (1, 8, 8, 2, 6, 4),
    # 64 channels on ragged planes: K6f tiles whole 4x4 blocks only, so these take the three-kernel form
    (3, 64, 64, 14, 14, 4), (2, 64, 32, 30, 18, 4), (2, 32, 64, 14, 22, 4),
]


def _errs(a, ref64):
    scale = float(ref64.abs().max()) + 1e-30
    return float((a.double().cpu() - ref64).abs().max()) / scale


@pytest.mark.parametrize("shape", SHAPES)
def test_forward_and_gradients_match_conv2d(gpu, shape):
    from fpsg_amd.winograd import conv3x3
    N, C, K, H, W, m = shape
    torch.manual_seed(N * 1000 + C)
    x = torch.randn(N, C, H, W)
    w = torch.randn(K, C, 3, 3) * (2.0 / (9 * C)) ** 0.5
    g = torch.randn(N, K, H, W)
    x64, w64 = x.double().requires_grad_(), w.double().requires_grad_()
    y64 = F.conv2d(x64, w64, None, 1, 1)
    y64.backward(g.double())
    xg, wg = x.to(gpu).requires_grad_(), w.to(gpu).requires_grad_()
    y = conv3x3(xg, wg, m)
    y.backward(g.to(gpu))
    xl, wl = x.to(gpu).requires_grad_(), w.to(gpu).requires_grad_()
    yl = F.conv2d(xl, wl, None, 1, 1)
    yl.backward(g.to(gpu))
    for ours, lib, ref in ((y, yl, y64), (xg.grad, xl.grad, x64.grad), (wg.grad, wl.grad, w64.grad)):
        e_ours, e_lib = _errs(ours.detach(), ref.detach()), _errs(lib.detach(), ref.detach())
        assert e_ours <= 1e-4, (shape, e_ours)
        # m = 2 stays within a small multiple of the library's own fp32 error; m = 4 trades about one
        # decimal digit for 4x fewer multiplications (transform constants up to 8 and 1/24)
        assert e_ours <= (max(8 * e_lib, 5e-6) if m == 2 else 4e-5), (shape, e_ours, e_lib)


def test_only_requested_gradients(gpu):
    from fpsg_amd.winograd import conv3x3
    x = torch.randn(2, 4, 8, 8, device=gpu)
    w = torch.randn(6, 4, 3, 3, device=gpu, requires_grad=True)
    conv3x3(x, w).sum().backward()          # first layer of a network: no data gradient
    assert w.grad is not None and x.grad is None
    x2 = torch.randn(2, 4, 8, 8, device=gpu, requires_grad=True)
    conv3x3(x2, w.detach()).sum().backward()
    assert x2.grad is not None


def test_argument_checks(gpu):
    from fpsg_amd.winograd import conv3x3
    from fpsg_amd import _hip
    with pytest.raises(ValueError):
        conv3x3(torch.randn(1, 4, 7, 8, device=gpu), torch.randn(4, 4, 3, 3, device=gpu))      # odd H
    with pytest.raises(ValueError):
        conv3x3(torch.randn(1, 4, 8, 8, device=gpu), torch.randn(4, 5, 3, 3, device=gpu))      # channel mismatch
    lib = _hip.load()
    assert lib.fpsg_wino_input_transform(2, None, 1, 1, 2, 2, None, 0, None) != 0
    assert b"null" in lib.fpsg_last_error()
    x = torch.randn(1, 1, 3, 4, device=gpu)
    assert lib.fpsg_wino_input_transform(2, _hip.ptr(x), 1, 1, 3, 4, _hip.ptr(x), 0, None) != 0           # odd H
    assert lib.fpsg_wino_input_transform(4, _hip.ptr(x), 1, 1, 3, 4, _hip.ptr(x), 0, None) != 0           # odd H, m = 4
    assert lib.fpsg_wino_input_transform(4, _hip.ptr(x), 1, 1, 2, 3, _hip.ptr(x), 0, None) != 0           # odd W
    assert lib.fpsg_wino_input_transform(3, _hip.ptr(x), 1, 1, 2, 4, _hip.ptr(x), 0, None) != 0           # m
    with pytest.raises(ValueError):
        conv3x3(torch.randn(1, 4, 6, 7, device=gpu), torch.randn(4, 4, 3, 3, device=gpu), 4)             # odd W
    # m = 4 on an even side that is not a multiple of 4 is served (half-empty edge tiles), not refused
    y = conv3x3(torch.randn(1, 4, 6, 8, device=gpu), torch.randn(4, 4, 3, 3, device=gpu), 4)
    assert y.shape == (1, 4, 6, 8)


@pytest.mark.parametrize("m", [2, 4])
def test_deterministic(gpu, m):
    from fpsg_amd.winograd import conv3x3
    x = torch.randn(3, 32, 28, 28, device=gpu)
    w = torch.randn(32, 32, 3, 3, device=gpu)
    outs = []
    for _ in range(2):
        xi, wi = x.clone().requires_grad_(), w.clone().requires_grad_()
        y = conv3x3(xi, wi, m)
        y.square().sum().backward()
        outs.append((y.detach().clone(), xi.grad.clone(), wi.grad.clone()))
    assert all(torch.equal(a, b) for a, b in zip(*outs))


def test_tile_size_rule():
    from fpsg_amd.winograd import tile_size
    assert tile_size(56, 56) == 4 and tile_size(28, 28) == 4 and tile_size(14, 14) == 4 and tile_size(30, 28) == 4
    assert tile_size(6, 6) == 2 and tile_size(10, 14) == 2 and tile_size(24, 24) == 2 and tile_size(7, 8) == 2


@pytest.mark.parametrize("shape", [(3, 64, 64, 32, 32), (2, 64, 128, 28, 36), (5, 64, 16, 8, 12), (1, 64, 64, 4, 4),
                                   (37, 64, 64, 56, 56)])
def test_fused_kernel_matches_three_kernel_form(gpu, shape, monkeypatch):
    """K6f (fpsg_wino_conv_fused): same transforms, channel sum in the MFMA's k order -> agrees with
    transform + GEMM + transform to fp32 round-off, forward and (where the output has 64 channels)
    data gradient; the weight gradient rebuilds V from x."""
    from fpsg_amd import winograd as wg
    N, C, K, H, W = shape
    torch.manual_seed(N + K)
    x = torch.randn(N, C, H, W, device=gpu)
    w = torch.randn(K, C, 3, 3, device=gpu) * (2.0 / (9 * C)) ** 0.5
    g = torch.randn(N, K, H, W, device=gpu)
    outs = {}
    for fused in ("0", "1"):
        monkeypatch.setenv("FPSG_WINOGRAD_FUSED", fused)
        xi, wi = x.clone().requires_grad_(), w.clone().requires_grad_()
        y = wg.conv3x3(xi, wi, 4)
        y.backward(g)
        outs[fused] = (y.detach(), xi.grad, wi.grad)
    for a, b in zip(outs["0"], outs["1"]):
        assert float((a - b).abs().max()) <= 2e-5 * float(a.abs().max()) + 1e-7
    x64 = F.conv2d(x.double().cpu(), w.double().cpu(), None, 1, 1)
    assert float((outs["1"][0].double().cpu() - x64).abs().max()) <= 4e-5 * float(x64.abs().max())


def test_fused_kernel_argument_checks(gpu):
    from fpsg_amd import _hip
    lib = _hip.load()
    x = torch.randn(1, 32, 8, 8, device=gpu)
    U = torch.randn(36, 16, 32, device=gpu)
    y = torch.empty(1, 16, 8, 8, device=gpu)
    assert lib.fpsg_wino_conv_fused(_hip.ptr(x), _hip.ptr(U), 1, 32, 16, 8, 8, _hip.ptr(y), None) != 0      # C != 64
    assert b"C must be 64" in lib.fpsg_last_error()


@pytest.mark.parametrize("shape", [(2, 8, 8), (3, 30, 36), (5, 64, 64), (2, 224, 224), (37, 56, 100), (3, 10, 112), (1, 7, 336)])
def test_first_layer_weight_gradient(gpu, shape):
    """K8 (fpsg_conv_first_dw): dw of the 3 -> 64 channel first convolution against conv2d's own
    weight gradient in float64 and against the library's fp32 result."""
    from fpsg_amd.conv_first import conv3x3_first
    N, H, W = shape
    torch.manual_seed(N + H)
    x = torch.randn(N, 3, H, W)
    w = torch.randn(64, 3, 3, 3) * 0.2
    g = torch.randn(N, 64, H, W)
    w64 = w.double().requires_grad_()
    F.conv2d(x.double(), w64, None, 1, 1).backward(g.double())
    wg = w.to(gpu).requires_grad_()
    y = conv3x3_first(x.to(gpu), wg)
    y.backward(g.to(gpu))
    wl = w.to(gpu).requires_grad_()
    yl = F.conv2d(x.to(gpu), wl, None, 1, 1)
    yl.backward(g.to(gpu))
    assert float((y - yl).detach().abs().max()) <= 2e-6 * float(yl.detach().abs().max())      # K8f forward vs the library's
    scale = float(w64.grad.abs().max())
    e_ours = float((wg.grad.double().cpu() - w64.grad).abs().max()) / scale
    e_lib = float((wl.grad.double().cpu() - w64.grad).abs().max()) / scale
    assert e_ours <= 1e-5 and e_ours <= max(4 * e_lib, 2e-6), (shape, e_ours, e_lib)
    # deterministic
    wg2 = w.to(gpu).requires_grad_()
    conv3x3_first(x.to(gpu), wg2).backward(g.to(gpu))
    assert torch.equal(wg.grad, wg2.grad)


@pytest.mark.parametrize("layer", [(64, 64, 224), (64, 128, 112), (128, 128, 112), (256, 256, 56), (512, 512, 28), (512, 512, 14)])
def test_full_size_layers_against_the_library_and_linearity(gpu, layer):
    """The VGG16 layer shapes of the BASELINE workload (37 images = 32 support + 5 query): K6 / K6f
    forward and both gradients against the library convolution, and linearity in the input,
    conv(a*x1 + x2) = a*conv(x1) + conv(x2) -- a size-independent property of every path."""
    from fpsg_amd import winograd as wg
    C, K, H = layer
    N = 37
    torch.manual_seed(C + H)
    x1 = torch.randn(N, C, H, H, device=gpu)
    x2 = torch.randn(N, C, H, H, device=gpu)
    w = (torch.randn(K, C, 3, 3, device=gpu) * (2.0 / (9 * C)) ** 0.5)
    xa, wa = x1.clone().requires_grad_(), w.clone().requires_grad_()
    xb, wb = x1.clone().requires_grad_(), w.clone().requires_grad_()
    ya = wg.conv3x3(xa, wa)
    yb = F.conv2d(xb, wb, None, 1, 1)
    scale = float(yb.detach().abs().max())
    assert float((ya.detach() - yb.detach()).abs().max()) <= 6e-5 * scale
    g = torch.randn_like(yb)
    ya.backward(g); yb.backward(g)
    assert float((xa.grad - xb.grad).abs().max()) <= 6e-5 * float(xb.grad.abs().max())
    assert float((wa.grad - wb.grad).abs().max()) <= 1e-4 * float(wb.grad.abs().max())
    with torch.no_grad():
        lhs = wg.conv3x3(1.5 * x1 + x2, w)
        rhs = 1.5 * wg.conv3x3(x1, w) + wg.conv3x3(x2, w)
    assert float((lhs - rhs).abs().max()) <= 6e-5 * float(rhs.abs().max())


@pytest.mark.parametrize("C,K,H,N", [(64, 64, 112, 6), (64, 128, 56, 6), (128, 128, 56, 6), (256, 256, 28, 24), (128, 256, 14, 90)])
@pytest.mark.parametrize("mode", ["train", "eval"])
def test_bn_relu_folded_into_next_conv_equals_two_ops(gpu, monkeypatch, C, K, H, N, mode):
    """``bn_relu_conv3x3(y, pre_bias, bn, w)`` -- BatchNorm + ReLU applied by the convolution's own input
    transform (fpsg_wino_input_transform_act; fpsg_wino_conv_fused_act for 64 input channels) -- against the
    two ops it replaces, ``conv3x3(bn_act(bn, y, relu, pre_bias), w)``: the activation values are formed with
    the same arithmetic, so output, running statistics and all gradients agree bit for bit (with the BatchNorm
    backward's sums from K5's own pass in both: FPSG_CONV_BWD_STATS=0; the epilogue sums have their own test)."""
    import copy
    monkeypatch.setenv("FPSG_CONV_BWD_STATS", "0")
    from fpsg_amd.fused_bn import bn_act
    from fpsg_amd.winograd import bn_relu_conv3x3, conv3x3
    torch.manual_seed(C + K + H)
    bn = torch.nn.BatchNorm2d(C).to(gpu).train(mode == "train")
    with torch.no_grad():
        bn.weight.copy_(torch.randn(C) * 0.3 + 1)
        bn.bias.copy_(torch.randn(C) * 0.1)
        bn.running_mean.copy_(torch.randn(C) * 0.1)
        bn.running_var.copy_(torch.rand(C) + 0.5)
    bn2 = copy.deepcopy(bn)
    y = torch.randn(N, C, H, H, device=gpu)
    pb = (torch.randn(C, device=gpu) * 0.05)
    w = torch.randn(K, C, 3, 3, device=gpu) * 0.05
    g = torch.randn(N, K, H, H, device=gpu)
    res = []
    for fold, mod in ((True, bn), (False, bn2)):
        yi, pbi, wi = y.clone().requires_grad_(), pb.clone().requires_grad_(), w.clone().requires_grad_()
        out = bn_relu_conv3x3(yi, pbi, mod, wi) if fold else conv3x3(bn_act(mod, yi, "relu", pre_bias=pbi), wi)
        out.backward(g)
        res.append((out.detach(), yi.grad, pbi.grad, wi.grad, mod.weight.grad, mod.bias.grad, mod.running_mean.clone(),
                    mod.running_var.clone(), int(mod.num_batches_tracked)))
    for a, b in zip(res[0][:-1], res[1][:-1]):
        assert torch.equal(a, b)
    assert res[0][-1] == res[1][-1]


def test_trunk_with_and_without_the_fold(gpu, monkeypatch):
    """The whole VGG16-BN trunk with the fold on and off (112x112: conv1_2 and conv2_2 take it): the library
    kernels in between are not bit-reproducible run to run, so the yardstick is a second run without it."""
    import copy
    from fpsg_amd import winograd
    from fpsg_amd.image_net import ImageEncoderWarpper
    torch.manual_seed(4)
    base = ImageEncoderWarpper("vgg_16").to(gpu).train()
    x = torch.rand(6, 3, 112, 112, device=gpu) * 2 - 1
    w = torch.randn(6, 512, device=gpu)
    calls = {"n": 0}
    orig = winograd._BNReluConv3x3.forward

    def counted(*a, **k):
        calls["n"] += 1
        return orig(*a, **k)

    monkeypatch.setattr(winograd._BNReluConv3x3, "forward", staticmethod(counted))

    monkeypatch.setenv("FPSG_CONV_STATS", "0")       # same statistics kernels in both arms (see the next test)
    monkeypatch.setenv("FPSG_CONV_BWD_STATS", "0")

    def run(fold):
        monkeypatch.setenv("FPSG_BN_FOLD", fold)
        calls["n"] = 0
        net = copy.deepcopy(base)
        out = net(x)
        (out * w).sum().backward()
        assert calls["n"] == (2 if fold == "1" else 0), (fold, calls)
        return out.detach(), torch.cat([p.grad.reshape(-1) for p in net.parameters()])

    o1, g1 = run("1")
    o0, g0 = run("0")
    o0b, g0b = run("0")
    noise_o = float((o0 - o0b).abs().max()) + 1e-6 * float(o0.abs().max())
    noise_g = float((g0 - g0b).norm()) + 1e-5 * float(g0.norm())
    assert float((o1 - o0).abs().max()) <= 3 * noise_o
    assert float((g1 - g0).norm()) <= 3 * noise_g


def test_trunk_with_batchnorm_statistics_from_the_convolutions(gpu, monkeypatch):
    """The VGG16-BN trunk (training mode) with the BatchNorm statistics accumulated by the producing convolutions'
    epilogues (K6f, K6's output transform) against K5's own statistics passes: the sums are grouped differently, so
    the two agree to accumulated fp32 round-off (13 BatchNorm layers) on the features and the running statistics.  The
    gradients of a randomly initialised trunk at batch 6 amplify any fp32-level perturbation to the percent level
    (tests/_gradcheck.py: the reference's own fp32 arithmetic is that far from float64), so they are bounded loosely
    here; the single-layer test above holds them to 2e-5.  The epilogue form must really be taken."""
    import copy
    from fpsg_amd import winograd
    from fpsg_amd.image_net import ImageEncoderWarpper
    torch.manual_seed(5)
    base = ImageEncoderWarpper("vgg_16").to(gpu).train()
    x = torch.rand(6, 3, 112, 112, device=gpu) * 2 - 1
    w = torch.randn(6, 512, device=gpu)
    used = {"n": 0}
    orig_out, orig_fused = winograd._output, winograd._fused_stats

    def out_counted(m, M, N, H, W, stats_bias=None, want_parts=False):
        used["n"] += 1 if want_parts else 0
        return orig_out(m, M, N, H, W, stats_bias, want_parts)

    def fused_counted(*a):
        used["n"] += 1
        return orig_fused(*a)

    monkeypatch.setattr(winograd, "_output", out_counted)
    monkeypatch.setattr(winograd, "_fused_stats", fused_counted)

    def run(flag):
        monkeypatch.setenv("FPSG_CONV_STATS", flag)
        used["n"] = 0
        net = copy.deepcopy(base)
        out = net(x)
        (out * w).sum().backward()
        rm = torch.cat([b.reshape(-1) for n_, b in net.named_buffers() if "running" in n_])
        return out.detach(), torch.cat([p.grad.reshape(-1) for p in net.parameters()]), rm, used["n"]

    o1, g1, r1, n1 = run("1")
    o0, g0, r0, n0 = run("0")
    assert n0 == 0 and n1 == 3, (n0, n1)        # 6 x 112x112: conv1_2, conv2_1 (K6f) and conv2_2 (K6) are large enough
    assert float((o1 - o0).abs().max()) <= 2e-4 * float(o0.abs().max())
    assert float((g1 - g0).norm()) <= 5e-2 * float(g0.norm())
    assert float(torch.dot(g1, g0) / (g1.norm() * g0.norm())) > 0.999
    assert float((r1 - r0).abs().max()) <= 1e-5 * float(r0.abs().max())


@pytest.mark.parametrize("m,shape", [(4, (3, 8, 12, 20)), (2, (5, 16, 6, 10)), (4, (37, 32, 28, 28)), (2, (37, 24, 14, 14)),
                                     (4, (2, 5, 112, 112)), (4, (37, 24, 14, 14)), (4, (3, 8, 18, 12)), (4, (5, 6, 6, 10))])
def test_output_transform_delivers_batchnorm_partial_sums(gpu, m, shape):
    """fpsg_wino_output_transform_stats: the same pixels as the plain output transform, and per channel the
    partial sums of (y + bias) and (y + bias)^2 over each workgroup's tiles (ragged last workgroup included)."""
    from fpsg_amd import winograd as wg
    N, K, H, W = shape
    torch.manual_seed(H + K)
    P = N * -(-H // m) * -(-W // m)
    Mt = torch.randn((m + 2) ** 2, K, P, device=gpu)
    b = torch.randn(K, device=gpu)
    y_ref = wg._output(m, Mt, N, H, W)
    for bias in (b, None):
        y, parts = wg._output(m, Mt, N, H, W, bias, True)
        assert torch.equal(y, y_ref)
        assert parts.shape == (K, (P + 255) // 256, 2)
        v = y.double() + (bias.double().view(1, -1, 1, 1) if bias is not None else 0.0)
        s0, s1 = v.sum((0, 2, 3)), (v * v).sum((0, 2, 3))
        got = parts.double().sum(1)
        n = N * H * W
        scale = float(v.abs().max())
        assert float((got[:, 0] - s0).abs().max()) <= 2e-6 * n * scale
        assert float((got[:, 1] - s1).abs().max()) <= 2e-6 * n * scale * scale
        y2, parts2 = wg._output(m, Mt, N, H, W, bias, True)
        assert torch.equal(parts, parts2)                      # deterministic


@pytest.mark.parametrize("C,K,H,N", [(128, 128, 56, 6), (32, 48, 28, 24), (64, 64, 56, 5), (64, 128, 36, 3)])
def test_batchnorm_statistics_from_the_convolution_epilogue(gpu, C, K, H, N):
    """conv -> BatchNorm -> ReLU -> conv / pool with the first convolution's epilogue delivering the BatchNorm
    statistics (``parts``) against the same ops with K5's own statistics pass: the partial sums are grouped
    differently (fp32 per workgroup, fp64 across), so values agree to fp32 round-off, not bit for bit."""
    import copy
    from fpsg_amd.fused_bn import _BNActPool
    from fpsg_amd.winograd import bn_relu_conv3x3, conv3x3
    torch.manual_seed(C + H)
    x = torch.randn(N, C, H, H, device=gpu)
    w0 = torch.randn(C, C, 3, 3, device=gpu) * (2.0 / (9 * C)) ** 0.5
    w1 = torch.randn(K, C, 3, 3, device=gpu) * (2.0 / (9 * C)) ** 0.5
    pb = torch.randn(C, device=gpu) * 0.05
    bn = torch.nn.BatchNorm2d(C).to(gpu).train()
    with torch.no_grad():
        bn.weight.copy_(torch.randn(C) * 0.3 + 1)
        bn.bias.copy_(torch.randn(C) * 0.1)
    res = []
    for use in (True, False):
        mod = copy.deepcopy(bn)
        xi = x.clone().requires_grad_()
        if use:
            y, parts = conv3x3(xi, w0, stats_bias=pb, want_parts=True)
            assert parts is not None and not parts.requires_grad
        else:
            y, parts = conv3x3(xi, w0), None
        out = bn_relu_conv3x3(y, pb, mod, w1, parts=parts)
        mom = 0.1
        pooled = _BNActPool.apply(y, mod.weight, mod.bias, None, None, True, mod.eps, 1, 0.0, pb, -1.0, parts)
        (out.sum() * 0.01 + (pooled * pooled).sum()).backward()
        res.append((out.detach(), pooled.detach(), xi.grad, mod.weight.grad, mod.running_mean.clone(), mod.running_var.clone()))
    for a, b_ in zip(res[0], res[1]):
        scale = float(b_.abs().max()) + 1e-30
        assert float((a - b_).abs().max()) <= 2e-5 * scale, float((a - b_).abs().max()) / scale


@pytest.mark.parametrize("shape", [(3, 64, 64, 32, 32), (2, 64, 128, 28, 36), (37, 64, 32, 56, 56), (1, 64, 16, 4, 4)])
@pytest.mark.parametrize("act", [False, True])
def test_fused_kernel_delivers_batchnorm_partial_sums(gpu, shape, act):
    """fpsg_wino_conv_fused_stats: the pixels of the plain / activating K6f bit for bit, and per output channel the
    partial sums of (y + out_bias) and its square per workgroup."""
    from fpsg_amd import winograd as wg
    N, C, K, H, W = shape
    torch.manual_seed(K + H)
    x = torch.randn(N, C, H, W, device=gpu)
    w = torch.randn(K, C, 3, 3, device=gpu) * 0.05
    U = wg._filter(4, w, False)
    ob = torch.randn(K, device=gpu)
    chan = pb = None
    if act:
        chan = torch.randn(4, C, device=gpu) * 0.5 + 1.0
        pb = torch.randn(C, device=gpu) * 0.1
    y_ref = wg._fused_act(x, chan, pb, U) if act else wg._fused(x, U)
    for bias in (ob, None):
        y, parts = wg._fused_stats(x, chan, pb, U, bias)
        assert torch.equal(y, y_ref)
        v = y.double() + (bias.double().view(1, -1, 1, 1) if bias is not None else 0.0)
        s0, s1 = v.sum((0, 2, 3)), (v * v).sum((0, 2, 3))
        got = parts.double().sum(1)
        n, scale = N * H * W, float(v.abs().max())
        assert float((got[:, 0] - s0).abs().max()) <= 2e-6 * n * scale
        assert float((got[:, 1] - s1).abs().max()) <= 2e-6 * n * scale * scale
        assert torch.equal(parts, wg._fused_stats(x, chan, pb, U, bias)[1])


@pytest.mark.parametrize("shape", [(2, 8, 8), (3, 30, 36), (1, 5, 4), (5, 64, 64), (2, 224, 224), (37, 56, 100)])
def test_first_layer_forward_and_its_batchnorm_partial_sums(gpu, shape):
    """K8f (fpsg_conv_first_fwd) against the reference's operator, F.conv2d(x, w, None, 1, 1) of
    vgg16_bn.features[0] (src/models/image_net.py:14), in float64; with parts: the same pixels and the partial
    sums of y + bias and its square.  The autograd op keeps K8 for the weight gradient."""
    from fpsg_amd.conv_first import _forward, conv3x3_first
    N, H, W = shape
    torch.manual_seed(H * 7 + W)
    x = torch.randn(N, 3, H, W)
    w = torch.randn(64, 3, 3, 3) * 0.2
    b = torch.randn(64)
    y64 = F.conv2d(x.double(), w.double(), None, 1, 1)
    xg, wg_, bg = x.to(gpu), w.to(gpu), b.to(gpu)
    y, none = _forward(xg, wg_)
    assert none is None
    assert _errs(y, y64) <= 1e-6
    y2, parts = _forward(xg, wg_, bg, True)
    assert torch.equal(y, y2)
    v = y.double() + bg.double().view(1, -1, 1, 1)
    got = parts.double().sum(1)
    n, scale = N * H * W, float(v.abs().max())
    assert float((got[:, 0] - v.sum((0, 2, 3))).abs().max()) <= 2e-6 * n * scale
    assert float((got[:, 1] - (v * v).sum((0, 2, 3))).abs().max()) <= 2e-6 * n * scale * scale
    assert torch.equal(parts, _forward(xg, wg_, bg, True)[1])
    # through autograd: same forward values, K8 weight gradient against float64
    wr = wg_.clone().requires_grad_()
    out = conv3x3_first(xg, wr)
    assert torch.equal(out, y)
    g = torch.randn(N, 64, H, W)
    out.backward(g.to(gpu))
    w64 = w.double().requires_grad_()
    F.conv2d(x.double(), w64, None, 1, 1).backward(g.double())
    assert _errs(wr.grad, w64.grad) <= 1e-5


@pytest.mark.parametrize("shape", [(2, 64, 64, 16, 16), (3, 64, 32, 8, 48), (1, 64, 16, 4, 16), (5, 64, 128, 28, 32),
                                   (37, 64, 64, 56, 48)])
@pytest.mark.parametrize("act", [False, True])
def test_fused_weight_gradient_matches_three_kernel_form(gpu, shape, act):
    """K6w (fpsg_wino_dw_fused): dU of a 64-input-channel layer against the three-kernel form (input transform,
    grad-output transform, batched GEMM over the tiles) and the resulting dw against conv2d's weight gradient in
    float64; with ``act`` the input is a pre-BatchNorm tensor.  Ranges without steps, image borders, one tile row
    per image and several images per range are among the shapes.  Deterministic."""
    from fpsg_amd import winograd as wg
    N, C, K, H, W = shape
    torch.manual_seed(K + H + W)
    x = torch.randn(N, C, H, W, device=gpu)
    gy = torch.randn(N, K, H, W, device=gpu)
    chan = pb = None
    a = x
    if act:
        chan = torch.randn(4, C, device=gpu) * 0.5 + 1.0
        pb = torch.randn(C, device=gpu) * 0.1
        a = torch.relu((x + pb.view(1, -1, 1, 1)) * chan[0].view(1, -1, 1, 1) + chan[1].view(1, -1, 1, 1))
    assert wg._can_fuse_dw(4, 64, 64, 37, 224, 224) and wg._can_fuse_dw(4, 64, 128, 37, 112, 112)
    assert not wg._can_fuse_dw(4, 64, 64, 2, 224, 224) and not wg._can_fuse_dw(4, 128, 128, 37, 112, 112)
    dU = wg._fused_dw(x, chan, pb, gy)
    V = wg._input_act(4, x, chan, pb) if act else wg._input(4, x)
    ref = torch.bmm(wg._grad_output(4, gy), V.transpose(1, 2))
    scale = float(ref.abs().max())
    assert float((dU - ref).abs().max()) <= 2e-5 * scale, float((dU - ref).abs().max()) / scale
    assert torch.equal(dU, wg._fused_dw(x, chan, pb, gy))
    w = torch.randn(K, C, 3, 3, device=gpu)
    gw = wg._filter_grad(4, dU, w)
    w64 = w.double().cpu().requires_grad_()
    F.conv2d(a.double().cpu(), w64, None, 1, 1).backward(gy.double().cpu())
    assert _errs(gw, w64.grad) <= 2e-5


def test_filter_bank_refreshes_every_registered_filter_in_one_launch(gpu, monkeypatch):
    """winograd._FilterBank: filters asked for inside a weights_frozen block are registered; the next block transforms
    them all with one fpsg_wino_filter_transform_batch launch into the same buffers -- bit-identical to the single
    form, after the weights changed in place too; entries a block never asks for are dropped, and FPSG_FILTER_BANK=0
    keeps the single launches."""
    from fpsg_amd import _hip, winograd
    lib = _hip.load()
    winograd._bank.__init__()
    torch.manual_seed(3)
    ws = [torch.randn(K, C, 3, 3, device=gpu) for K, C in ((64, 64), (128, 64), (130, 70), (512, 256))]
    combos = [(4, ws[0], False), (4, ws[0], True), (2, ws[1], False), (4, ws[2], True), (2, ws[3], True), (4, ws[3], False)]

    def single(m, w, flip):
        K, C = w.shape[:2]
        U = torch.empty(((m + 2) ** 2, C, K) if flip else ((m + 2) ** 2, K, C), device=gpu)
        assert lib.fpsg_wino_filter_transform(m, w.data_ptr(), K, C, int(flip), U.data_ptr(), None) == 0
        return U

    calls = {"single": 0, "batch": 0}
    orig = winograd._call

    def counting(name, *a):
        if name == "fpsg_wino_filter_transform":
            calls["single"] += 1
        if name == "fpsg_wino_filter_transform_batch":
            calls["batch"] += 1
        return orig(name, *a)

    monkeypatch.setattr(winograd, "_call", counting)
    with winograd.weights_frozen():                       # block 1: single launches, registration
        first = [winograd._filter(m, w, f) for m, w, f in combos]
    assert calls == {"single": 6, "batch": 0} and len(winograd._bank.entries) == 6
    for w in ws:
        w.mul_(1.5).add_(0.1)                             # "optimizer step": same memory, new values
    with winograd.weights_frozen():                       # block 2: one launch, the same buffers
        assert calls["batch"] == 1
        again = [winograd._filter(m, w, f) for m, w, f in combos]
        assert calls["single"] == 6
        for (m, w, f), U0, U1 in zip(combos, first, again):
            assert U1.data_ptr() == U0.data_ptr()
            assert torch.equal(U1, single(m, w, f))
    with winograd.weights_frozen():                       # block 3 asks for two of them only
        winograd._filter(*combos[0]); winograd._filter(*combos[5])
    with winograd.weights_frozen():                       # block 4: the others are gone
        assert len(winograd._bank.entries) == 2 and calls["batch"] == 3
        assert torch.equal(winograd._filter(*combos[5]), single(*combos[5]))
    assert winograd._bank.lookup((ws[0].data_ptr(), 4, False), ws[0], False) is None      # outside a block: not valid
    winograd._bank.__init__()
    monkeypatch.setenv("FPSG_FILTER_BANK", "0")
    n0 = calls["single"]
    for _ in range(2):
        with winograd.weights_frozen():
            winograd._filter(*combos[0])
    assert calls["single"] == n0 + 2 and not winograd._bank.entries


@pytest.mark.parametrize("m,shape", [(4, (6, 128, 56, 56)), (2, (5, 48, 50, 30)), (4, (37, 32, 28, 28)), (4, (3, 7, 8, 12)),
                                     (4, (37, 24, 14, 14)), (4, (3, 8, 18, 12))])
def test_output_transform_delivers_batchnorm_backward_sums(gpu, m, shape):
    """fpsg_wino_output_transform_bwd_stats: y equals the plain output transform bit for bit, and the per-workgroup
    partial sums add up (float64) to sum(dz) and sum(dz * xhat) with dz = y * [bn(xpre + bias) > 0], as
    fpsg_bn_act_bwd's own pass would form them (1e-5 of the sums' scale); last workgroup partly empty included."""
    from fpsg_amd import _hip
    from fpsg_amd import winograd as wg
    lib = _hip.load()
    N, K, H, W = shape
    torch.manual_seed(H + K)
    P = N * -(-H // m) * -(-W // m)
    M = torch.randn((m + 2) ** 2, K, P, device=gpu)
    xpre = torch.randn(N, K, H, W, device=gpu)
    pb = torch.randn(K, device=gpu) * 0.1
    scale, shift = torch.randn(K, device=gpu) * 0.5 + 1, torch.randn(K, device=gpu) * 0.3
    mean, rstd = torch.randn(K, device=gpu) * 0.1, torch.rand(K, device=gpu) + 0.5
    chan = torch.stack([scale, shift, mean, rstd]).contiguous()
    y0 = wg._output(m, M, N, H, W)
    y1, parts = wg._output_bwd_stats(m, M, N, H, W, xpre, pb, chan)
    assert torch.equal(y0, y1)
    assert parts.shape == (K, lib.fpsg_wino_stats_parts(m, N, H, W), 2)
    x = (xpre + pb.view(1, K, 1, 1)).double()
    z = x * scale.view(1, K, 1, 1).double() + shift.view(1, K, 1, 1).double()
    dz = torch.where(z > 0, y0.double(), torch.zeros_like(z))
    xhat = (x - mean.view(1, K, 1, 1).double()) * rstd.view(1, K, 1, 1).double()
    s0, s1 = dz.sum(dim=(0, 2, 3)), (dz * xhat).sum(dim=(0, 2, 3))
    got = parts.double().sum(dim=1)
    tol0 = 1e-5 * float(dz.abs().sum(dim=(0, 2, 3)).max())
    tol1 = 1e-5 * float((dz * xhat).abs().sum(dim=(0, 2, 3)).max())
    assert float((got[:, 0] - s0).abs().max()) <= tol0 and float((got[:, 1] - s1).abs().max()) <= tol1


@pytest.mark.parametrize("C,K,H,N", [(128, 128, 56, 6), (32, 48, 28, 24), (128, 64, 36, 16)])
def test_batchnorm_backward_sums_from_the_data_gradient_convolution(gpu, monkeypatch, C, K, H, N):
    """BatchNorm -> ReLU -> conv as one node: its backward with the sums of the BatchNorm backward delivered by the
    data-gradient convolution's output transform (default) against K5's own pass over (x, dy)
    (FPSG_CONV_BWD_STATS=0): every gradient to 2e-5 of its scale (the sums are grouped differently)."""
    import copy
    from fpsg_amd.winograd import bn_relu_conv3x3
    torch.manual_seed(C + H)
    y = torch.randn(N, C, H, H, device=gpu)
    w = torch.randn(K, C, 3, 3, device=gpu) * (2.0 / (9 * C)) ** 0.5
    pb = torch.randn(C, device=gpu) * 0.05
    bn = torch.nn.BatchNorm2d(C).to(gpu).train()
    with torch.no_grad():
        bn.weight.copy_(torch.randn(C) * 0.3 + 1)
        bn.bias.copy_(torch.randn(C) * 0.1)
    g = torch.randn(N, K, H, H, device=gpu)
    res = []
    for flag in ("1", "0"):
        monkeypatch.setenv("FPSG_CONV_BWD_STATS", flag)
        mod = copy.deepcopy(bn)
        yi, wi, pbi = y.clone().requires_grad_(), w.clone().requires_grad_(), pb.clone().requires_grad_()
        out = bn_relu_conv3x3(yi, pbi, mod, wi)
        out.backward(g)
        res.append((out.detach(), yi.grad, wi.grad, mod.weight.grad, mod.bias.grad, pbi.grad))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][2], res[1][2])
    for a, b_ in zip(res[0][:-1], res[1][:-1]):
        scale = float(b_.abs().max()) + 1e-30
        assert float((a - b_).abs().max()) <= 2e-5 * scale, float((a - b_).abs().max()) / scale
    # the gradient of a bias in front of a training-mode BatchNorm is zero: both forms leave only the round-off of
    # sum(dx) over N*H*W values
    bound = 1e-6 * float(res[1][1].abs().sum(dim=(0, 2, 3)).max())
    assert float(res[0][-1].abs().max()) <= bound and float(res[1][-1].abs().max()) <= bound


def test_graph_replay_reads_refreshed_filters_and_refuses_a_stale_bank(gpu, monkeypatch):
    """a hipGraph captured inside a weights_frozen block holds no filter-transform launches and reads the
    bank's pinned buffers.  After the weights change (an optimizer step), a replay inside a new block equals the eager
    convolution with the NEW weights -- also when FPSG_FILTER_BANK was switched off after the capture -- and a replay
    outside a refreshed block is refused by the check TrainStep runs before g.replay()."""
    from fpsg_amd import winograd
    winograd._bank.__init__()
    torch.manual_seed(11)
    w = torch.randn(128, 128, 3, 3, device=gpu) * 0.05
    x = torch.randn(2, 128, 32, 32, device=gpu)
    static_x = x.clone()
    with winograd.weights_frozen():                                   # block 1: registers the filter
        winograd.conv3x3(static_x, w)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    with winograd.weights_frozen():                                   # block 2: refresh, then capture
        with torch.cuda.stream(side):
            winograd.conv3x3(static_x, w)                             # warm-up on the capture stream
            with torch.cuda.graph(g, stream=side):
                static_y = winograd.conv3x3(static_x, w)
    torch.cuda.current_stream().wait_stream(side)
    assert winograd._bank.has_pinned()
    ref0 = torch.nn.functional.conv2d(x.double(), w.double(), padding=1)
    w.mul_(-0.7).add_(0.01)                                           # "optimizer.step()": same memory, new values
    ref1 = torch.nn.functional.conv2d(x.double(), w.double(), padding=1)
    with pytest.raises(RuntimeError, match="replay outside"):
        winograd.check_bank_before_replay()                           # no block open: the buffers hold the old filters
    for env in ("1", "0"):                                            # the bank refreshes pinned entries either way
        monkeypatch.setenv("FPSG_FILTER_BANK", env)
        with winograd.weights_frozen():
            winograd.check_bank_before_replay()
            g.replay()
            torch.cuda.synchronize()
            got = static_y.double()
        assert float((got - ref1).abs().max()) < 1e-3 * float(ref1.abs().max())
        assert float((got - ref0).abs().max()) > 0.1 * float(ref0.abs().max())       # and it is not the old result


@pytest.mark.parametrize("shape", [(3, 224, 224), (5, 96, 128), (2, 100, 136)])
def test_first_layer_weight_gradient_with_the_batchnorm_backward_folded_in(gpu, shape):
    """fpsg_bn_act_bwd_coef + fpsg_conv_first_dw_fold (K8 forms dy from the convolution's output, the activation gradient
    and K5's coefficients while it stages a tile) against fpsg_bn_act_bwd + fpsg_conv_first_dw: the same dw, dgamma,
    dbeta bit for bit -- 112-pixel and 64-pixel segments, rows that are not whole segments."""
    from fpsg_amd import _hip
    lib = _hip.load()
    N, H, W = shape
    g = torch.Generator().manual_seed(N * H)
    x = torch.randn(N, 3, H, W, generator=g).to(gpu)
    y = torch.randn(N, 64, H, W, generator=g).to(gpu)
    ga = torch.randn(N, 64, H, W, generator=g).to(gpu)
    pb = (torch.randn(64, generator=g) * 0.1).to(gpu)
    gamma = (torch.randn(64, generator=g) * 0.5 + 1).to(gpu)
    beta = (torch.randn(64, generator=g) * 0.1).to(gpu)
    st = torch.cuda.current_stream().cuda_stream
    chan = torch.empty(4, 64, device=gpu)
    ws = torch.empty(lib.fpsg_bn_workspace_floats(N, 64, H * W), device=gpu)
    assert lib.fpsg_bn_stats(y.data_ptr(), pb.data_ptr(), gamma.data_ptr(), beta.data_ptr(), None, None, -1.0, N, 64, H * W,
                             1, 1e-5, chan.data_ptr(), None, None, ws.data_ptr(), None, 0, st) == 0, lib.fpsg_last_error()
    # the two-kernel chain
    dy = torch.empty_like(y)
    dg0, db0, coef0 = torch.empty(64, device=gpu), torch.empty(64, device=gpu), torch.empty(3, 64, device=gpu)
    assert lib.fpsg_bn_act_bwd(y.data_ptr(), pb.data_ptr(), ga.data_ptr(), chan.data_ptr(), N, 64, H * W, 1, 1, 0.0,
                               dy.data_ptr(), dg0.data_ptr(), db0.data_ptr(), None, coef0.data_ptr(), ws.data_ptr(), st) == 0
    dw0 = torch.empty(64, 3, 3, 3, device=gpu)
    ws1 = torch.empty(lib.fpsg_conv_first_dw_workspace_floats(N, H, W), device=gpu)
    assert lib.fpsg_conv_first_dw(x.data_ptr(), dy.data_ptr(), N, 3, 64, H, W, dw0.data_ptr(), ws1.data_ptr(), st) == 0
    # the folded form
    dg1, db1, coef1 = torch.empty(64, device=gpu), torch.empty(64, device=gpu), torch.empty(3, 64, device=gpu)
    assert lib.fpsg_bn_act_bwd_coef(y.data_ptr(), pb.data_ptr(), ga.data_ptr(), chan.data_ptr(), N, 64, H * W, 1, 1, 0.0,
                                    dg1.data_ptr(), db1.data_ptr(), coef1.data_ptr(), ws.data_ptr(), st) == 0, lib.fpsg_last_error()
    dw1 = torch.empty(64, 3, 3, 3, device=gpu)
    assert lib.fpsg_conv_first_dw_fold(x.data_ptr(), y.data_ptr(), ga.data_ptr(), chan.data_ptr(), coef1.data_ptr(),
                                       pb.data_ptr(), N, 3, 64, H, W, dw1.data_ptr(), ws1.data_ptr(), st) == 0, lib.fpsg_last_error()
    assert torch.equal(dg0, dg1) and torch.equal(db0, db1) and torch.equal(coef0, coef1)
    assert torch.equal(dw0, dw1)
    # and against float64 autograd of the chain conv -> (+bias) -> BN -> relu, with ga as the upstream gradient
    w = torch.randn(64, 3, 3, 3, generator=g, dtype=torch.float64, requires_grad=True)
    x64 = x.double().cpu()
    yc = F.conv2d(x64, w, None, 1, 1)
    # (the kernels were given an arbitrary y; the float64 check uses the matching one)
    y32 = yc.detach().float().to(gpu).contiguous()
    assert lib.fpsg_bn_stats(y32.data_ptr(), pb.data_ptr(), gamma.data_ptr(), beta.data_ptr(), None, None, -1.0, N, 64, H * W,
                             1, 1e-5, chan.data_ptr(), None, None, ws.data_ptr(), None, 0, st) == 0
    assert lib.fpsg_bn_act_bwd_coef(y32.data_ptr(), pb.data_ptr(), ga.data_ptr(), chan.data_ptr(), N, 64, H * W, 1, 1, 0.0,
                                    dg1.data_ptr(), db1.data_ptr(), coef1.data_ptr(), ws.data_ptr(), st) == 0
    assert lib.fpsg_conv_first_dw_fold(x.data_ptr(), y32.data_ptr(), ga.data_ptr(), chan.data_ptr(), coef1.data_ptr(),
                                       pb.data_ptr(), N, 3, 64, H, W, dw1.data_ptr(), ws1.data_ptr(), st) == 0
    act = F.relu(F.batch_norm(yc + pb.double().cpu().view(1, -1, 1, 1), None, None, gamma.double().cpu(), beta.double().cpu(),
                              True, 0.0, 1e-5))
    act.backward(ga.double().cpu())
    assert _errs(dw1, w.grad) <= 2e-4, _errs(dw1, w.grad)


def test_trunk_with_the_stem_as_one_function_equals_two_functions(gpu, monkeypatch):
    """ImageEncoderWarpper in training mode with conv1_1 + BatchNorm + ReLU + conv1_2 as ONE autograd function
    (winograd._StemConvBNReluConv: no dy of conv1_1's output) against FPSG_STEM_FOLD=0 (conv3x3_first + _BNReluConv3x3):
    outputs, running statistics and every parameter's gradient bit for bit, except conv1_1's bias gradient (round-off
    of an exactly cancelled sum in the two-function form, 0 in the fused one)."""
    import copy
    from fpsg_amd.image_net import ImageEncoderWarpper
    from fpsg_amd import winograd
    torch.manual_seed(4)
    base = ImageEncoderWarpper().to(gpu).train()
    x = torch.rand(3, 3, 224, 224, device=gpu) * 2 - 1
    gfeat = torch.randn(3, 512, device=gpu)
    res = {}
    for fold in ("1", "0"):
        monkeypatch.setenv("FPSG_STEM_FOLD", fold)
        net = copy.deepcopy(base)
        assert winograd.stem_applies(x, net.img_feature_extractor[0], net.img_feature_extractor[1],
                                     net.img_feature_extractor[3]) == (fold == "1")
        out = net(x)
        out.backward(gfeat)
        res[fold] = (out.detach(), {n: p.grad.clone() for n, p in net.named_parameters()},
                     {n: b.clone() for n, b in net.named_buffers()})
    o1, g1, b1 = res["1"]
    o0, g0, b0 = res["0"]
    assert torch.equal(o1, o0)
    for n in b0:
        assert torch.equal(b1[n], b0[n]), n
    for n in g0:
        if n == "img_feature_extractor.0.bias":
            assert float(g1[n].abs().max()) == 0.0
            assert float(g0[n].abs().max()) <= 1e-4 * float(g0["img_feature_extractor.0.weight"].abs().max())
            continue
        assert torch.equal(g1[n], g0[n]), n


@pytest.mark.parametrize("m,shape", [(4, (3, 128, 112, 112)), (4, (37, 40, 28, 28)), (4, (5, 24, 14, 14)), (4, (2, 8, 18, 12)),
                                     (2, (3, 16, 14, 14)), (4, (1, 3, 4, 4)), (4, (37, 256, 56, 56))])
def test_both_transforms_of_an_output_gradient_from_one_pass(gpu, m, shape, monkeypatch):
    """fpsg_wino_grad_transforms (the weight gradient's A g A^T tiles are the interior of the data gradient's B^T d B
    patches: one read of the gradient) against the two separate launches: bit for bit, ragged planes and the tensors
    beyond the non-temporal threshold included; and conv3x3's gradients with the switch on and off."""
    from fpsg_amd import winograd as wg
    torch.manual_seed(shape[1] + shape[2])
    gy = torch.randn(*shape, device=gpu)
    V, dM = wg._grad_transforms(m, gy)
    assert torch.equal(V, wg._input(m, gy)) and torch.equal(dM, wg._grad_output(m, gy))
    N, K, H, W = shape
    if K * H * W * N > 4e7:
        return
    C = 16
    x = torch.randn(N, C, H, W, device=gpu)
    w = torch.randn(K, C, 3, 3, device=gpu) * 0.1
    res = []
    for sw in ("1", "0"):
        monkeypatch.setenv("FPSG_GRAD_TRANSFORMS", sw)
        xr, wr = x.clone().requires_grad_(), w.clone().requires_grad_()
        wg.conv3x3(xr, wr, m).backward(gy)
        res.append((xr.grad, wr.grad))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])


@pytest.mark.parametrize("m,shape", [(4, (37, 24, 28, 28)), (4, (3, 16, 14, 14)), (2, (3, 16, 14, 14)), (4, (2, 8, 18, 12)),
                                     (4, (1, 3, 4, 4)), (4, (8, 32, 16, 16))])
@pytest.mark.parametrize("align", [32, 64, 7])
def test_padded_rows_of_the_transform_domain_tensors(gpu, m, shape, align, monkeypatch):
    """Round 5: the transform-domain tensors are [A*A, channels, row stride] with the stride rounded up to whole 128-byte
    lines (FPSG_WINO_ROW_ALIGN, default 32 floats).  Against the dense form (align 1): the first P columns of every
    transform are the dense transform's bit for bit, the pad columns are zeros, an output transform reads a padded
    product as it reads a dense one, and the convolution's value and gradients keep their float64 bounds."""
    from fpsg_amd import winograd as wg
    N, C, H, W = shape
    torch.manual_seed(C + H)
    x = torch.randn(*shape, device=gpu)
    chan = torch.stack([torch.rand(C, device=gpu) + 0.5, torch.randn(C, device=gpu), torch.zeros(C, device=gpu),
                        torch.ones(C, device=gpu)]).contiguous()
    pb = torch.randn(C, device=gpu)
    P = N * wg._tiles(H, W, m)
    out = {}
    for a in (1, align):
        monkeypatch.setenv("FPSG_WINO_ROW_ALIGN", str(a))
        Vg, dMg = wg._grad_transforms(m, x)
        out[a] = (wg._input(m, x), wg._input_act(m, x, chan, pb), wg._grad_output(m, x), Vg, dMg)
    Ps = (P + align - 1) // align * align
    for dense, padded in zip(out[1], out[align]):
        assert dense.shape[2] == P and padded.shape[2] == Ps
        assert torch.equal(padded[:, :, :P], dense)
        assert int(torch.count_nonzero(padded[:, :, P:])) == 0
    K = 8
    Md = torch.randn((m + 2) ** 2, K, P, device=gpu)
    Mp = torch.full(((m + 2) ** 2, K, Ps), float("nan"), device=gpu)        # the pad columns are never read
    Mp[:, :, :P] = Md
    y0 = wg._output(m, Md, N, H, W)
    assert torch.equal(wg._output(m, Mp, N, H, W), y0)
    bias = torch.randn(K, device=gpu)
    ys, parts = wg._output(m, Mp, N, H, W, bias, True)
    yd, parts_d = wg._output(m, Md, N, H, W, bias, True)
    assert torch.equal(ys, y0) and torch.equal(parts, parts_d)
    # the convolution through padded rows: value and gradients against float64
    w = torch.randn(K, C, 3, 3) * (2.0 / (9 * C)) ** 0.5
    g = torch.randn(N, K, H, W)
    x64, w64 = x.double().cpu().requires_grad_(), w.double().requires_grad_()
    y64 = F.conv2d(x64, w64, None, 1, 1)
    y64.backward(g.double())
    xg, wgr = x.clone().requires_grad_(), w.to(gpu).requires_grad_()
    y = wg.conv3x3(xg, wgr, m)
    y.backward(g.to(gpu))
    for ours, ref in ((y, y64), (xg.grad, x64.grad), (wgr.grad, w64.grad)):
        assert _errs(ours.detach(), ref.detach()) <= (1e-5 if m == 2 else 4e-5)


def test_row_stride_arguments(gpu, monkeypatch):
    from fpsg_amd import _hip, winograd as wg
    lib = _hip.load()
    x = torch.randn(1, 1, 4, 4, device=gpu)
    V = torch.empty(36, 1, 8, device=gpu)
    assert lib.fpsg_wino_input_transform(4, _hip.ptr(x), 1, 1, 4, 4, _hip.ptr(V), -3, None) != 0      # stride below P
    assert b"row stride" in lib.fpsg_last_error()
    with pytest.raises(ValueError):
        wg._output(4, torch.randn(36, 2, 3, device=gpu), 1, 8, 8)                                      # 4 tiles, 3 columns
    monkeypatch.setenv("FPSG_WINO_ROW_ALIGN", "0")
    with pytest.raises(ValueError):
        wg.row_stride(100)


SPLIT_SHAPES = [(7, 256, 128, 28, 28, 4), (3, 128, 256, 56, 56, 4), (5, 512, 512, 14, 14, 4), (2, 160, 136, 28, 36, 4),
                (3, 128, 128, 14, 14, 2)]


@pytest.mark.parametrize("shape", SPLIT_SHAPES)
def test_forward_and_gradients_with_split_operand_products(gpu, shape, monkeypatch):
    """``FPSG_GEMM_SPLIT=1`` (opt-in, round 5): the transform-domain products of the layers with at least 128 channels on
    the bf16 matrix pipe with exactly split fp32 operands (K10, ``fpsg_gemm_split``) instead of the library's fp32
    GEMMs.  The bounds of ``test_forward_and_gradients_match_conv2d`` UNCHANGED, and against the library-GEMM form of
    the same Winograd convolution: no further from float64 than 1.5x that form (+ 1e-7 of the scale)."""
    from fpsg_amd import gemm_split, winograd
    from fpsg_amd.winograd import conv3x3
    N, C, K, H, W, m = shape
    torch.manual_seed(N * 1000 + C)
    x = torch.randn(N, C, H, W)
    w = torch.randn(K, C, 3, 3) * (2.0 / (9 * C)) ** 0.5
    g = torch.randn(N, K, H, W)
    x64, w64 = x.double().requires_grad_(), w.double().requires_grad_()
    y64 = F.conv2d(x64, w64, None, 1, 1)
    y64.backward(g.double())
    res = {}
    for flag in ("0", "1"):
        monkeypatch.setenv("FPSG_GEMM_SPLIT", flag)
        assert gemm_split.enabled() == (flag == "1")
        xg, wg = x.to(gpu).requires_grad_(), w.to(gpu).requires_grad_()
        y = conv3x3(xg, wg, m)
        y.backward(g.to(gpu))
        res[flag] = [_errs(t.detach(), r.detach()) for t, r in ((y, y64), (xg.grad, x64.grad), (wg.grad, w64.grad))]
    for e_split, e_lib in zip(res["1"], res["0"]):
        assert e_split <= 1e-4 and e_split <= (4e-5 if m == 4 else 5e-6 + 8 * e_lib), (shape, res)
        assert e_split <= 1.5 * e_lib + 1e-7, (shape, res)


def test_split_products_are_used_only_from_128_channels(gpu, monkeypatch):
    """Below 128 channels (and with the switch off) the products stay with the library: bit-identical results."""
    from fpsg_amd.winograd import conv3x3
    torch.manual_seed(3)
    x = torch.randn(3, 64, 28, 28, device=gpu)
    w = torch.randn(96, 64, 3, 3, device=gpu) * 0.05
    monkeypatch.setenv("FPSG_GEMM_SPLIT", "0")
    monkeypatch.setenv("FPSG_WINOGRAD_FUSED", "0")       # the three-kernel form (64 input channels would take K6f)
    a = conv3x3(x, w, 4)
    monkeypatch.setenv("FPSG_GEMM_SPLIT", "1")
    b = conv3x3(x, w, 4)
    assert torch.equal(a, b)
