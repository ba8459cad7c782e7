"""Decoder: reference state-dict layout / parameter budget, and the split first layer equals
the reference's repeat + concat formulation (point_cloud_net.py:97-112) on the same grid."""
import torch

from fpsg_amd.engine import default_options
from fpsg_amd.point_cloud_net import PCDecoder, PrimitiveNode


def _decoder(**kw):
    torch.manual_seed(0)
    return PCDecoder(conf=default_options(device="cpu", **kw))


def test_parameter_budget_and_keys():
    dec = _decoder()
    assert sum(p.numel() for p in dec.parameters()) == 61775804
    node = dec.cluster_pool[0].node_pool[0]
    assert sum(p.numel() for p in node.parameters()) == 3856539
    assert sum(p.numel() for p in dec.cluster_pool[0].deformer.parameters()) == 17795
    keys = set(dec.state_dict())
    for k in ("cluster_pool.0.deformer.conv1.weight", "cluster_pool.3.deformer.bn2.running_var",
              "cluster_pool.2.node_pool.3.conv4.bias", "cluster_pool.1.node_pool.0.bn3.num_batches_tracked"):
        assert k in keys
    assert dec.cluster_pool[0].node_pool[0].conv1.weight.shape == (1539, 1539, 1)


def _reference_formulation(dec, hidden, grids):
    """What the reference computes, written with its tensor shapes."""
    outs = []
    for ci, cluster in enumerate(dec.cluster_pool):
        deformed = [cluster.deformer(g) for g in grids[ci]]
        x = hidden.unsqueeze(2).repeat(1, 1, cluster.pts_per_node).contiguous()
        node_out = [cluster.node_pool[i](torch.cat((x, deformed[i]), dim=1)).unsqueeze(1)
                    for i in range(cluster.num_nodes)]
        outs.append(torch.cat(node_out, dim=3).squeeze(1))
    return torch.cat(outs, dim=2).transpose(1, 2).contiguous()


def test_split_first_layer_equals_reference_formulation():
    import copy
    dec = _decoder()
    ref = copy.deepcopy(dec)
    B = 3
    hidden = torch.randn(B, 1536)
    grids = [cl.sample_grids(B, "cpu", torch.Generator().manual_seed(ci)) for ci, cl in enumerate(dec.cluster_pool)]
    out = dec(hidden, grid=grids)
    exp = _reference_formulation(ref, hidden, grids)
    assert out.shape == (B, 2048, 3) and out.is_contiguous()
    assert torch.allclose(out, exp, rtol=1e-4, atol=2e-5), (out - exp).abs().max()
    # running statistics advanced identically (one update per deformer / node call)
    a, b = dec.state_dict(), ref.state_dict()
    for k in a:
        assert torch.allclose(a[k].float(), b[k].float(), rtol=1e-4, atol=1e-5), k
    # and the gradients agree: both formulations are compared with a float64 run of the same
    # network (training-mode BatchNorm over 3x128 values is ill-conditioned, so the yardstick
    # is the error of the REFERENCE formulation itself in fp32)
    dec64 = copy.deepcopy(dec).double()
    h64 = hidden.detach().double().requires_grad_()
    dec64(h64, grid=[[g.double() for g in c] for c in grids]).square().sum().backward()
    truth = h64.grad.float()
    hidden.requires_grad_()
    dec.zero_grad(); ref.zero_grad()
    dec(hidden, grid=grids).square().sum().backward()
    g_split = hidden.grad.clone(); hidden.grad = None
    _reference_formulation(ref, hidden, grids).square().sum().backward()
    g_ref = hidden.grad
    scale = truth.abs().max()
    err_split, err_ref = (g_split - truth).abs().max(), (g_ref - truth).abs().max()
    assert err_split <= 3e-3 * scale and err_split <= 2.0 * err_ref + 1e-6 * scale, (err_split, err_ref, scale)
    w64 = dec64.cluster_pool[1].node_pool[2].conv1.weight.grad.float()
    w1 = dec.cluster_pool[1].node_pool[2].conv1.weight.grad
    w2 = ref.cluster_pool[1].node_pool[2].conv1.weight.grad
    ws = w64.abs().max()
    assert (w1 - w64).abs().max() <= 3e-3 * ws and (w2 - w64).abs().max() <= 3e-3 * ws


def test_generator_reproducibility_and_fresh_grids():
    dec = _decoder()
    h = torch.randn(2, 1536)
    a = dec(h, generator=torch.Generator().manual_seed(3))
    b = dec(h, generator=torch.Generator().manual_seed(3))
    c = dec(h)
    assert torch.equal(a, b) and not torch.equal(a, c)       # a fresh grid per forward (F11)
    assert a.abs().max() <= 1.0                               # tanh output


def test_other_shapes():
    dec = PCDecoder(conf=default_options(device="cpu", bottleneck_size=1024), num_pts=1024)  # config-1 wiring
    assert dec(torch.randn(2, 1024)).shape == (2, 1024, 3)
    one = PCDecoder(conf=default_options(device="cpu", num_nodes=1, num_clusters=2, bottleneck_size=64), num_pts=256)
    assert one(torch.randn(1, 64)).shape == (1, 256, 3)
    node = PrimitiveNode(default_options(), 67)
    x = torch.randn(2, 67, 16)
    x[:, :64] = x[:, :64, :1]                                  # latent constant over the patch
    node.eval()
    assert torch.allclose(node(x), node.forward_split(x[:, :64, 0], x[:, 64:]), atol=1e-5)


def test_batched_equals_looped():
    """The batched evaluation (stacked weights, one GEMM + one grouped BatchNorm per layer) is
    the looped per-patch evaluation: outputs, gradients and every running statistic."""
    import copy
    dec = _decoder()
    with torch.no_grad():
        for mod in dec.modules():
            if isinstance(mod, torch.nn.BatchNorm1d):
                mod.weight.copy_(torch.randn_like(mod.weight) * 0.5 + 1)
                mod.bias.copy_(torch.randn_like(mod.bias) * 0.1)
    loop = copy.deepcopy(dec)
    loop.batched = False
    assert dec.batched
    for mode in ("train", "train", "eval"):
        dec.train(mode == "train"); loop.train(mode == "train")
        h1 = torch.randn(3, 1536, requires_grad=True)
        h2 = h1.detach().clone().requires_grad_()
        grids = dec.sample_grids(3, "cpu", torch.Generator().manual_seed(4))
        a, b = dec(h1, grid=grids), loop(h2, grid=grids)
        assert torch.allclose(a, b, rtol=1e-4, atol=5e-5)
        dec.zero_grad(); loop.zero_grad()
        a.square().sum().backward(); b.square().sum().backward()
        scale = h2.grad.abs().max()
        assert (h1.grad - h2.grad).abs().max() <= 3e-3 * scale
        ga = dec.cluster_pool[2].node_pool[1].conv2.weight.grad
        gb = loop.cluster_pool[2].node_pool[1].conv2.weight.grad
        assert (ga - gb).abs().max() <= 3e-3 * gb.abs().max()
        gd = dec.cluster_pool[3].deformer.conv1.weight.grad       # shared by the cluster's 4 patches
        ge = loop.cluster_pool[3].deformer.conv1.weight.grad
        assert (gd - ge).abs().max() <= 3e-3 * ge.abs().max()
    sa, sb = dec.state_dict(), loop.state_dict()
    for k in sa:
        assert torch.allclose(sa[k].float(), sb[k].float(), rtol=1e-5, atol=1e-6), k
    assert int(sa["cluster_pool.0.deformer.bn1.num_batches_tracked"]) == 8     # 4 calls per forward
    assert int(sa["cluster_pool.0.node_pool.0.bn1.num_batches_tracked"]) == 2
