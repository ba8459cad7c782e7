"""configs[3] (DGCNN encoder) at its full size on the HIP path.

  * K3 with the widest layer input (C = 128) at N = 2048 against the oracle, bit for bit;
  * K4b (fused EdgeConv) at N = 2048 for the four layer shapes (C -> Co = 3->64, 64->64, 64->128,
    128->256) against the oracle-assembled literal chain of reference ``dgcnn/model.py:23-42,63-76``
    (oracle kNN + oracle edge features, then Conv2d / BatchNorm2d(train) / LeakyReLU / max in
    float64 on the CPU): forward, running statistics, input / weight / gamma / beta gradients;
  * the reference module's own training-mode gradients (``gradient_goldens.npz``);
  * B = 64 x N = 2048 (configs[3]'s 2S clouds): shape, determinism of forward + backward, and
    batch independence in eval mode (a cloud's feature does not depend on its neighbours in the batch).
"""
import copy
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, unit_ball_clouds

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("B,C", [(1, 128), (2, 64), (2, 3)])
def test_knn_full_size_bit_exact(gpu, oracle, B, C):
    from fpsg_amd.dgcnn import knn
    rng = np.random.default_rng(1000 + C)
    x = rng.standard_normal((B, C, 2048)).astype(np.float32)
    got = knn(torch.from_numpy(x).to(gpu), 20).cpu().numpy()
    assert np.array_equal(got, oracle.knn(x, 20))


def _literal_chain64(x_cm, idx, conv_w, gamma, beta, slope, oracle, w_out):
    """Reference EdgeConv layer in float64 on the oracle's graph: returns out [B,Co,N] and the
    gradients of sum(out * w_out) w.r.t. x, W, gamma, beta, plus the batch statistics."""
    edge = torch.from_numpy(oracle.edge_feature(x_cm, idx)).double()              # [B,2C,N,k] (model.py:23-42)
    # d(edge)/dx is linear: differentiate through the oracle's gather by its own backward
    edge.requires_grad_()
    W = torch.from_numpy(conv_w).double().requires_grad_()
    g = torch.from_numpy(gamma).double().requires_grad_()
    b = torch.from_numpy(beta).double().requires_grad_()
    y = torch.nn.functional.conv2d(edge, W)                                        # model.py:51 (bias=False)
    mean = y.mean(dim=(0, 2, 3))
    var = y.var(dim=(0, 2, 3), unbiased=False)
    z = (y - mean.view(1, -1, 1, 1)) * torch.rsqrt(var.view(1, -1, 1, 1) + 1e-5) * g.view(1, -1, 1, 1) + b.view(1, -1, 1, 1)
    out = torch.nn.functional.leaky_relu(z, slope).max(dim=-1)[0]                  # model.py:64-65
    (out * torch.from_numpy(w_out).double()).sum().backward()
    gx = oracle.edge_feature_bwd(edge.grad.float().numpy(), idx)                   # float32 scatter of a float64 gradient
    n = y.numel() // y.shape[1]
    return (out.detach(), gx, W.grad, g.grad, b.grad, mean.detach(), (var * n / (n - 1)).detach())


@pytest.mark.parametrize("C,Co", [(3, 64), (64, 64), (64, 128), (128, 256)])
def test_fused_edgeconv_full_size_vs_oracle_chain(gpu, oracle, C, Co):
    from fpsg_amd.dgcnn import _edge_block, edgeconv_fused, knn_int32
    B, N, k = 2, 2048, 20
    rng = np.random.default_rng(C * 31 + Co)
    if C == 3:
        x = unit_ball_clouds(rng, B, N).transpose(0, 2, 1).copy()                  # [B,3,N]
    else:
        x = np.maximum(rng.standard_normal((B, C, N)), -0.3).astype(np.float32)    # LeakyReLU-like features
    torch.manual_seed(C + Co)
    block = _edge_block(2 * C, Co)
    with torch.no_grad():
        block[1].weight.copy_(torch.randn(Co) * 0.7)                                # gammas of both signs: max and min paths
        block[1].bias.copy_(torch.randn(Co) * 0.1)
    block = block.to(gpu).train()
    w_out = rng.standard_normal((B, Co, N)).astype(np.float32)

    x_cm = torch.from_numpy(x).to(gpu)
    idx32 = knn_int32(x_cm, k)
    idx = idx32.cpu().numpy()
    assert np.array_equal(idx, oracle.knn(x, k))                                    # the chain below runs on the oracle's graph
    x_pm = x_cm.transpose(1, 2).contiguous().requires_grad_()
    out = edgeconv_fused(x_pm, idx32, block)                                        # [B,N,Co]
    (out * torch.from_numpy(w_out).to(gpu).transpose(1, 2)).sum().backward()

    conv_w = block[0].weight.detach().cpu().numpy()
    ref_out, ref_gx, ref_gW, ref_gg, ref_gb, ref_mean, ref_var = _literal_chain64(
        x, idx, conv_w, block[1].weight.detach().cpu().numpy(), block[1].bias.detach().cpu().numpy(), 0.2, oracle, w_out)

    def rel(got, ref, l2=False):
        ref = torch.as_tensor(ref).double()
        d = got.double().cpu() - ref
        if l2:
            return float(d.norm() / (ref.norm() + 1e-30))
        return float(d.abs().max() / (ref.abs().max() + 1e-30))

    gdx, gdW = x_pm.grad.transpose(1, 2), block[0].weight.grad
    dev = {"out": rel(out.detach().transpose(1, 2), ref_out),
           "dx": rel(gdx, ref_gx), "dx_l2": rel(gdx, ref_gx, True),
           "dW": rel(gdW, ref_gW), "dW_l2": rel(gdW, ref_gW, True),
           "dgamma": rel(block[1].weight.grad, ref_gg),
           "dbeta": rel(block[1].bias.grad, ref_gb),
           "running_mean": rel(block[1].running_mean, 0.1 * ref_mean),
           "running_var": rel(block[1].running_var, 0.9 + 0.1 * ref_var)}
    print(f"K4b C={C} Co={Co} N=2048 deviation from the float64 chain: {dev}")
    assert dev["out"] <= 1e-4, dev
    assert dev["running_mean"] <= 1e-4 and dev["running_var"] <= 1e-4, dev
    # a max over k = 20 whose two largest candidates differ by less than an fp32 ulp selects another edge than
    # the float64 chain (expected a few times per million outputs): one channel's gradient moves between two
    # points.  Hence a tight L2 bound and a loose max bound.
    assert max(dev["dx_l2"], dev["dW_l2"]) <= 1e-3, dev
    assert max(dev["dx"], dev["dW"], dev["dgamma"], dev["dbeta"]) <= 2e-2, dev


def test_training_gradients_match_reference_module(gpu):
    """Fused path against the reference ``DGCNNfeat``'s own autograd on the golden input."""
    from fpsg_amd.dgcnn import DGCNNfeat
    gold = np.load(os.path.join(GOLDEN, "dgcnn_goldens.npz"))
    gg = np.load(os.path.join(GOLDEN, "gradient_goldens.npz"))
    net = DGCNNfeat()
    net.load_state_dict(torch.load(os.path.join(GOLDEN, "dgcnn_state.pt"), weights_only=True), strict=True)
    net = net.to(gpu).train()
    x = torch.from_numpy(gold["dgcnn_x"]).to(gpu).requires_grad_()
    feat = net(x)
    (feat * torch.from_numpy(gg["dg_w"]).to(gpu)).sum().backward()
    worst = {}
    for name, got in [("x", x.grad)] + [(n, p.grad) for n, p in net.named_parameters()]:
        ref = gg["dg_gx"] if name == "x" else gg[f"dg_g_{name}"]
        got = (got[:64] if got.numel() > 65536 else got).cpu().numpy()
        worst[name] = float(np.abs(got - ref).max() / np.abs(ref).max())
    print("DGCNN gradient deviation from the reference module:", worst)
    import json
    out = os.path.join(os.path.dirname(GOLDEN), os.pardir, "gpurun_out")
    if os.path.isdir(out):      # on the GPU box: the measurement the bounds below rest on (profiles/r04/episode_parity_deviation.jsonl)
        with open(os.path.join(out, "parity_deviation.jsonl"), "a") as f:
            f.write(json.dumps({"test": "dgcnn_gradient_goldens", "max": max(worst.values()),
                                "median": float(np.median(list(worst.values()))), "per_tensor": worst}) + "\n")
    # Measured on MI355X (profiles/r04/episode_parity_deviation.jsonl, "dgcnn_gradient_goldens"): worst tensor 2.7e-6,
    # median 1.4e-6 of the reference module's largest component -- on this golden input the kernel's neighbour lists
    # are the reference's, so only summation order differs.  Bounds = 3x the measurement (rounds 1-3 carried 2e-2 /
    # 5e-3 for near-tie neighbour swaps between torch.matmul's and the kernel's inner products; none occurs here: a
    # swap would move single tensors by ~1e-3 and trip this bound -- then look at the graph, not at the bound).
    assert max(worst.values()) <= 8.1e-6, worst
    assert np.median(list(worst.values())) <= 4.3e-6, worst


def test_b64_shape_determinism_and_batch_independence(gpu):
    from fpsg_amd.dgcnn import DGCNNfeat
    torch.manual_seed(11)
    net = DGCNNfeat().to(gpu)
    with torch.no_grad():
        for mod in net.modules():
            if isinstance(mod, (torch.nn.BatchNorm1d, torch.nn.BatchNorm2d)):
                mod.weight.copy_(torch.randn_like(mod.weight) * 0.6)
                mod.running_mean.copy_(torch.randn_like(mod.running_mean) * 0.05)
                mod.running_var.copy_(torch.rand_like(mod.running_var) + 0.5)
    rng = np.random.default_rng(5)
    x = torch.from_numpy(unit_ball_clouds(rng, 64, 2048)).to(gpu).transpose(1, 2).contiguous()     # [64,3,2048]
    net.train()
    runs = []
    for _ in range(2):
        state = copy.deepcopy(net.state_dict())
        net.zero_grad()
        xi = x.clone().requires_grad_()
        f = net(xi)
        assert f.shape == (64, 1024) and torch.isfinite(f).all()
        f.square().sum().backward()
        runs.append((f.detach().clone(), xi.grad.clone(), net.conv4[0].weight.grad.clone(),
                     net.conv1[1].weight.grad.clone()))
        net.load_state_dict(state)                                                   # same running statistics for run 2
    for a, b in zip(*runs):
        assert torch.equal(a, b)                                                     # deterministic forward + backward
    net.eval()
    with torch.no_grad():
        full = net(x)
        part = net(x[5:7].contiguous())
    assert torch.allclose(full[5:7], part, rtol=1e-4, atol=1e-5), (full[5:7] - part).abs().max()
