"""PointNet / STN3d on the HIP path (K5 conv+BatchNorm(+ReLU), K5 max-over-points with pre-bias,
1x1 convolutions as batched GEMMs) against goldens produced by the reference's own module
(``/root/reference/src/pointnet/model.py:28-45,214-236``) with the shipped checkpoint:
forward in eval and training mode, running statistics, and the training-mode gradients of the
input and of EVERY parameter (``tests/golden/make_golden.py:gradient_goldens``)."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN

pytestmark = pytest.mark.gpu
CKPT = os.path.join(GOLDEN, "pretrained_pcencoder_pointnet.pt")


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(GOLDEN, "pointnet_goldens.npz"))


@pytest.fixture(scope="module")
def ggold():
    return np.load(os.path.join(GOLDEN, "gradient_goldens.npz"))


def _encoder(gpu):
    from fpsg_amd.point_cloud_net import PCEncoder
    enc = PCEncoder("pointnet")
    enc.load_state_dict(torch.load(CKPT, map_location="cpu", weights_only=True), strict=True)
    return enc.to(gpu)


def _fused_path_taken(monkeypatch):
    """Counts the K5 launches so that a silent library dispatch cannot pass for the HIP path."""
    from fpsg_amd import fused_bn
    seen = {"bn": 0, "max": 0}
    f0, m0 = fused_bn._BNAct.forward, fused_bn._BNActMax.forward

    def f1(*a, **k):
        seen["bn"] += 1
        return f0(*a, **k)

    def m1(*a, **k):
        seen["max"] += 1
        return m0(*a, **k)

    c0 = fused_bn._ConvBNActMax.forward            # conv1x1 + BatchNorm + max as one op (K5m backward)

    def c1(*a, **k):
        seen["max"] += 1
        return c0(*a, **k)

    monkeypatch.setattr(fused_bn._BNAct, "forward", staticmethod(f1))
    monkeypatch.setattr(fused_bn._BNActMax, "forward", staticmethod(m1))
    monkeypatch.setattr(fused_bn._ConvBNActMax, "forward", staticmethod(c1))
    return seen


def _float64_module(tag, gold, ggold=None, train=True):
    """The host port in float64 on the CPU (itself pinned to the reference goldens by
    tests/test_pointnet_cpu.py): the 'truth' that both the reference's fp32 outputs (the goldens)
    and the HIP path are measured against."""
    from fpsg_amd.point_cloud_net import PCEncoder
    enc = PCEncoder("pointnet")
    enc.load_state_dict(torch.load(CKPT, map_location="cpu", weights_only=True), strict=True)
    net = enc.double().train(train).pc_encoder.pointnet_feat_extractor
    x = torch.from_numpy(gold[f"x_{tag}"]).double().requires_grad_()
    feat, trans, _ = net(x)
    if ggold is not None:
        (feat * torch.from_numpy(ggold[f"pn_w_{tag}"]).double()).sum().backward()
    return net, x, feat.detach(), trans.detach()


@pytest.mark.parametrize("tag", ["a", "b"])
@pytest.mark.parametrize("mode", ["eval", "train"])
def test_features_match_reference_on_hip_path(gpu, gold, monkeypatch, tag, mode):
    seen = _fused_path_taken(monkeypatch)
    enc = _encoder(gpu).train(mode == "train")
    net = enc.pc_encoder.pointnet_feat_extractor
    tracked0 = int(net.bn3.num_batches_tracked)
    x = torch.from_numpy(gold[f"x_{tag}"]).to(gpu)
    with torch.no_grad():
        feat, trans, tf = net(x)
    assert tf is None
    assert seen["bn"] == 4 and seen["max"] == 2, seen         # 4 fused BN(+ReLU), 2 fused BN+max
    if mode == "eval":
        np.testing.assert_allclose(trans.cpu().numpy(), gold[f"trans_{tag}_{mode}"], rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(feat.cpu().numpy(), gold[f"feat_{tag}_{mode}"], rtol=1e-4, atol=1e-4)
        with torch.no_grad():
            stn = net.stn(x)
        np.testing.assert_allclose(stn.cpu().numpy(), gold[f"stn_{tag}_eval"], rtol=1e-4, atol=1e-5)
        return
    # training mode: BatchNorm over B = 4 / 2 rows in the STN's fc layers amplifies fp32 round-off; the
    # reference's own fp32 output (the golden) is the yardstick for the distance from a float64 run
    _, _, feat64, trans64 = _float64_module(tag, gold)
    for name, got, ref64, golden in (("trans", trans, trans64, gold[f"trans_{tag}_train"]),
                                     ("feat", feat, feat64, gold[f"feat_{tag}_train"])):
        scale = float(ref64.abs().max())
        e_hip = float((got.cpu().double() - ref64).abs().max()) / scale
        e_ref = float((torch.from_numpy(golden).double() - ref64).abs().max()) / scale
        print(f"PointNet train {tag} {name}: deviation from float64 -- HIP {e_hip:.2e}, reference fp32 {e_ref:.2e}")
        assert e_hip <= 4 * e_ref + 2e-5, (name, e_hip, e_ref)
    np.testing.assert_allclose(net.bn3.running_mean.cpu().numpy(), gold[f"bn3_mean_{tag}_train"],
                               rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(net.stn.bn5.running_var.cpu().numpy(), gold[f"stn_bn5_var_{tag}_train"],
                               rtol=2e-3, atol=1e-6)                  # variance of 4 / 2 values
    assert int(net.bn3.num_batches_tracked) == tracked0 + 1 and int(net.stn.bn3.num_batches_tracked) == tracked0 + 1


@pytest.mark.parametrize("tag", ["a", "b"])
def test_training_gradients_match_reference_per_parameter(gpu, gold, ggold, tag):
    """Every parameter's gradient: the HIP path must sit as close to a float64 run as the reference
    module's own fp32 autograd (gradient_goldens.npz) does -- per-tensor deviations, tests/_gradcheck.py."""
    from _gradcheck import assert_like_yardstick
    enc = _encoder(gpu).train()
    net = enc.pc_encoder.pointnet_feat_extractor
    x = torch.from_numpy(gold[f"x_{tag}"]).to(gpu).requires_grad_()
    w = torch.from_numpy(ggold[f"pn_w_{tag}"]).to(gpu)
    feat, _, _ = net(x)
    (feat * w).sum().backward()
    net64, x64, _, _ = _float64_module(tag, gold, ggold)

    def rows(t, like):
        return t[: like.shape[0]]

    yard = {"input.x.grad": ggold[f"pn_gx_{tag}"]}
    yard.update({n: ggold[f"pn_g_{tag}_{n}"] for n, _ in net.named_parameters()})
    truth = {"input.x.grad": x64.grad}
    truth.update({n: rows(p.grad, yard[n]) for n, p in net64.named_parameters()})
    got = {"input.x.grad": x.grad}
    got.update({n: rows(p.grad, yard[n]) for n, p in net.named_parameters()})
    assert_like_yardstick(got, yard, truth, f"PointNet train {tag}")
