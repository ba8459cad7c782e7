"""Host-side PointNet vs goldens produced by the reference's own module + checkpoint."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN

CKPT = os.path.join(GOLDEN, "pretrained_pcencoder_pointnet.pt")


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(GOLDEN, "pointnet_goldens.npz"))


def _encoder():
    from fpsg_amd.point_cloud_net import PCEncoder
    enc = PCEncoder("pointnet")
    missing = enc.load_state_dict(torch.load(CKPT, map_location="cpu", weights_only=True), strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    return enc


def test_checkpoint_loads_key_for_key():
    enc = _encoder()
    assert len(enc.state_dict()) == 58
    assert sum(v.numel() for v in enc.state_dict().values()) == 952593


@pytest.mark.parametrize("tag", ["a", "b"])
@pytest.mark.parametrize("mode", ["eval", "train"])
def test_features_match_reference(gold, tag, mode):
    enc = _encoder()
    enc.train(mode == "train")
    x = torch.from_numpy(gold[f"x_{tag}"])
    net = enc.pc_encoder.pointnet_feat_extractor
    with torch.no_grad():
        feat, trans, tf = net(x)
        stn = net.stn(x) if mode == "eval" else None
    assert tf is None
    np.testing.assert_allclose(trans.numpy(), gold[f"trans_{tag}_{mode}"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(feat.numpy(), gold[f"feat_{tag}_{mode}"], rtol=1e-4, atol=1e-4)
    if stn is not None:
        np.testing.assert_allclose(stn.numpy(), gold[f"stn_{tag}_eval"], rtol=1e-4, atol=1e-5)
    if mode == "train":
        np.testing.assert_allclose(net.bn3.running_mean.numpy(), gold[f"bn3_mean_{tag}_train"],
                                   rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(net.stn.bn5.running_var.numpy(), gold[f"stn_bn5_var_{tag}_train"],
                                   rtol=1e-4, atol=1e-6)


def test_encoder_wrapper_returns_feature_only(gold):
    enc = _encoder().eval()
    with torch.no_grad():
        out = enc(torch.from_numpy(gold["x_a"]))
    assert out.shape == (4, 1024)
    np.testing.assert_allclose(out.numpy(), gold["feat_a_eval"], rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("tag", ["a", "b"])
def test_training_gradients_match_reference(gold, tag):
    """Host port against the reference module's own autograd (gradient_goldens.npz)."""
    gg = np.load(os.path.join(GOLDEN, "gradient_goldens.npz"))
    enc = _encoder().train()
    net = enc.pc_encoder.pointnet_feat_extractor
    x = torch.from_numpy(gold[f"x_{tag}"]).requires_grad_()
    feat, _, _ = net(x)
    (feat * torch.from_numpy(gg[f"pn_w_{tag}"])).sum().backward()
    np.testing.assert_allclose(x.grad.numpy(), gg[f"pn_gx_{tag}"], rtol=1e-3, atol=1e-5 * np.abs(gg[f"pn_gx_{tag}"]).max())
    for name, p in net.named_parameters():
        ref = gg[f"pn_g_{tag}_{name}"]
        got = (p.grad[:64] if p.grad.numel() > 65536 else p.grad).numpy()
        assert np.abs(got - ref).max() <= 1e-3 * np.abs(ref).max() + 1e-6, name
