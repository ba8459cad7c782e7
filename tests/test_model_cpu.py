"""Episode loss / evaluation host logic on CPU, with the oracle's Chamfer injected as the
checker (the product's own metric is HIP-only)."""
import numpy as np
import pytest
import torch

from fpsg_amd.engine import TrainStep, build_model, build_optimizer, default_options
from fpsg_amd.episodes import synthetic_episode


@pytest.fixture(scope="module")
def cd(oracle):
    return oracle.make_torch_chamfer()


def _model(cd, **kw):
    torch.manual_seed(0)
    m = build_model(default_options(device="cpu", **kw))
    m.pc_metric = cd
    return m


def test_state_dict_layout_matches_reference_checkpoints(cd):
    m = _model(cd)
    keys = list(m.state_dict())
    assert len(keys) == 581 and sum(p.numel() for p in m.parameters()) == 77445125
    assert "img_encoder.img_feature_extractor.0.weight" in keys
    assert "img_encoder.img_feature_extractor.41.running_var" in keys   # last BN of VGG16-BN
    assert "pc_encoder.pc_encoder.pointnet_feat_extractor.stn.fc3.bias" in keys
    assert "pc_decoder.cluster_pool.3.node_pool.3.conv4.weight" in keys
    assert all(p.requires_grad for p in m.parameters())


def test_loss_dict_contract(cd):
    ep = synthetic_episode(2, 3, n_pts=128, img_size=32, seed=1)
    m = _model(cd, intra_recon=False)
    out = m.loss(ep)
    assert set(out) == {"ttl_loss", "recon_loss", "query_rec_loss", "support_rec_loss"}
    assert out["ttl_loss"].shape == (1,) and out["query_rec_loss"].dim() == 0   # few_shot.py:15,119
    assert float(out["support_rec_loss"].detach()) == 0.0
    m2 = _model(cd, intra_recon=True, query_factor=2.0, support_factor=0.5)
    out2 = m2.loss(ep)
    assert out2["ttl_loss"].dim() == 0 and float(out2["support_rec_loss"].detach()) > 0
    out2 = {k: v.detach() for k, v in out2.items()}
    np.testing.assert_allclose(float(out2["ttl_loss"]),
                               2.0 * float(out2["query_rec_loss"]) + 0.5 * float(out2["support_rec_loss"]), rtol=1e-6)


def test_eval_dict_and_emd_hook(cd):
    m = _model(cd).eval()
    m.emd_metric = lambda a, b: a.norm(dim=-1).sum() + b.norm(dim=-1).sum()   # 0-dim stand-in
    ep = synthetic_episode(2, 2, n_pts=128, img_size=32, seed=2)
    with torch.no_grad():
        out = m._return_reconstruction(ep)
    assert set(out) == {"cd_loss", "emd_loss"} and out["cd_loss"].dim() == 0 and out["emd_loss"].dim() == 0


def test_train_step_updates_every_parameter(cd):
    m = _model(cd, intra_recon=True).train()
    optimizer, _ = build_optimizer(m, default_options(lr=1e-3))
    step = TrainStep(m, optimizer)
    before = [p.detach().clone() for p in m.parameters()]
    ep = synthetic_episode(2, 1, n_pts=128, img_size=32, seed=3)
    losses = [float(step([ep])[0]["ttl_loss"].sum()) for _ in range(2)]
    assert np.isfinite(losses).all() and all(l > 0 for l in losses)
    # gradients live in ONE flat buffer (what the all-reduce operates on) ...
    flat = step.buckets.flat
    assert flat.numel() == 77445125 and float(flat.abs().sum()) > 0
    for p in m.parameters():
        assert p.grad.data_ptr() >= flat.data_ptr() and p.grad.data_ptr() < flat.data_ptr() + flat.numel() * 4
    # ... and every parameter of all three networks moved (all of VGG is trainable, F9)
    assert all(not torch.equal(a, b) for a, b in zip(before, m.parameters()))


def test_constructor_validation(cd):
    from fpsg_amd.few_shot import ImgPCProtoNet
    with pytest.raises(NotImplementedError):
        ImgPCProtoNet(None, None, None, aggregate="bogus")
    with pytest.raises(NotImplementedError):
        ImgPCProtoNet(None, None, None, metric="l2")
    from fpsg_amd.point_cloud_net import PCEncoder
    with pytest.raises(NotImplementedError):
        PCEncoder("transformer")
