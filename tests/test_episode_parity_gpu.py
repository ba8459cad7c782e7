"""End-to-end parity of the episode step: the product (GPU: MIOpen/hipBLASLt networks + HIP
Chamfer / kNN / EdgeConv / EMD through the C ABI) against the CPU port (PyTorch-CPU networks
+ the C oracle) on identical weights, (image, point-cloud) pairs and decoder patch grids.
north_star tolerance for the losses: 1e-4 relative fp32 on the distance ops themselves (checked
bit-exactly elsewhere); here the whole network sits in front, so the yardstick is fp32
round-off through ~40 layers with training-mode BatchNorm."""
import copy

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _fixed_grids(model, sizes, device):
    return {b: model.pc_decoder.sample_grids(b, device, torch.Generator(device=device).manual_seed(100 + b))
            for b in sizes}


def _pin_grids(model, grids):
    orig = model.pc_decoder.forward
    model.pc_decoder.forward = lambda h, grid=None, generator=None, pack=None: orig(h, grid=grids[h.size(0)], pack=pack)
    pair = model.pc_decoder.forward_pair        # the episode's two decodes side by side (its fallback calls .forward)
    model.pc_decoder.forward_pair = lambda a, b, generator=None, pack=None, grids_=None: pair(
        a, b, pack=pack, grids=(grids[a.size(0)], grids[b.size(0)]))


def _record(name, payload):
    """Measured deviations are printed and, on the GPU box, left under gpurun_out/ for DESIGN.md."""
    import json
    import os
    print(name, json.dumps(payload))
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(out):
        with open(os.path.join(out, "parity_deviation.jsonl"), "a") as f:
            f.write(json.dumps({"test": name, **payload}) + "\n")


def _chamfer64(p1, p2, w1=1.0, w2=1.0):
    """Kaolin's chamfer_distance semantics (few_shot.py:57,110) in float64 torch, differentiable
    through the arg-min: the 'truth' metric of the float64 run."""
    d = (p1.unsqueeze(2) - p2.unsqueeze(1)).square().sum(-1)          # [B,N,M]
    return w1 * d.min(dim=2)[0].mean(dim=1) + w2 * d.min(dim=1)[0].mean(dim=1)



def _group_grad_distances(dev_model, cpu_model):
    """Relative L2 distance ||g_hip - g_cpu32|| / ||g_cpu32|| per module group, every parameter of the group
    concatenated: the HIP path's gradients against the fp32 CPU port's (reference arithmetic), no float64."""
    out = {}
    for top in ("img_encoder", "pc_encoder", "pc_decoder"):
        a = torch.cat([p.grad.reshape(-1).double().cpu() for n, p in dev_model.named_parameters()
                       if n.startswith(top) and p.grad is not None])
        b = torch.cat([p.grad.reshape(-1).double() for n, p in cpu_model.named_parameters()
                       if n.startswith(top) and p.grad is not None])
        assert a.numel() == b.numel() and a.numel() > 0, top
        out[top] = float((a - b).norm() / b.norm())
    return out


# Loss bounds (relative, against the fp32 CPU port = reference arithmetic + C-oracle Chamfer): north_star's 1e-4 for
# every loss and both tile sizes.  Measured on MI355X at 224x224 images (profiles/r03/episode_parity_deviation.jsonl,
# DESIGN.md section 5): S = 4, Q = 2 training mode <= 2e-5 with the default F(4x4,3x3) tiles and <= 1.2e-5 with F(2x2);
# eval mode <= 1.2e-7; the configs[2]-sized episode (S = 32, Q = 5) below.
LOSS_TOL = {"4": {"ttl_loss": 1e-4, "query_rec_loss": 1e-4, "support_rec_loss": 1e-4},
            "2": {"ttl_loss": 1e-4, "query_rec_loss": 1e-4, "support_rec_loss": 1e-4}}


@pytest.mark.parametrize("wino_m", ["4", "2"])
@pytest.mark.parametrize("mode", ["train", "eval"])
def test_pointnet_episode_loss_and_gradients(gpu, oracle, monkeypatch, mode, wino_m):
    if wino_m == "2":
        monkeypatch.setenv("FPSG_WINOGRAD_M", "2")
    _pointnet_episode(gpu, oracle, mode, wino_m, S=4, Q=2, intra=True)


def test_pointnet_episode_with_split_operand_products(gpu, oracle, monkeypatch):
    """``FPSG_GEMM_SPLIT=1`` (opt-in: K10, the trunk's transform-domain products on the bf16 matrix pipe with exactly split
    fp32 operands): the 4-shot training episode's losses and every gradient under the SAME bounds as the default path."""
    monkeypatch.setenv("FPSG_GEMM_SPLIT", "1")
    _pointnet_episode(gpu, oracle, "train", "4", S=4, Q=2, intra=True)


def test_config1_one_shot_episode_loss_and_gradients(gpu, oracle):
    """BASELINE configs[1] -- 1-way 1-shot 1-query, no intra-reconstruction (few_shot.py:75-129 with S = Q = 1): the
    image trunk's training-mode BatchNorm runs over TWO images (support + query; the ad pair is a second call), the
    decoder's over one cloud's 128 patch points -- the reference's most fragile arithmetic.  Losses and every
    parameter's gradient of the HIP path against the fp32 CPU port and a float64 run, same yardstick rule as the
    4-shot episode's."""
    _pointnet_episode(gpu, oracle, "train", "4", S=1, Q=1, intra=False)


def _pointnet_episode(gpu, oracle, mode, wino_m, S, Q, intra):
    from _gradcheck import assert_like_yardstick
    from fpsg_amd.engine import build_model, default_options
    from fpsg_amd.episodes import synthetic_episode
    torch.manual_seed(1)
    cpu = build_model(default_options(device="cpu", intra_recon=intra)).train(mode == "train")
    dev = copy.deepcopy(cpu).to(gpu)
    cpu64 = copy.deepcopy(cpu).double()
    cpu.pc_metric = oracle.make_torch_chamfer()
    cpu64.pc_metric = _chamfer64
    ep = synthetic_episode(S, Q, n_pts=2048, img_size=224, seed=3)         # the real image size: F(4x4) on 4 stages
    ep_gpu = {k: (v.to(gpu) if torch.is_tensor(v) else v) for k, v in ep.items()}
    ep64 = {k: (v.double() if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in ep.items()}
    grids_cpu = _fixed_grids(cpu, (S, Q), "cpu")
    grids_gpu = {b: [[t.to(gpu) for t in c] for c in g] for b, g in grids_cpu.items()}
    grids_64 = {b: [[t.double() for t in c] for c in g] for b, g in grids_cpu.items()}
    _pin_grids(cpu, grids_cpu)
    _pin_grids(dev, grids_gpu)
    _pin_grids(cpu64, grids_64)
    out_c = cpu.loss(ep)
    out_g = dev.loss(ep_gpu)
    out_t = cpu64.loss(ep64)
    measured = {"mode": mode, "wino_m": wino_m, "S": S, "Q": Q, "intra_recon": intra}
    loss_keys = ("query_rec_loss", "support_rec_loss", "ttl_loss") if intra else ("query_rec_loss", "ttl_loss")
    if not intra:       # few_shot.py:119: the support term is the zero holder
        assert float(out_g["support_rec_loss"].sum()) == 0.0 and float(out_c["support_rec_loss"].sum()) == 0.0
        assert tuple(out_g["ttl_loss"].shape) == tuple(out_c["ttl_loss"].shape)
    for key in loss_keys:
        a, b, t = (float(o[key].detach().sum()) for o in (out_c, out_g, out_t))
        measured[key] = abs(a - b) / abs(a)
        measured[key + "_hip_vs_f64"] = abs(b - t) / abs(t)
        measured[key + "_cpu32_vs_f64"] = abs(a - t) / abs(t)
    stats = None
    if mode == "train":
        out_c["ttl_loss"].sum().backward()
        out_g["ttl_loss"].sum().backward()
        out_t["ttl_loss"].sum().backward()
        named = lambda m: {n: p.grad for n, p in m.named_parameters()}
        # every parameter tensor's gradient: HIP path vs float64, with the fp32 CPU port (the reference's
        # arithmetic) as the yardstick -- tests/_gradcheck.py
        # per module group: relative L2 distance from the float64 gradient, HIP path and fp32 CPU port
        for top in ("img_encoder", "pc_encoder", "pc_decoder"):
            cat = lambda m: torch.cat([p.grad.reshape(-1).double().cpu() for n, p in m.named_parameters() if n.startswith(top)])
            t64 = cat(cpu64)
            measured[f"grad_l2[{top}]_hip_vs_f64"] = float((cat(dev) - t64).norm() / t64.norm())
            measured[f"grad_l2[{top}]_cpu32_vs_f64"] = float((cat(cpu) - t64).norm() / t64.norm())
        _record("episode_parity_groups", measured)
        # every parameter tensor's gradient: HIP path vs float64, with the fp32 CPU port (the reference's
        # arithmetic) as the yardstick -- tests/_gradcheck.py.  A 6-image training-mode BatchNorm network is
        # chaotic at the percent level in fp32 whatever the arithmetic (the CPU port itself: median 1 %, worst
        # tensor 8 % from float64), so this end-to-end bound is statistical; the per-module tests
        # (test_pointnet_gpu, test_decoder_gpu, test_dgcnn_size_gpu, test_winograd_gpu, test_bnact_gpu) are tight.
        aside = {}
        stats = assert_like_yardstick(named(dev), named(cpu), named(cpu64), f"episode train m={wino_m} S={S} Q={Q}",
                                      factor=5.0, hard_max=0.5, cancelled_max=5e-2, ill_conditioned_ok=(S == 1),
                                      set_aside=aside)
        if S == 1:
            # pin WHAT is set aside.  Behind a training-mode BatchNorm over two samples only ENCODER tensors
            # (and the biases whose gradient a BatchNorm cancels) can be ill-conditioned; a decoder weight never is --
            # the decoder's BatchNorms run over one cloud's 128 patch points.
            groups = {}
            for n in aside["ill_conditioned"]:
                groups[n.split(".")[0]] = groups.get(n.split(".")[0], 0) + 1
            measured["set_aside_ill_conditioned"] = groups
            measured["set_aside_cancelled"] = len(aside["cancelled"])
            n_params = {top: sum(1 for n, _ in dev.named_parameters() if n.startswith(top))
                        for top in ("img_encoder", "pc_encoder", "pc_decoder")}
            assert groups.get("pc_decoder", 0) == 0, aside["ill_conditioned"]
            assert all(groups.get(t, 0) <= n_params[t] for t in ("img_encoder", "pc_encoder")), (groups, n_params)
            # (measured: every encoder tensor -- weights too -- has an exactly cancelled float64 gradient behind the
            # two-sample BatchNorm; in the decoder only biases in front of a BatchNorm may)
            assert all(n.startswith(("img_encoder.", "pc_encoder.")) or n.endswith(".bias") for n in aside["cancelled"]), \
                aside["cancelled"]
        measured["grad_dev_hip"], measured["grad_dev_cpu32"] = stats
        for top in ("img_encoder", "pc_encoder", "pc_decoder"):
            # (1-shot: behind a BatchNorm over a batch of two the encoders' true gradients are O(eps) and both fp32 runs
            # hold round-off: their group distances are ~1e7 on both sides and say nothing)
            if S == 1 and measured[f"grad_l2[{top}]_cpu32_vs_f64"] > 0.5:
                continue
            assert measured[f"grad_l2[{top}]_hip_vs_f64"] <= 5 * measured[f"grad_l2[{top}]_cpu32_vs_f64"] + 1e-3, measured
    _record("episode_parity", measured)
    for key, tol in LOSS_TOL[wino_m].items():
        if key not in loss_keys:
            continue
        assert measured[key] <= tol, measured
        # and no further from the float64 run than the reference arithmetic is, within the same bound
        assert measured[key + "_hip_vs_f64"] <= measured[key + "_cpu32_vs_f64"] + tol, measured


def test_config2_sized_episode_losses(gpu, oracle, monkeypatch):
    """the same comparison ONCE at configs[2] size -- 32-shot, 5-query, intra_recon, 224x224 images,
    69 images and 69 clouds of 2048 points, training-mode BatchNorm -- losses of the HIP path against the fp32 CPU port
    and a float64 run of it (forward only in float64: ~1 min of host time).  Bound: north_star's 1e-4."""
    from fpsg_amd.engine import build_model, default_options
    from fpsg_amd.episodes import synthetic_episode
    torch.manual_seed(7)
    S, Q = 32, 5
    cpu = build_model(default_options(device="cpu", intra_recon=True)).train()
    dev = copy.deepcopy(cpu).to(gpu)
    cpu64 = copy.deepcopy(cpu).double()
    cpu.pc_metric = oracle.make_torch_chamfer()
    cpu64.pc_metric = _chamfer64
    ep = synthetic_episode(S, Q, n_pts=2048, img_size=224, seed=11)
    ep_gpu = {k: (v.to(gpu) if torch.is_tensor(v) else v) for k, v in ep.items()}
    ep64 = {k: (v.double() if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in ep.items()}
    grids_cpu = _fixed_grids(cpu, (S, Q), "cpu")
    _pin_grids(cpu, grids_cpu)
    _pin_grids(dev, {b: [[t.to(gpu) for t in c] for c in g] for b, g in grids_cpu.items()})
    _pin_grids(cpu64, {b: [[t.double() for t in c] for c in g] for b, g in grids_cpu.items()})
    with torch.no_grad():
        out_t = cpu64.loss(ep64)
    out_c, out_g = cpu.loss(ep), dev.loss(ep_gpu)
    measured = {"mode": "train", "wino_m": "4", "S": S, "Q": Q}
    for key in ("query_rec_loss", "support_rec_loss", "ttl_loss"):
        a, b, t = (float(o[key].detach().sum()) for o in (out_c, out_g, out_t))
        measured[key] = abs(a - b) / abs(a)
        measured[key + "_hip_vs_f64"] = abs(b - t) / abs(t)
        measured[key + "_cpu32_vs_f64"] = abs(a - t) / abs(t)
    # one backward of the HIP path and of the fp32 CPU port at this size (no float64 backward): the
    # per-group relative L2 distance of the gradients, bounded at 3x what was measured on MI355X
    # (profiles/r05/episode_parity_deviation.jsonl)
    out_c["ttl_loss"].sum().backward()
    out_g["ttl_loss"].sum().backward()
    dist = _group_grad_distances(dev, cpu)
    for top, v in dist.items():
        measured[f"grad_l2[{top}]_hip_vs_cpu32"] = v
    # where the HIP path's distance from float64 comes from: the same forward with F(2x2,3x3) tiles
    # (transform constants 1/2 instead of 8 and 1/24) and, separately, with the trunk on the library's direct convolutions
    for tag, env in (("m2", {"FPSG_WINOGRAD_M": "2"}), ("library_conv", {"FPSG_WINOGRAD": "0"})):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        with torch.no_grad():
            o = dev.loss(ep_gpu)
        for k in env:
            monkeypatch.delenv(k)
        for key in ("query_rec_loss", "support_rec_loss"):
            b, t = float(o[key].detach().sum()), float(out_t[key].detach().sum())
            measured[f"{key}_hip_vs_f64[{tag}]"] = abs(b - t) / abs(t)
    _record("episode_parity_config2_size", measured)
    for key in ("query_rec_loss", "support_rec_loss", "ttl_loss"):
        assert measured[key] <= 1e-4, measured
        assert measured[key + "_hip_vs_f64"] <= measured[key + "_cpu32_vs_f64"] + 1e-4, measured
    for top, bound in CONFIG2_GRAD_L2_BOUND.items():
        assert dist[top] <= bound, (top, dist)


# 3x the distances measured on MI355X at configs[2] size (S = 32, Q = 5, 224 x 224; profiles/r05/episode_parity_deviation.jsonl)
# measured: PointNet episode 0.0166 / 0.0229 / 0.0113, DGCNN episode (same graphs on both sides) 0.0160 / 0.0056 / 0.0054
CONFIG2_GRAD_L2_BOUND = {"img_encoder": 0.050, "pc_encoder": 0.069, "pc_decoder": 0.034}
DGCNN32_GRAD_L2_BOUND = {"img_encoder": 0.048, "pc_encoder": 0.017, "pc_decoder": 0.016}


def test_evaluation_dict_with_emd(gpu, oracle):
    """evaluate_Network's per-item values: Chamfer and the Sinkhorn-form EMD."""
    from fpsg_amd.engine import build_model, default_options
    from fpsg_amd.episodes import synthetic_episode
    torch.manual_seed(2)
    S, Q = 2, 1
    cpu = build_model(default_options(device="cpu")).eval()
    dev = copy.deepcopy(cpu).to(gpu)
    cpu.pc_metric = oracle.make_torch_chamfer()
    cpu.emd_metric = lambda a, b: torch.from_numpy(
        oracle.sinkhorn_divergence(a.detach().numpy(), b.detach().numpy())).sum()   # emd_loss(sinkhorn=True)
    ep = synthetic_episode(S, Q, n_pts=2048, img_size=96, seed=4)
    ep_gpu = {k: (v.to(gpu) if torch.is_tensor(v) else v) for k, v in ep.items()}
    grids_cpu = _fixed_grids(cpu, (Q,), "cpu")
    _pin_grids(cpu, grids_cpu)
    _pin_grids(dev, {b: [[t.to(gpu) for t in c] for c in g] for b, g in grids_cpu.items()})
    with torch.no_grad():
        a, b = cpu._return_reconstruction(ep), dev._return_reconstruction(ep_gpu)
    # north_star's 1e-4 (the bounds were 1e-3 / 5e-3 until round 5: eval-mode episodes measure ~1e-7)
    assert abs(float(a["cd_loss"]) - float(b["cd_loss"])) <= 1e-4 * abs(float(a["cd_loss"]))
    assert abs(float(a["emd_loss"]) - float(b["emd_loss"])) <= 1e-4 * abs(float(a["emd_loss"]))


def test_evaluation_dict_at_config2_size(gpu, oracle):
    """``_return_reconstruction`` (few_shot.py:131-176, what evaluate_Network.py:107-118 sums per
    item) ONCE at configs[2] size -- 32-shot, 5-query, 224 x 224 images, eval mode as the evaluation loop runs it --
    against the fp32 CPU port with the C oracle's Chamfer and the CPU restatement of the Sinkhorn divergence.
    Bounds: north_star's 1e-4 on ``cd_loss``; ``emd_loss`` at 3x the deviation measured on MI355X (<= 1e-4)."""
    from fpsg_amd.engine import build_model, default_options
    from fpsg_amd.episodes import synthetic_episode
    torch.manual_seed(12)
    S, Q = 32, 5
    cpu = build_model(default_options(device="cpu")).eval()
    dev = copy.deepcopy(cpu).to(gpu)
    cpu.pc_metric = oracle.make_torch_chamfer()
    cpu.emd_metric = lambda a, b: torch.from_numpy(
        oracle.sinkhorn_divergence(a.detach().numpy(), b.detach().numpy())).sum()   # emd_loss(sinkhorn=True)
    ep = synthetic_episode(S, Q, n_pts=2048, img_size=224, seed=14)
    ep_gpu = {k: (v.to(gpu) if torch.is_tensor(v) else v) for k, v in ep.items()}
    grids_cpu = _fixed_grids(cpu, (Q,), "cpu")
    _pin_grids(cpu, grids_cpu)
    _pin_grids(dev, {b: [[t.to(gpu) for t in c] for c in g] for b, g in grids_cpu.items()})
    with torch.no_grad():
        a, b = cpu._return_reconstruction(ep), dev._return_reconstruction(ep_gpu)
    measured = {"S": S, "Q": Q, "mode": "eval", "image": 224}
    for key in ("cd_loss", "emd_loss"):
        measured[key] = abs(float(a[key]) - float(b[key])) / abs(float(a[key]))
    _record("evaluation_dict_config2_size", measured)
    assert measured["cd_loss"] <= 1e-6, measured       # north_star: 1e-4; measured 9.4e-8
    assert measured["emd_loss"] <= EVAL_EMD_BOUND, measured


EVAL_EMD_BOUND = 1e-6      # measured on MI355X: cd_loss 9.4e-8, emd_loss 0.0 (profiles/r05/episode_parity_deviation.jsonl); floor 1e-6


def test_dgcnn_encoder_forward_vs_oracle_graph_ops(gpu, oracle):
    """DGCNN encoder (HIP kNN + fused EdgeConv) against a CPU forward assembled from the oracle's
    kNN / edge features and PyTorch-CPU layers with the same weights (eval-mode BatchNorm)."""
    from fpsg_amd.dgcnn import DGCNNfeat
    torch.manual_seed(5)
    net = DGCNNfeat().eval()
    with torch.no_grad():
        for mod in net.modules():
            if isinstance(mod, (torch.nn.BatchNorm1d, torch.nn.BatchNorm2d)):
                mod.weight.copy_(torch.randn_like(mod.weight) * 0.6)      # both signs: max and min paths
                mod.running_mean.copy_(torch.randn_like(mod.running_mean) * 0.1)
                mod.running_var.copy_(torch.rand_like(mod.running_var) + 0.5)
    g = torch.Generator().manual_seed(6)
    x = torch.randn(2, 3, 300, generator=g)
    x = x / x.norm(dim=1, keepdim=True).amax(dim=2, keepdim=True)

    def cpu_forward(x):
        feats = []
        h = x
        for block in (net.conv1, net.conv2, net.conv3, net.conv4):
            idx = oracle.knn(h.numpy(), 20)
            edge = torch.from_numpy(oracle.edge_feature(h.numpy(), idx))
            h = block(edge).max(dim=-1)[0]
            feats.append(h)
        z = net.conv5(torch.cat(feats, dim=1))
        return torch.cat((z.max(dim=2)[0], z.mean(dim=2)), dim=1)

    with torch.no_grad():
        ref = cpu_forward(x)
        got = copy.deepcopy(net).to(gpu)(x.to(gpu)).cpu()
    close = torch.isclose(got, ref, rtol=2e-3, atol=2e-4)
    assert close.float().mean() > 0.995, float(close.float().mean())   # a near-tie neighbour swap moves few features


@pytest.mark.parametrize("S,Q", [(4, 2), (32, 5)])
def test_dgcnn_episode_losses(gpu, oracle, monkeypatch, S, Q):
    """configs[3]'s path end to end, at a small size and ONCE at configs[3]'s own (32-shot 5-query: 64 clouds of 2048
    points through four EdgeConv layers, 2.6 M edges each; the CPU port's oracle kNN takes about a minute): an intra_recon episode with the DGCNN encoder (HIP kNN + fused EdgeConv, training-mode
    BatchNorm over the edges) against a CPU port whose encoder is assembled from the oracle's edge features and
    PyTorch-CPU layers on the same weights (dgcnn/model.py:59-88: kNN on the CURRENT features, conv + BatchNorm2d +
    LeakyReLU on [x_j - x_i ; x_i], max over k; conv5, max | mean over the points), the rest of the model being the fp32
    CPU port of the PointNet test.  Forward only.
    A kNN graph is a discrete function of features that differ in the last bits between the two runs: a near-tie
    neighbour swap moves a few pooled features by 1e-3 and the losses with them (measured with free graphs: query loss
    4e-5, support loss 1.3e-3 -- the reference on another device has the same sensitivity).  So the arithmetic is
    compared on the SAME graphs -- the CPU port takes the neighbour lists the HIP run produced -- to north_star's 1e-4,
    and the graphs themselves are compared with the oracle's kNN on the CPU port's features: layer 1 (the raw
    clouds: identical inputs) must agree exactly, the deeper layers on all but a few near-tie edges."""
    import numpy as np
    from fpsg_amd import dgcnn as dg
    from fpsg_amd.engine import build_model, default_options
    from fpsg_amd.episodes import synthetic_episode
    torch.manual_seed(9)
    cpu = build_model(default_options(device="cpu", intra_recon=True, pc_encoder="dgcnn")).train()
    dev = copy.deepcopy(cpu).to(gpu)
    net = cpu.pc_encoder.pc_encoder                      # DGCNNfeat: its fused forward needs the GPU
    graphs = []                                          # the HIP run's neighbour lists, in call order
    real_knn = dg.knn_int32

    def recording_knn(x, k, point_major=False, flags=0):
        idx = real_knn(x, k, point_major=point_major, flags=flags)
        graphs.append(idx.cpu().numpy().astype(np.int64))
        return idx

    monkeypatch.setattr(dg, "knn_int32", recording_knn)
    agree = []

    full = S <= 4          # the small episode: every cloud's graph, the oracle's own edge features, forward only

    def edge_features(h, idx):
        """[x_j - x_i ; x_i] (dgcnn/model.py:23-42) with torch ops, so that the CPU port's backward reaches the features
        (the oracle's C function is not differentiable); bit-identical to ``oracle.edge_feature`` (checked below)."""
        hT = h.transpose(1, 2)                                            # [B, N, C]
        nb = hT[torch.arange(h.size(0))[:, None, None], torch.from_numpy(idx)]        # [B, N, k, C]
        ctr = hT[:, :, None, :].expand_as(nb)
        return torch.cat((nb - ctr, ctr), dim=3).permute(0, 3, 1, 2).contiguous()

    def cpu_forward(x):
        feats, h = [], x
        for li, block in enumerate((net.conv1, net.conv2, net.conv3, net.conv4)):
            idx = graphs.pop(0)
            # graph agreement: every cloud on the small episode and for layer 1 (cheap: 3 channels); at configs[3] size
            # the deeper layers on the first and last three clouds of the call (the oracle's scalar kNN on 64 clouds x
            # 64-128 channels was 3 of this test's 3.5 minutes)
            sel = np.arange(h.size(0)) if (full or li == 0 or h.size(0) <= 6) else np.r_[0:3, h.size(0) - 3:h.size(0)]
            own = oracle.knn(h.detach().numpy()[sel], net.k)
            agree.append(float((np.sort(own, -1) == np.sort(idx[sel], -1)).all(-1).mean()))   # points with the same neighbour SET
            edge = edge_features(h, idx)
            if full:
                assert np.array_equal(edge.detach().numpy(), oracle.edge_feature(h.detach().numpy(), idx))
            h = block(edge).max(dim=-1)[0]
            feats.append(h)
        z = net.conv5(torch.cat(feats, dim=1))
        return torch.cat((z.max(dim=2)[0], z.mean(dim=2)), dim=1) if net.dual_flag else z.max(dim=2)[0]

    net.forward = cpu_forward
    cpu.pc_metric = oracle.make_torch_chamfer()
    ep = synthetic_episode(S, Q, n_pts=2048, img_size=224, seed=21)
    ep_gpu = {k: (v.to(gpu) if torch.is_tensor(v) else v) for k, v in ep.items()}
    grids_cpu = _fixed_grids(cpu, (S, Q), "cpu")
    _pin_grids(cpu, grids_cpu)
    _pin_grids(dev, {b: [[t.to(gpu) for t in c] for c in g] for b, g in grids_cpu.items()})
    with torch.set_grad_enabled(not full):
        out_g = dev.loss(ep_gpu)                         # first: fills `graphs` (one list per layer and encoder call)
        out_c = cpu.loss(ep)
    assert not graphs
    measured = {"mode": "train", "encoder": "dgcnn", "S": S, "Q": Q, "same_neighbour_sets": agree}
    dist = None
    if not full:
        # one backward of both at configs[3] size, per-group relative L2 distance of the gradients
        out_c["ttl_loss"].sum().backward()
        out_g["ttl_loss"].sum().backward()
        dist = _group_grad_distances(dev, cpu)
        for top, v in dist.items():
            measured[f"grad_l2[{top}]_hip_vs_cpu32"] = v
    for key in ("query_rec_loss", "support_rec_loss", "ttl_loss"):
        a, b = float(out_c[key].detach().sum()), float(out_g[key].detach().sum())
        measured[key] = abs(a - b) / abs(a)
    # the encoders' running statistics after the training-mode forward agree too
    for (n1, b1), (_, b2) in zip(net.named_buffers(), dev.pc_encoder.pc_encoder.named_buffers()):
        if "running" in n1:
            measured[f"buf[{n1}]"] = float((b1 - b2.cpu()).abs().max() / (b1.abs().max() + 1e-12))
    _record("episode_parity_dgcnn", measured)
    for key in ("query_rec_loss", "support_rec_loss", "ttl_loss"):
        assert measured[key] <= 1e-4, measured
    assert max(v for k, v in measured.items() if k.startswith("buf[")) <= 1e-3, measured
    n_layers = 4
    assert all(a == 1.0 for a in agree[0::n_layers]), agree          # layer 1 of every encoder call: identical inputs
    assert min(agree) >= 0.98, agree
    if dist is not None:
        for top, bound in DGCNN32_GRAD_L2_BOUND.items():
            assert dist[top] <= bound, (top, dist)
