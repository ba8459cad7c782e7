"""The three entry points run end to end on the GPU with synthetic data (HIP losses)."""
import os
import subprocess
import sys

import pytest

from conftest import GOLDEN, ROOT

pytestmark = pytest.mark.gpu


def _run(args, cwd=ROOT, timeout=600):
    env = dict(os.environ, PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable] + args, cwd=cwd, env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    return r.stdout


def test_train_then_evaluate(gpu, tmp_path):
    ck = str(tmp_path)
    out = _run(["trainNetwork.py", "--synthetic", "--resident", "--n_shot", "2", "--n_query", "1", "--intra_recon",
                "--epoch", "2", "--n_episode", "3", "--eval_interval", "2", "--save_interval", "2",
                "--sample_interval", "2", "--model_path", ck, "--name", "t",
                "--pc_encoder_path", os.path.join(GOLDEN, "pretrained_pcencoder_pointnet.pt")])
    assert "Training Results for Epoch -- 2 are: Query_rec:" in out and "Pretrained Model exist, loading" in out
    assert "Avg testing results across all classes Epoch -- 2" in out and "Class: class00 -- Rec CD:" in out
    files = os.listdir(os.path.join(ck, "t"))
    assert "model_epoch_2.pt" in files and any(f.startswith("log_") for f in files)
    imgs = os.listdir(os.path.join(ck, "t", "images"))
    assert "sample_img_2.png" in imgs and any(f.endswith("_gt.npy") for f in imgs)       # F10 fixed
    ev = _run(["evaluate_Network.py", "--synthetic", "--n_shot", "2", "--n_query", "1", "--sequential_eval",
               "--model_path", ck, "--name", "t", "--eval_model", "model_epoch_2.pt"])
    assert "Class: class00 -- Rec CD:" in ev and "Rec EMD:" in ev
    # resume from the saved weights, EMD as the training metric (dead in the reference, F3)
    out2 = _run(["trainNetwork.py", "--synthetic", "--n_shot", "1", "--n_query", "1", "--epoch", "3", "--resume", "2",
                 "--n_episode", "2", "--pc_dist", "emd", "--model_path", ck, "--name", "t"])
    assert "Resume previous training, start from epoch 2" in out2 and "Epoch -- 3" in out2
    assert "train_state_epoch_2.pt" in files and "optimizer / scheduler state restored" in out2
    assert "the next epoch is 3" in out2


def test_dgcnn_encoder_and_ae_mode(gpu, tmp_path):
    out = _run(["trainNetwork.py", "--synthetic", "--pc_encoder", "dgcnn", "--n_shot", "2", "--n_query", "1",
                "--epoch", "1", "--n_episode", "2", "--eval_interval", "9", "--sample_interval", "9",
                "--model_path", str(tmp_path), "--name", "d"])
    assert "Training Results for Epoch -- 1" in out
    ae = _run(["trainPointAE.py", "--ae", "--epoch", "2", "--batch_size", "8"])
    assert "AE epoch 2: Chamfer" in ae
    pre = _run(["trainPointAE.py", "--synthetic", "--core", "dgcnn", "--epoch", "1", "--n_pts", "256",
                "--batch_size", "16", "--model_path", str(tmp_path), "--name", "pre"])
    assert os.path.exists(os.path.join(str(tmp_path), "pre", "pre_dgcnn.pt")) and "Running CrossEntropy" in pre


def test_gemm_tuning_records_load_and_keep_results(gpu):
    """fpsg_amd.gemm_tuning: the committed records file loads (validators match this image) and a
    GEMM of a recorded shape still equals the float64 product; disable() restores the default."""
    import os
    import torch
    from fpsg_amd import gemm_tuning
    assert os.path.exists(gemm_tuning.DEFAULT_FILE)
    info = gemm_tuning.enable()
    try:
        assert info["gemm_tuning"] == "file" and info["gemm_records_loaded"] is True
        import torch.cuda.tunable as tunable
        assert tunable.is_enabled() and not tunable.tuning_is_enabled()
        assert len(tunable.get_results()) > 50
        torch.manual_seed(0)
        U = torch.randn(36, 512, 512, device=gpu)
        V = torch.randn(36, 512, 1813, device=gpu)          # conv4_2 of the c5 workload, F(4x4,3x3)
        M = torch.bmm(U, V)
        ref = torch.bmm(U[:2].double(), V[:2].double())
        assert float((M[:2].double() - ref).abs().max()) <= 1e-4 * float(ref.abs().max())
    finally:
        gemm_tuning.disable()
    import torch.cuda.tunable as tunable
    assert not tunable.is_enabled()


def test_evaluation_item_replayed_as_a_graph_equals_the_plain_method(gpu, monkeypatch):
    """``engine.EvalItem`` (the body of evaluate_Network's loop: everything in front of the EMD captured once per input
    shape and replayed as a hipGraph, BatchNorm coefficients and transformed filters made once per block) against
    ``ImgPCProtoNet._return_reconstruction`` in the reference's full form (``FPSG_EVAL_PRUNE=0``: the ad images and ad
    clouds, whose features evaluation drops unused, encoded too) on the same items with the patch grids pinned: ``cd_loss`` and
    ``emd_loss`` within 1e-6, on items the graph was NOT captured on; no coefficient kernel runs
    for a BatchNorm the block has already seen."""
    import torch
    from fpsg_amd import winograd
    from fpsg_amd.engine import EvalItem, build_model, default_options
    from fpsg_amd.episodes import synthetic_episode
    torch.manual_seed(3)
    model = build_model(default_options(device="cuda")).to(gpu).eval()
    S, Q = 4, 2
    grids = model.pc_decoder.sample_grids(Q, gpu, torch.Generator(device=gpu).manual_seed(9))
    orig = model.pc_decoder.forward
    model.pc_decoder.forward = lambda h, grid=None, generator=None, pack=None: orig(h, grid=grids, pack=pack)
    eps = [synthetic_episode(S, Q, n_pts=2048, img_size=96, seed=40 + i, device=gpu) for i in range(5)]
    monkeypatch.setenv("FPSG_EVAL_CHAN_CACHE", "0")
    monkeypatch.setenv("FPSG_EVAL_PRUNE", "0")             # the reference's full forward (ad images and ad clouds encoded too)
    with torch.no_grad():
        plain = [model._return_reconstruction(ep) for ep in eps]
    monkeypatch.delenv("FPSG_EVAL_CHAN_CACHE")
    monkeypatch.delenv("FPSG_EVAL_PRUNE")
    with EvalItem(model) as item:
        got = [item(ep) for ep in eps]                     # two eager items, then the capture and replays
        assert item._graphs, "the third item of a shape must have been captured"
        cache = winograd.frozen_cache_ro()
        assert any(isinstance(k, tuple) and k and k[0] == "bn_chan" for k in cache), "coefficients cached for the block"
    for a, b in zip(plain, got):
        for key in ("cd_loss", "emd_loss"):
            x, y = float(a[key]), float(b[key])
            assert abs(x - y) <= 1e-6 * abs(x), (key, x, y)
    # a model whose metrics were replaced takes the plain method (tests drive it with the oracle's functions)
    model.emd_metric = lambda p, q: (p - q).abs().sum()
    with EvalItem(model) as item:
        for ep in eps[:4]:
            item(ep)
        assert not item._graphs
