"""Entry-point plumbing that needs no GPU: parsers build (the reference's raise on a typo),
file-backed datasets read the reference's list / PLY / NPY layouts, and BASELINE configs[0]
(point auto-encoder + Chamfer, CPU) steps when driven through the oracle."""
import os

import numpy as np
import pytest
import torch
from PIL import Image

from fpsg_amd import cli


def test_parsers_build_and_keep_reference_defaults():
    opt = cli.few_shot_parser().parse_args(["--synthetic"])
    assert (opt.n_shot, opt.n_query, opt.n_episode, opt.epoch, opt.lr, opt.lr_decay) == (20, 0, 100, 500, 1e-3, 350)
    assert (opt.num_clusters, opt.num_nodes, opt.bottleneck_size, opt.template_type) == (4, 4, 1536, "SQUARE")
    assert opt.sequential_eval is False and opt.pc_dist == "cd" and opt.aggregate == "single"
    ev = cli.few_shot_parser(evaluation=True).parse_args(["--synthetic", "--sequential_eval", "--eval_model", "m.pt"])
    assert ev.sequential_eval is True and ev.eval_model == "m.pt"
    with pytest.raises(SystemExit):
        cli.validate(cli.few_shot_parser().parse_args([]))          # needs paths or --synthetic


def _write_ply(path, pts):
    with open(path, "w") as f:
        f.write("ply\nformat ascii 1.0\nelement vertex %d\nproperty float x\nproperty float y\nproperty float z\nend_header\n" % len(pts))
        for p in pts:
            f.write("%f %f %f\n" % tuple(p))


def test_modelnet_file_layout(tmp_path):
    from fpsg_amd.datasets import FewShotModelNet, ImageTransform, ply_reader
    rng = np.random.default_rng(0)
    aux = tmp_path / "aux"; aux.mkdir()
    lines = []
    for cls in ("cup", "door"):
        cl = []
        for i in range(5):
            d = tmp_path / "img" / cls / "train" / f"{cls}_{i}"
            d.mkdir(parents=True)
            img = d / "v0.png"
            Image.fromarray(rng.integers(0, 255, (600, 600, 3), dtype=np.uint8)).save(img)
            ply = tmp_path / f"{cls}_{i}.ply"
            _write_ply(ply, rng.standard_normal((40 + i, 3)) * 3 + 1)
            cl.append(f"{img}\t{ply}")
        (aux / f"modelnet+{cls}.txt").write_text("\n".join(cl))
        lines += cl
    cfg = tmp_path / "modelnet_train.txt"
    cfg.write_text("\n".join(lines))
    assert len(ply_reader(str(tmp_path / "cup_0.ply"))) == 40
    ds = FewShotModelNet(str(cfg), str(aux), n_classes=1, n_support=2, n_query=1, transform=ImageTransform(550))
    assert len(ds) == 10 and ds.img_corpus.shape == (10, 3, 224, 224) and ds.pc_corpus.shape == (10, 2048, 3)
    ep = ds[7]
    assert ep["class"] == "door" and ep["xs"].shape == (2, 3, 224, 224) and ep["pcq"].shape == (1, 2048, 3)
    assert ep["xad"].shape == (2, 3, 224, 224) and ep["pcad"].shape == (2, 2048, 3)
    pc = ds.pc_corpus
    assert torch.allclose(pc.norm(dim=-1).amax(dim=-1), torch.ones(10), atol=1e-5)
    assert pc.mean(dim=1).abs().max() < 1e-5 and -1 <= ds.img_corpus.min() and ds.img_corpus.max() <= 1


def test_shapenet_file_layout(tmp_path):
    from fpsg_amd.datasets import FewShotShapeNet, shapenet_transform
    rng = np.random.default_rng(1)
    aux = tmp_path / "aux"; aux.mkdir()
    lines = []
    for syn in ("02880940", "03797390"):
        cl = []
        for i in range(4):
            item = tmp_path / "ShapeNet" / syn / f"m{i}"
            (item / "images").mkdir(parents=True)
            Image.fromarray(rng.integers(0, 255, (256, 256, 3), dtype=np.uint8)).save(item / "images" / "00.png")
            np.save(item / "npy_file.npy", rng.standard_normal((15000, 3)).astype(np.float32))
            cl.append(str(item))
        (aux / f"shapenet+{syn}.txt").write_text("\n".join(cl))
        lines += cl
    cfg = tmp_path / "shapenet_test.txt"
    cfg.write_text("\n".join(lines))
    ds = FewShotShapeNet(str(cfg), str(aux), n_classes=1, n_support=1, n_query=2, transform=shapenet_transform())
    ep = ds[5]
    assert ep["class"] == "mug" and ep["xq"].shape == (2, 3, 224, 224) and ep["pcs"].shape == (1, 2048, 3)


def test_config1_point_autoencoder_cpu(oracle):
    """BASELINE.json configs[0]: PointNet encoder -> decoder(1024 pts) -> Chamfer on CPU."""
    import trainPointAE
    from fpsg_amd.episodes import synthetic_clouds
    torch.manual_seed(0)
    cd = oracle.make_torch_chamfer()
    model = trainPointAE.PointAutoEncoder("pointnet", num_pts=1024, device="cpu").train()
    optim = torch.optim.Adam(model.parameters(), lr=1e-3)
    data = synthetic_clouds(8, 1024, torch.Generator().manual_seed(1))     # [8,1024,3] (32 in the full config)
    losses = []
    for _ in range(3):
        optim.zero_grad()
        rec = model(data)
        assert rec.shape == (8, 1024, 3)
        loss = cd(rec, data).mean()
        loss.backward()
        optim.step()
        losses.append(float(loss.detach()))
    assert np.isfinite(losses).all() and losses[0] > 0
