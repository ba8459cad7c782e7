"""Argument parsers and data plumbing shared by the three entry points.

The flag set, defaults and help semantics are those of reference
``src/trainNetwork.py:212-261`` / ``src/evaluate_Network.py:129-178`` (including the flags the
reference parses but never reads, SURVEY.md F13, so that existing command lines keep
working).  ``--sequential_eval`` is declared with ``store_true``: the reference's
``action='store_ture'`` typo makes argparse raise while the parser is being built
(SURVEY.md F4).

Additions (all optional): ``--synthetic`` (file-free corpora of the reference's shapes),
``--episodes_per_step`` (episodes per optimizer step across all ranks; data-parallel under
``torchrun``), ``--img_encoder_path`` (local VGG16-BN weights; nothing is downloaded),
``--resident`` (keep the corpora in HBM, assemble episodes on the device).
"""
from __future__ import annotations

import argparse

import os

import torch
from torch.utils.data import DataLoader

from .episodes import EpisodicBatchSampler, SequentialBatchSampler, SyntheticFewShot


def few_shot_parser(evaluation: bool = False) -> argparse.ArgumentParser:
    p = argparse.ArgumentParser()
    g = p.add_argument_group("data / episode")
    g.add_argument("--synthetic", action="store_true", help="Use synthetic corpora (no files needed);")
    g.add_argument("--config_path", type=str, default="", help="Path to the configuration file: {DATASET}_{SPLIT}.txt;")
    g.add_argument("--test_path", type=str, default="", help="Path to the test file: {DATASET}_{SPLIT}.txt;")
    g.add_argument("--refer_path", type=str, default="./modelnet_files/", help="Path to the reference folder [default: ./modelnet_files/];")
    g.add_argument("--dataset", type=str, default="modelnet", choices=["modelnet", "shapenet"])
    g.add_argument("--pc_encoder_path", type=str, default="", help="Path to the pre-trained pcencoder;")
    g.add_argument("--img_encoder_path", type=str, default="", help="Local VGG16-BN state dict (optional);")
    g.add_argument("--n_way", type=int, default=1)
    g.add_argument("--n_shot", type=int, default=20)
    g.add_argument("--n_query", type=int, default=0, help="Number of Query set [default: --n_shot];")
    g.add_argument("--resident", action="store_true", help="Keep corpora on the GPU (synthetic mode);")

    g = p.add_argument_group("network")
    g.add_argument("--img_encoder", type=str, default="vgg_16")
    g.add_argument("--pc_encoder", type=str, default="pointnet")
    g.add_argument("--support_factor", type=float, default=1.0)
    g.add_argument("--query_factor", type=float, default=1.0)
    g.add_argument("--intra_recon", action="store_true")
    g.add_argument("--epoch_start_recon", type=int, default=0)
    g.add_argument("--num_clusters", type=int, default=4)
    g.add_argument("--ori_dim", type=int, default=2)
    g.add_argument("--raw_dim", type=int, default=3)
    g.add_argument("--num_nodes", type=int, default=4)
    g.add_argument("--device", type=str, default="cuda")
    g.add_argument("--bottleneck_size", type=int, default=1536)
    g.add_argument("--template_type", type=str, default="SQUARE")
    g.add_argument("--activation", type=str, default="relu")
    g.add_argument("--dim_template", type=int, default=2)
    g.add_argument("--aggregate", type=str, default="single", choices=["single", "multi", "mask_single", "mask_multi"])

    g = p.add_argument_group("training")
    g.add_argument("--n_episode", type=int, default=100)
    g.add_argument("--epoch", type=int, default=500)
    g.add_argument("--lr", type=float, default=1e-3)
    g.add_argument("--lr_decay", type=float, default=350)
    g.add_argument("--resume", type=int, default=-1)
    g.add_argument("--pc_dist", type=str, default="cd", choices=["cd", "emd"])
    g.add_argument("--SGD", action="store_true")
    g.add_argument("--episodes_per_step", type=int, default=0,
                   help="Episodes per optimizer step over all ranks [default: one per rank];")

    g = p.add_argument_group("experiment")
    g.add_argument("--name", type=str, default="0")
    g.add_argument("--dir_name", type=str, default="")
    g.add_argument("--model_path", type=str, default="../checkpoint")
    g.add_argument("--save_interval", type=int, default=50)
    g.add_argument("--sample_interval", type=int, default=10)
    g.add_argument("--eval_interval", type=int, default=20)
    g.add_argument("--eval_model", type=str, default="NONE")
    g.add_argument("--sequential_eval", action="store_true")
    if evaluation:
        g.add_argument("--npy_folder", type=str, default="", help="Where draw_reconstruction dumps go;")
    return p


def validate(opt) -> None:
    if not opt.synthetic and not (opt.config_path and opt.test_path):
        raise SystemExit("--config_path and --test_path are required unless --synthetic is given")
    if opt.n_way != 1:
        raise SystemExit("only 1-way episodes are defined by the model (as in the reference)")


def build_datasets(opt, n_query: int, device):
    """(train, test) datasets with the reference's item layout."""
    if opt.synthetic:
        where = device if opt.resident else "cpu"
        need = opt.n_shot + max(n_query, 1)
        ds = SyntheticFewShot(n_classes=4, per_class=max(need, 8), n_support=opt.n_shot,
                              n_query=n_query, seed=1234, device=where)
        ds_test = SyntheticFewShot(n_classes=2, per_class=max(need, 8), n_support=opt.n_shot,
                                   n_query=n_query, seed=4321, device=where)
        return ds, ds_test
    from .datasets import FewShotModelNet, FewShotShapeNet, modelnet_transform, shapenet_transform
    if opt.dataset == "modelnet":
        cls, tfs = FewShotModelNet, modelnet_transform()
    else:
        cls, tfs = FewShotShapeNet, shapenet_transform()
    ds = cls(opt.config_path, opt.refer_path, n_classes=opt.n_way, n_support=opt.n_shot,
             n_query=n_query, transform=tfs)
    ds_test = cls(opt.test_path, opt.refer_path, n_classes=opt.n_way, n_support=opt.n_shot,
                  n_query=n_query, transform=tfs)
    return ds, ds_test


def _collate(batch):
    """One episode per batch (n_way = 1): add the leading axis DataLoader's default collate
    would add, without copying device-resident tensors through the CPU."""
    from .episodes import collate_episode
    assert len(batch) == 1
    return collate_episode(batch[0])


def build_loaders(opt, ds, ds_test):
    sampler = EpisodicBatchSampler(len(ds), opt.n_way, opt.n_episode)
    ran_sampler = EpisodicBatchSampler(len(ds_test), opt.n_way, opt.n_episode)
    seq_sampler = SequentialBatchSampler(len(ds_test))
    dl = DataLoader(ds, batch_sampler=sampler, num_workers=0, collate_fn=_collate)
    dl_test = DataLoader(ds_test, batch_sampler=seq_sampler if opt.sequential_eval else ran_sampler,
                         num_workers=0, collate_fn=_collate)
    return dl, dl_test


def usable_cores() -> int:
    """Cores this process may really use: affinity mask and cgroup CPU quota, not the host's core count (a container
    on a GPU node sees every host core but is granted a share of them).  ``FPSG_CPU_THREADS`` caps it (default 16)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except AttributeError:
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, int(os.environ.get("FPSG_CPU_THREADS", "16"))))


def limit_host_threads(world: int = 1) -> int:
    """PyTorch sizes its intra-op pool by the HOST's core count; with a 16-core share of a 200-core node every host
    op that parallelises (the row gathers of episode assembly, normalisation, collate) then thrashes -- the gather
    of one 32-shot episode took 30 ms instead of 3.  One call per process, before the first parallel op."""
    n = max(1, usable_cores() // max(world, 1))
    if torch.get_num_threads() > n:
        torch.set_num_threads(n)
    return n


def pick_device(opt) -> torch.device:
    limit_host_threads()
    if opt.device.startswith("cuda") and not torch.cuda.is_available():
        raise SystemExit("--device cuda requested but no ROCm GPU is visible "
                         "(the Chamfer / EMD / kNN ops are HIP kernels; there is no CPU fallback)")
    device = torch.device(opt.device)
    if device.type == "cuda":
        from . import gemm_tuning
        gemm_tuning.enable()          # recorded library-GEMM kernel choices for this path's shapes
    return device
