"""Model construction and the optimizer step shared by the entry points and ``bench.py``.

``build_model`` follows reference ``src/trainNetwork.py:45-64`` (same ``opt`` namespace: the
argparse result doubles as the decoder's config).  ``TrainStep`` is the hot loop body of
``trainNetwork.py:140-148`` (zero_grad -> loss -> backward -> optimizer.step) extended to
``E`` episodes per optimizer step and ``W`` ranks (``fpsg_amd.dist``); with ``E = W = 1`` it
is exactly the reference step.
"""
from __future__ import annotations

import argparse
import os

import torch
import torch.optim as optim

from . import dist as fdist
from .few_shot import ImgPCProtoNet
from .image_net import ImageEncoderWarpper
from .point_cloud_net import PCDecoder, PCEncoder


def default_options(**overrides) -> argparse.Namespace:
    """The reference's argparse defaults (``trainNetwork.py:215-261``) for everything the
    model constructors read."""
    opt = argparse.Namespace(
        img_encoder="vgg_16", pc_encoder="pointnet", pc_encoder_path="", img_encoder_path="",
        support_factor=1.0, query_factor=1.0, intra_recon=False, num_clusters=4, ori_dim=2,
        raw_dim=3, num_nodes=4, device="cuda", bottleneck_size=1536, template_type="SQUARE",
        activation="relu", aggregate="single", pc_dist="cd", lr=1e-3, lr_decay=350, SGD=False,
        n_way=1, n_shot=20, n_query=0)
    for k, v in overrides.items():
        setattr(opt, k, v)
    return opt


def build_model(opt) -> ImgPCProtoNet:
    img_encoder = ImageEncoderWarpper(opt.img_encoder, finetune_layer=3,
                                      weights=getattr(opt, "img_encoder_path", "") or None)
    pc_encoder = PCEncoder(opt.pc_encoder)
    pc_decoder = PCDecoder(conf=opt)
    path = getattr(opt, "pc_encoder_path", "")
    if path and os.path.exists(path):
        print("Pretrained Model exist, loading")
        pc_encoder.load_state_dict(torch.load(path, map_location="cpu", weights_only=True))
    return ImgPCProtoNet(img_encoder, pc_encoder, pc_decoder, mask_learner=None,
                         query_factor=opt.query_factor, support_factor=opt.support_factor,
                         metric=getattr(opt, "pc_dist", "cd"), intra_support=opt.intra_recon,
                         aggregate=opt.aggregate)


def build_optimizer(model, opt):
    """Adam(lr, betas=(.9,.999)) or SGD(weight_decay=1e-2) + StepLR(gamma=.5)
    (``trainNetwork.py:118-130``)."""
    on_gpu = next(model.parameters()).is_cuda
    if not opt.SGD:
        optimizer = optim.Adam(model.parameters(), lr=opt.lr, betas=(0.9, 0.999),
                               **({"fused": True} if on_gpu else {}))
    else:
        optimizer = optim.SGD(model.parameters(), lr=opt.lr, weight_decay=1e-2)
    scheduler = optim.lr_scheduler.StepLR(optimizer, step_size=int(opt.lr_decay), gamma=0.5)
    return optimizer, scheduler


def to_device(sample: dict, device) -> dict:
    """``trainNetwork.py:37-43`` ``to_cuda``: moves the six tensors of an episode."""
    for k in ("xs", "xq", "xad", "pcs", "pcq", "pcad"):
        if k in sample and sample[k].device != device:
            sample[k] = sample[k].to(device, non_blocking=True)
    return sample


class TrainStep:
    """One optimizer step over the local share of a global step's episodes."""

    def __init__(self, model, optimizer, world: int = 1, bucket_mb: float = 80.0):
        self.model, self.optimizer, self.world = model, optimizer, world
        self.buckets = fdist.FlatGradBuckets(model, bucket_mb=bucket_mb)

    def __call__(self, local_episodes: list[dict], n_episodes_global: int | None = None):
        """Returns the list of loss dicts (device tensors; call ``.item()`` outside the
        timed path).  ``n_episodes_global`` defaults to ``len(local) * world``."""
        if n_episodes_global is None:
            n_episodes_global = len(local_episodes) * self.world
        self.buckets.zero()
        results = []
        if not local_episodes:  # still take part in the step's collectives
            self.buckets.arm()
        last = len(local_episodes) - 1
        for k, sample in enumerate(local_episodes):
            out = self.model.loss(sample)
            if k == last:
                self.buckets.arm()
            out["ttl_loss"].sum().backward()
            results.append({n: v.detach() for n, v in out.items()})
        self.buckets.finish(n_episodes_global)
        self.optimizer.step()
        return results
