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
from . import bn_counters, winograd
from .few_shot import ImgPCProtoNet
from .image_net import ImageEncoderWarpper
from .optim import FlatAdam
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
        if on_gpu and os.environ.get("FPSG_FLAT_ADAM", "1") != "0":
            # K7: parameters, gradients and moments in flat buffers, the step is one HBM stream
            optimizer = FlatAdam(model.parameters(), lr=opt.lr, betas=(0.9, 0.999))
        else:
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


def _total(loss):
    """The scalar to differentiate: the loss itself when it already is one (no reduction launch)."""
    return loss if loss.dim() == 0 else loss.sum()


class TrainStep:
    """One optimizer step over the local share of a global step's episodes.

    ``graph=True`` (GPU only): the forward + backward of an episode is captured once per
    episode shape into a hipGraph and replayed -- an episode is ~1300 kernel launches, many of
    them a few microseconds long, and the eager step is partly host-bound.  Inputs are copied
    into the graph's static buffers; the captured sequence ends with the multi-tensor adds of
    the episode's gradients into the flat buffer (``FlatGradBuckets.absorb``).  With more than
    one rank the LAST local episode of a step always runs eagerly, so that its backward can
    launch the bucketed all-reduce from the autograd hooks."""

    def __init__(self, model, optimizer, world: int = 1, bucket_mb: float = 80.0, graph: bool = False):
        self.model, self.optimizer, self.world = model, optimizer, world
        self.buckets = fdist.FlatGradBuckets(model, bucket_mb=bucket_mb)
        if isinstance(optimizer, FlatAdam):
            optimizer.bind_gradients(self.buckets.flat)      # same layout: the step reads it in place
        self.use_graph = bool(graph) and next(model.parameters()).is_cuda
        if self.use_graph:
            self.buckets.lazy = False       # a captured episode ends with its own adds into the flat buffer, in order
        self._graphs = {}
        self._eager_runs = {}

    # ------------------------------------------------------------------ graph plumbing
    _KEYS = ("xs", "xq", "xad", "pcs", "pcq", "pcad")

    def _shape_key(self, sample, first):
        return tuple((k, tuple(sample[k].shape)) for k in self._KEYS) + (bool(first),)

    def _episode(self, sample, first=False, absorb=True):
        """Forward + backward of one episode into fresh gradient tensors, then their
        multi-tensor copy (``first``) / add into the flat buffer (``absorb=False``: the step is
        this one episode and the optimizer reads the gradients where they are)."""
        self.buckets.detach()
        with bn_counters.deferred():        # the ~20 num_batches_tracked increments as multi-tensor adds
            out = self.model.loss(sample)
        _total(out["ttl_loss"]).backward()
        if absorb:
            self.buckets.absorb(first)
        return {n: v.detach() for n, v in out.items()}

    def _episode_hooked(self, sample, first):
        """The overlapped form (last local episode of a multi-rank step): fresh gradient tensors
        as in ``_episode``; the armed hooks absorb each bucket and launch its all-reduce as soon
        as its gradients are complete."""
        self.buckets.detach()
        self.buckets.arm(first)
        with bn_counters.deferred():
            out = self.model.loss(sample)
        _total(out["ttl_loss"]).backward()
        return {n: v.detach() for n, v in out.items()}

    def _capture(self, sample, first, absorb):
        static = {k: sample[k].clone() for k in self._KEYS}
        g = torch.cuda.CUDAGraph()
        # thread_local: an RCCL watchdog thread may query events while this thread captures
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            static_out = self._episode(static, first, absorb)
        # the captured gradient tensors belong to the graph's pool: kept (static addresses) when the
        # optimizer reads them in place, dropped otherwise
        grads = None if absorb else [p.grad for p in self.buckets.params]
        self.buckets.detach()
        return g, static, static_out, grads

    def _run_graphed(self, sample, first, absorb=True):
        key = self._shape_key(sample, first) + (bool(absorb),)
        if key not in self._graphs:
            # the first eager runs of a shape let MIOpen / hipBLASLt pick their kernels and
            # warm the allocator; capture happens on the third use
            n = self._eager_runs.get(key, 0)
            if n < 2:
                self._eager_runs[key] = n + 1
                return self._episode(sample, first, absorb)
            self._graphs[key] = self._capture(sample, first, absorb)   # records only; replayed below
        g, static, static_out, grads = self._graphs[key]
        if grads is not None:
            for p, gr in zip(self.buckets.params, grads):
                p.grad = gr
        for k in self._KEYS:
            static[k].copy_(sample[k], non_blocking=True)
        winograd.check_bank_before_replay()
        g.replay()
        return {n: v.clone() for n, v in static_out.items()}

    # ----------------------------------------------------------------------- the step
    def __call__(self, local_episodes: list[dict], n_episodes_global: int | None = None):
        """Returns the list of loss dicts (device tensors; call ``.item()`` outside the
        timed path).  ``n_episodes_global`` defaults to ``len(local) * world``."""
        if n_episodes_global is None:
            n_episodes_global = len(local_episodes) * self.world
        results = []
        multi = self.buckets.world_initialised()
        # one episode, one rank, K7: no flat gradient buffer at all -- the optimizer reads the
        # gradients through a pointer table (saves the 310 MB gather)
        direct = (isinstance(self.optimizer, FlatAdam) and len(local_episodes) == 1 and n_episodes_global == 1
                  and not multi and os.environ.get("FPSG_DIRECT_GRADS", "1") != "0")
        if not local_episodes:  # still take part in the step's collectives
            self.buckets.zero()
            self.buckets.attach()
            self.buckets.arm()
        last = len(local_episodes) - 1
        with winograd.weights_frozen():      # no parameter changes until optimizer.step() below
            for k, sample in enumerate(local_episodes):
                first, final = k == 0, k == last
                if final and multi:
                    results.append(self._episode_hooked(sample, first))
                elif self.use_graph:
                    results.append(self._run_graphed(sample, first, absorb=not direct))
                else:
                    results.append(self._episode(sample, first, absorb=not direct))
        if not direct:
            # K7 multiplies the gradient by 1/E as it reads it (bit-identical to scaling the buffer first: one fp32
            # product per element either way); the flat buffer then holds the SUM over the step's episodes.
            # FPSG_FOLD_GRAD_SCALE=0: the separate pass (A/B).
            fold = isinstance(self.optimizer, FlatAdam) and os.environ.get("FPSG_FOLD_GRAD_SCALE", "1") != "0"
            self.buckets.finish(n_episodes_global, scale=not fold)
            self.buckets.attach()       # the optimizer reads the step's gradient from the flat buffer
            if fold:
                self.optimizer.grad_scale = 1.0 / n_episodes_global if n_episodes_global > 1 else 1.0
        try:
            self.optimizer.step()
        finally:
            if isinstance(self.optimizer, FlatAdam):
                self.optimizer.grad_scale = 1.0
        return results


class EvalItem:
    """The body of the evaluation loop (reference ``src/evaluate_Network.py:107-118``: ``_return_reconstruction`` per test
    item, then two ``.item()`` reads) for a model in eval mode on a ROCm device.

        with EvalItem(model) as item:
            for sample in loader:
                out = item(sample)            # {"cd_loss": 0-dim tensor, "emd_loss": 0-dim tensor}

    The context is one ``winograd.weights_frozen(constant=True)`` block -- weights and running statistics do not change
    while evaluating, so transformed filters, stacked decoder weights and the BatchNorms' channel coefficients are made
    once, not per item.  ``graph=True`` [default on a GPU, ``FPSG_EVAL_GRAPH=0`` to switch off]: an item is ~200 launches of a
    few microseconds and the eager loop is ~17 % host / drain-bound, so everything in front of the EMD -- both encoders,
    the query decode, K1, the clouds' diameter -- is captured once per input shape and replayed; the Sinkhorn form's
    annealing schedule depends on that diameter (geomloss' rule), so the loop reads it (its one mid-item host read, as
    before) and launches K2b's ~12 kernels eagerly.  The first two items of a shape run eagerly (the libraries pick
    their kernels, the filter bank registers the layers); capture happens on the third.  A model whose ``emd_metric`` /
    ``pc_metric`` were replaced (tests drive the module with the oracle's functions) takes the plain method."""

    _KEYS = ("xs", "xq", "xad", "pcs", "pcq", "pcad")

    def __init__(self, model, graph: bool | None = None):
        self.model = model
        on_gpu = next(model.parameters()).is_cuda
        if graph is None:
            graph = os.environ.get("FPSG_EVAL_GRAPH", "1") != "0"
        self.use_graph = bool(graph) and on_gpu
        self._graphs = {}
        self._eager = {}
        self._block = None

    def __enter__(self):
        # constant: weights and running statistics stay as they are for the block's whole life, so a capture inside may
        # read what the eager items before it left in the block's cache (transformed filters, stacked decoder weights,
        # BatchNorm coefficients): none of those launches is in the graph
        self._block = winograd.weights_frozen(constant=True)
        self._block.__enter__()
        self._no_grad = torch.no_grad()
        self._no_grad.__enter__()
        return self

    def __exit__(self, *exc):
        self._graphs.clear()            # they read tensors of the block's cache, which ends here
        self._eager.clear()
        self._no_grad.__exit__(*exc)
        self._block.__exit__(*exc)
        self._block = None
        return False

    def _default_metrics(self) -> bool:
        from .metrics import chamfer_distance
        from .utils import emd_wrapper
        return self.model.emd_metric is emd_wrapper and self.model.pc_metric is chamfer_distance

    def __call__(self, sample):
        from .metrics import sinkhorn_divergence
        model = self.model
        if self._block is None:
            raise RuntimeError("EvalItem: call it inside its `with` block")
        if not self.use_graph or not self._default_metrics() or model.training:
            return model._return_reconstruction(sample)
        key = tuple((k, tuple(sample[k].shape)) for k in self._KEYS)
        if key not in self._graphs:
            n = self._eager.get(key, 0)
            if n < 2:
                self._eager[key] = n + 1
                return model._return_reconstruction(sample)
            # (pruned evaluation reads only the query images and the support / query clouds: the other three inputs --
            # 2 x 19 MB of images at 32 shots -- are neither cloned nor copied per item)
            used = ("xq", "pcs", "pcq") if model._eval_prune() else self._KEYS
            static = {k: (sample[k].clone() if k in used else sample[k]) for k in self._KEYS}
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, capture_error_mode="thread_local"):
                outs = model._reconstruct_for_eval(static)
            self._graphs[key] = (g, {k: static[k] for k in used}, outs)
        g, static, (syn_pc, ref_pc_q, cd_loss, diameter) = self._graphs[key]
        for k, buf in static.items():
            buf.copy_(sample[k], non_blocking=True)
        winograd.check_bank_before_replay()
        g.replay()
        emd = sinkhorn_divergence(syn_pc, ref_pc_q, diameter=float(diameter)).sum()      # emd_wrapper's value
        return {"cd_loss": cd_loss.clone(), "emd_loss": emd}
