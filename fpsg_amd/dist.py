"""Episode-sharded data parallelism: one process per GPU, RCCL all-reduce over xGMI.

The reference is single-process / single-GPU (no ``torch.distributed`` anywhere, SURVEY.md
2b); this module is the new multi-GPU row of the hot path:

  * a global optimizer step consists of ``E`` independent episodes; rank ``r`` of ``W``
    runs episodes ``r, r+W, ...`` one after the other (BatchNorm statistics stay
    per-episode -- no SyncBN, as in the reference) and accumulates their gradients;
  * the step's gradient lives in ONE flat fp32 buffer (77 M elements = 310 MB for the full
    model), cut into a few large buckets.  An episode's backward writes fresh ``.grad``
    tensors which a handful of multi-tensor launches copy / add into the flat buffer
    (``absorb``) -- autograd's own accumulation would be one small add kernel per parameter
    per episode, ~600 launches.  With more than one rank the LAST local episode instead
    also leaves fresh gradient tensors; the autograd hooks count them per bucket and, as soon
    as a bucket is complete (decoder first, VGG last), absorb ITS gradients with one
    multi-tensor launch and start its ``all_reduce(SUM)`` asynchronously, overlapping the
    collective with the rest of backward; xGMI is point-to-point, so few large messages are
    what keep each link busy;
  * the summed gradient is divided by ``E`` (mean over the step's episodes), so a step has
    the gradient scale of the reference's one-episode step and ``--lr`` keeps its meaning.

``backend='nccl'`` is RCCL on ROCm; on CPU the same code runs over ``gloo`` (tests).
"""
from __future__ import annotations

import os
from typing import Iterable

import torch
import torch.distributed as dist
import torch.nn as nn

from .optim import layout_order


def env_world() -> tuple[int, int, int]:
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def init_distributed(device_type: str | None = None) -> tuple[int, int, torch.device]:
    """Initialises the default process group from the torchrun environment.
    Returns ``(rank, world, device)``; a no-op (0, 1, device) when WORLD_SIZE is 1."""
    rank, local_rank, world = env_world()
    if device_type is None:
        device_type = "cuda" if torch.cuda.is_available() else "cpu"
    if device_type == "cuda":
        # FPSG_LOCAL_DEVICE: rehearsals of the multi-rank path on a box with fewer GPUs than ranks (every rank
        # on the given device, with FPSG_DIST_BACKEND=gloo -- RCCL refuses two ranks on one device)
        index = int(os.environ.get("FPSG_LOCAL_DEVICE", local_rank))
        torch.cuda.set_device(index)
        device = torch.device("cuda", index)
    else:
        device = torch.device("cpu")
    force = bool(os.environ.get("FPSG_FORCE_DIST"))   # single-rank group: exercises RCCL on one GPU
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        backend = os.environ.get("FPSG_DIST_BACKEND") or ("nccl" if device_type == "cuda" else "gloo")
        kwargs = {}
        if device_type == "cuda" and backend == "nccl":
            kwargs["device_id"] = device
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kwargs)
    return rank, world, device


def shutdown() -> None:
    if dist.is_initialized():
        dist.destroy_process_group()


class FlatGradBuckets:
    """Flat gradient storage + bucketed, overlapped all-reduce for ``model``."""

    _MAX_STASH = 8          # fpsg_flat_accumulate_tables' limit

    def __init__(self, model: nn.Module, bucket_mb: float = 80.0, group=None):
        params = [p for p in model.parameters() if p.requires_grad]
        if not params:
            raise ValueError("model has no trainable parameters")
        self.params = params
        self.group = group
        dev, dt = params[0].device, params[0].dtype
        total = sum(p.numel() for p in params)
        self.flat = torch.zeros(total, dtype=dt, device=dev)
        # Buckets follow the order gradients become ready in backward = reverse of the
        # registration order (decoder, point encoder, image encoder).
        cap = max(1, int(bucket_mb * (1 << 20) / self.flat.element_size()))
        self.buckets: list[tuple[int, int]] = []   # [start, end) into flat
        self._bucket_of: dict[int, int] = {}
        self._bucket_size: list[int] = []
        off, start, count = 0, 0, 0
        self.views: dict[int, torch.Tensor] = {}
        order = layout_order(params)             # the layout of fpsg_amd.optim.flat_layout
        for p in order:
            n = p.numel()
            # same memory format as the parameter (e.g. channels_last conv weights): the fused
            # optimizer requires param and grad layouts to match
            self.views[id(p)] = self.flat[off:off + n].as_strided(p.size(), p.stride())
            self._bucket_of[id(p)] = len(self.buckets)
            off += n
            count += 1
            if off - start >= cap:
                self.buckets.append((start, off))
                self._bucket_size.append(count)
                start, count = off, 0
        if off > start:
            self.buckets.append((start, off))
            self._bucket_size.append(count)
        self._pending = [0] * len(self.buckets)
        self._next_launch = 0        # buckets [0, _next_launch) have had their all-reduce launched this step
        self._bucket_params: list[list] = [[] for _ in self.buckets]
        for p in params:
            self._bucket_params[self._bucket_of[id(p)]].append(p)
        self._handles: list = []
        self._armed = False
        self._armed_first = False
        self._streams: dict = {}     # every stream a gradient was produced on in this backward
        self._hooks = [p.register_post_accumulate_grad_hook(self._on_grad) for p in params]
        # segment table of the flat layout (K7's format): absorb() adds an episode's gradient tensors in one launch
        self._layout_params = order
        self._seg_off = self._gtab = None
        if dev.type == "cuda" and dt == torch.float32:
            offs, o = [], 0
            for p in self._layout_params:
                offs.append(o)
                o += p.numel()
            self._seg_off = torch.tensor(offs + [o], dtype=torch.int64, device=dev)
            self._gtab = torch.zeros(len(self._layout_params), dtype=torch.int64, device=dev)
            self._gtabs = torch.zeros(self._MAX_STASH * len(self._layout_params), dtype=torch.int64, device=dev)
        # absorb() keeps an episode's gradient tensors and adds up to _MAX_STASH episodes in one launch (flush()): the
        # flat buffer is read and written once per flush instead of once per episode; same sums in the same order
        self.lazy = True
        self._stash: list = []
        self._stash_first = False
        # the stash keeps whole gradient sets alive (310 MB each for the full model): at most _MAX_STASH of them, at most
        # FPSG_STASH_MB megabytes [4096], and never more than an eighth of the memory that is free when a step's first
        # episode is stashed (``_current_stash_cap``: activations and the allocator's pools exist by then) -- a nearly
        # full device flushes after every episode or two instead of failing an allocation.  The cap is rank-local and
        # changes only WHEN the adds happen: flush() adds the stashed sets in episode order, so the sums are the same
        # for every cap (tests/test_step_switches_gpu.py: cap 1 against _MAX_STASH, bit for bit).
        raw = os.environ.get("FPSG_STASH_MB", "4096")
        try:
            self._stash_budget = float(raw) * (1 << 20)
        except ValueError:
            raise ValueError(f"FPSG_STASH_MB={raw!r}: expected a number of megabytes") from None
        if not (self._stash_budget >= 0.0):
            raise ValueError(f"FPSG_STASH_MB={raw!r}: expected a non-negative number of megabytes")
        self.stash_cap_override = None       # tests / experiments: a fixed number of episodes per flush
        self._stash_cap = self._current_stash_cap()
        self.attach()

    # -- bookkeeping ---------------------------------------------------------------
    @property
    def world(self) -> int:
        return dist.get_world_size(self.group) if dist.is_initialized() else 1

    @staticmethod
    def world_initialised() -> bool:
        return dist.is_initialized()

    def zero(self) -> None:
        """Replaces ``optimizer.zero_grad()``: the flat buffer is the step's gradient."""
        self._stash, self._stash_first = [], False
        self.flat.zero_()

    def attach(self) -> None:
        """``p.grad`` = the parameter's view of the flat buffer: backward accumulates in place
        (the armed, overlapped episode) and the optimizer reads the step's gradient."""
        for p in self.params:
            p.grad = self.views[id(p)]

    def detach(self) -> None:
        """``p.grad = None``: the next backward leaves fresh gradient tensors (no add kernels)."""
        for p in self.params:
            p.grad = None

    def absorb(self, first: bool = False) -> None:
        """Adds the gradients a backward left in ``p.grad`` into the flat buffer with
        multi-tensor launches; ``first`` overwrites instead (no ``zero()`` needed).  The sums
        are the same fp32 adds, in the same order over episodes, as in-place accumulation."""
        if self.lazy and self._stash_episode(first):
            return
        self.flush()
        if self._absorb_segments(first):
            return
        dst, src, missing = [], [], []
        for p in self.params:
            if p.grad is None:
                missing.append(self.views[id(p)])
            else:
                dst.append(self.views[id(p)])
                src.append(p.grad)
        with torch.no_grad():
            if first:
                if dst:
                    torch._foreach_copy_(dst, src)
                if missing:
                    torch._foreach_zero_(missing)
            elif dst:
                torch._foreach_add_(dst, src)

    def _segment_pointers(self):
        """The gradient tensors' addresses in layout order (0: no gradient) when every gradient is a dense fp32 tensor
        laid out like its parameter, else None."""
        dev = self.flat.device
        ptrs = []
        for p in self._layout_params:
            g = p.grad
            if g is None:
                ptrs.append(0)
            elif (g.dtype == torch.float32 and g.device == dev and g.numel() == p.numel() and g.stride() == p.stride()
                  and g.data_ptr() != self.views[id(p)].data_ptr()):
                ptrs.append(g.data_ptr())
            else:
                return None
        return ptrs

    def _current_stash_cap(self) -> int:
        """Episodes per flush for the step that starts now: the budget over the size of one gradient set, with the
        device's free memory (outside plus inside the caching allocator's pools) sampled at this moment."""
        if self.stash_cap_override is not None:
            return max(1, min(self._MAX_STASH, int(self.stash_cap_override)))
        budget = self._stash_budget
        dev = self.flat.device
        if dev.type == "cuda":
            try:
                free = torch.cuda.mem_get_info(dev)[0] + torch.cuda.memory_reserved(dev) - torch.cuda.memory_allocated(dev)
                budget = min(budget, free / 8)
            except RuntimeError:
                pass
        return max(1, min(self._MAX_STASH, int(budget // max(1, self.flat.numel() * self.flat.element_size()))))

    def _stash_episode(self, first: bool) -> bool:
        """``absorb`` deferred: the episode's gradient tensors are kept (``detach()`` drops the parameters' references,
        not these) and added by ``flush()`` together with the following episodes'.  False -> not applicable here."""
        if (self._gtab is None or os.environ.get("FPSG_ABSORB_SEGMENTS", "1") == "0"
                or os.environ.get("FPSG_ABSORB_LAZY", "1") == "0" or torch.cuda.is_current_stream_capturing()):
            return False
        ptrs = self._segment_pointers()
        if ptrs is None:
            return False
        if first and self._stash:
            self.flush()
        if not self._stash:
            self._stash_first = bool(first)
            self._stash_cap = self._current_stash_cap()      # re-sampled per group: peak-time free memory, not construction-time
        self._stash.append((ptrs, [p.grad for p in self._layout_params]))
        if len(self._stash) >= self._stash_cap:
            self.flush()
        return True

    def flush(self) -> None:
        """Adds the kept episodes' gradients into the flat buffer, in episode order, in ONE launch
        (``fpsg_flat_accumulate_tables``); called before anything reads or adds to the flat buffer."""
        if not self._stash:
            return
        from . import _hip
        stash, first = self._stash, self._stash_first
        self._stash, self._stash_first = [], False
        n = len(self._layout_params)
        table = torch.tensor([q for ptrs, _ in stash for q in ptrs], dtype=torch.int64).pin_memory()
        self._gtabs[:len(stash) * n].copy_(table, non_blocking=True)
        with torch.cuda.device(self.flat.device):
            rc = _hip.load().fpsg_flat_accumulate_tables(_hip.ptr(self.flat), _hip.ptr(self._gtabs), _hip.ptr(self._seg_off),
                                                         n, len(stash), self.flat.numel(), 0 if first else 1,
                                                         _hip.stream_of(self.flat))
        _hip.check(rc, "fpsg_flat_accumulate_tables")

    def _absorb_segments(self, first: bool) -> bool:
        """``absorb`` as ONE launch (``fpsg_flat_accumulate_segments``: the gradient tensors read through a pointer
        table, the flat buffer updated at HBM rate) when every gradient is a dense fp32 tensor laid out like its
        parameter; False -> the multi-tensor path.  Not under stream capture (the table upload is a host copy)."""
        if self._gtab is None or os.environ.get("FPSG_ABSORB_SEGMENTS", "1") == "0" or torch.cuda.is_current_stream_capturing():
            return False
        dev = self.flat.device
        ptrs = self._segment_pointers()
        if ptrs is None:
            return False
        from . import _hip
        # a fresh pinned tensor per call: the caching host allocator keeps it until the copy ran
        self._gtab.copy_(torch.tensor(ptrs, dtype=torch.int64).pin_memory(), non_blocking=True)
        with torch.cuda.device(dev):
            rc = _hip.load().fpsg_flat_accumulate_segments(_hip.ptr(self.flat), _hip.ptr(self._gtab), _hip.ptr(self._seg_off),
                                                           len(ptrs), self.flat.numel(), 0 if first else 1,
                                                           _hip.stream_of(self.flat))
        _hip.check(rc, "fpsg_flat_accumulate_segments")
        return True

    def arm(self, first: bool = False) -> None:
        """Call (after ``detach()``) before the backward of the LAST local episode of a step:
        each bucket is absorbed into the flat buffer (copied if ``first``: the step's only local
        episode) and all-reduced as soon as its gradients are complete during that backward."""
        self.flush()            # the hooks add to the flat buffer during the backward
        self._armed = dist.is_initialized()
        self._armed_first = bool(first)
        self._pending = list(self._bucket_size)
        self._next_launch = 0
        self._handles = []
        self._streams = {}

    def _absorb_bucket(self, b: int) -> None:
        dst, src, missing = [], [], []
        for p in self._bucket_params[b]:
            if p.grad is None:
                missing.append(self.views[id(p)])
            elif p.grad.data_ptr() != self.views[id(p)].data_ptr():
                dst.append(self.views[id(p)])
                src.append(p.grad)
        with torch.no_grad():
            if self._armed_first:
                if dst:
                    torch._foreach_copy_(dst, src)
                if missing:
                    torch._foreach_zero_(missing)
            elif dst:
                torch._foreach_add_(dst, src)

    def _on_grad(self, p: torch.Tensor) -> None:
        if not self._armed:
            return
        b = self._bucket_of[id(p)]
        self._pending[b] -= 1
        if p.is_cuda:
            cur = torch.cuda.current_stream(p.device)
            self._streams[cur.cuda_stream] = cur
        if self._pending[b] == 0:
            self._launch_ready(cur if p.is_cuda else None)

    def _launch_ready(self, cur=None, flush: bool = False) -> None:
        """Launches the all-reduce of every complete bucket IN BUCKET-INDEX ORDER: bucket b goes out
        only once buckets 0..b-1 have gone.  Every rank therefore issues the same sequence of
        collectives whatever order its backward completed them in (a rank without local episodes
        flushes all of them from ``finish``; the stacked decoder weights deliver the gradients of
        several buckets at once) -- mismatched sequences would hang or mis-sum in RCCL.  ``flush``
        also launches the incomplete ones (parameters that received no gradient this step)."""
        while self._next_launch < len(self.buckets) and (flush or self._pending[self._next_launch] == 0):
            b = self._next_launch
            self._next_launch += 1
            if cur is not None:
                # a bucket may hold gradients produced on different streams (the encoders can
                # run on two): the collective is ordered after all of them
                for key, st in self._streams.items():
                    if key != cur.cuda_stream:
                        cur.wait_stream(st)
            self._absorb_bucket(b)
            s, e = self.buckets[b]
            self._handles.append(
                dist.all_reduce(self.flat[s:e], op=dist.ReduceOp.SUM, group=self.group,
                                async_op=True))

    def finish(self, n_episodes_global: int, scale: bool = True) -> None:
        """Waits for the in-flight buckets, reduces any bucket whose parameters received no
        gradient in the armed backward, and turns the sum into the mean over episodes --
        unless ``scale`` is False: the caller's optimizer then applies 1/E itself while it
        reads the buffer (``fpsg_adam_step``'s ``grad_scale``: the same fp32 product per
        element, without a separate read-modify-write pass over the 310 MB buffer on the
        critical path behind the last all-reduce), and the buffer keeps the SUM."""
        self.flush()
        if self._armed:
            cur = torch.cuda.current_stream(self.flat.device) if self.flat.is_cuda else None
            late = len(self.buckets) - self._next_launch
            fired = sum(self._bucket_size) - sum(self._pending)
            if late and fired and not getattr(self, "_warned_late", False):
                # a backward ran here, yet some buckets never completed: parameters without a gradient hold their bucket
                # -- and, because launches go out in index order, every later one -- back until now (no overlap with the
                # backward).  Results are unaffected.
                self._warned_late = True
                import warnings
                silent = [b for b in range(self._next_launch, len(self.buckets)) if self._pending[b] > 0]
                warnings.warn(f"FlatGradBuckets: {late} of {len(self.buckets)} gradient buckets were all-reduced only "
                              f"after the backward (buckets {silent} hold parameters that received no gradient); "
                              "freeze such parameters (requires_grad=False) to restore the overlap")
            self._launch_ready(cur, flush=True)      # the rest, in index order (e.g. parameters unused by this step's graph)
            for h in self._handles:
                h.wait()
            self._handles = []
            self._armed = False
        if scale and n_episodes_global > 1:
            self.flat.mul_(1.0 / n_episodes_global)

    def remove(self) -> None:
        for h in self._hooks:
            h.remove()


def broadcast_parameters(model: nn.Module, src: int = 0) -> None:
    """Makes every rank start from rank ``src``'s weights and buffers."""
    if not dist.is_initialized():
        return
    for t in list(model.parameters()) + list(model.buffers()):
        dist.broadcast(t.data, src=src)


class BufferSync:
    """Reconciles the BatchNorm buffers of the replicas before rank 0 evaluates or saves.

    The reference trains on one GPU, so its checkpoint's running statistics have seen every episode
    (``trainNetwork.py:140-148,192-197``).  Under episode-level data parallelism each rank's BatchNorm layers see 1/W
    of the episodes; there is no SyncBN (per-episode batch statistics are the reference's semantics), so the running
    statistics drift apart.  ``sync()`` replaces every floating-point buffer (``running_mean``, ``running_var``) by its
    mean over the ranks -- an exponential average is linear in its inputs, so this is the running average a single
    process would hold had each of its updates been the mean of the W ranks' batch statistics -- and every integer
    buffer (``num_batches_tracked``) by its value at the last sync plus the SUM of the ranks' increments since, the
    count a single process would have reached.  One flat all-reduce per dtype; a no-op without a process group."""

    def __init__(self, model: nn.Module, group=None):
        self.group = group
        bufs = [b for b in model.buffers() if b is not None]
        self.floats = [b for b in bufs if b.is_floating_point()]
        self.ints = [b for b in bufs if not b.is_floating_point()]
        self._base = [b.detach().clone() for b in self.ints]

    def rebase(self) -> None:
        """After the buffers were overwritten from outside (a checkpoint load)."""
        self._base = [b.detach().clone() for b in self.ints]

    @torch.no_grad()
    def sync(self) -> None:
        if not dist.is_initialized():
            return
        world = dist.get_world_size(self.group)
        if self.floats:
            flat = torch.cat([b.detach().reshape(-1).to(torch.float32) for b in self.floats])
            dist.all_reduce(flat, group=self.group)
            flat /= world
            off = 0
            for b in self.floats:
                n = b.numel()
                b.copy_(flat[off:off + n].view(b.shape))
                off += n
        if self.ints:
            delta = torch.cat([(b.detach() - base).reshape(-1).to(torch.int64) for b, base in zip(self.ints, self._base)])
            dist.all_reduce(delta, group=self.group)
            off = 0
            for b, base in zip(self.ints, self._base):
                n = b.numel()
                b.copy_((base.reshape(-1).to(torch.int64) + delta[off:off + n]).view(b.shape).to(b.dtype))
                off += n
            self.rebase()


def gather_objects(obj, group=None) -> list:
    """``[obj of rank 0, obj of rank 1, ...]`` on every rank (``[obj]`` without a process group)."""
    if not dist.is_initialized():
        return [obj]
    out = [None] * dist.get_world_size(group)
    dist.all_gather_object(out, obj, group=group)
    return out


def broadcast_object(obj, src: int = 0, group=None):
    """Rank ``src``'s ``obj`` on every rank."""
    if not dist.is_initialized():
        return obj
    box = [obj]
    dist.broadcast_object_list(box, src=src, group=group)
    return box[0]


def agree(error: str | None, what: str, group=None) -> None:
    """Every rank reports how a rank-local step went (``None`` = fine, else a message); if ANY rank failed, EVERY rank
    raises the same RuntimeError -- a rank that raised alone would leave the others waiting in the next collective
    (file reads at resume: a checkpoint that one node cannot see)."""
    if not dist.is_initialized():
        if error is not None:
            raise RuntimeError(f"{what}: {error}")
        return
    box = [None] * dist.get_world_size(group)
    dist.all_gather_object(box, error, group=group)
    bad = [(r, e) for r, e in enumerate(box) if e is not None]
    if bad:
        raise RuntimeError(f"{what} failed on " + "; ".join(f"rank {r}: {e}" for r, e in bad))


def all_reduce_scalars(values: Iterable[float], device) -> list[float]:
    """Sum of a few python floats over ranks (logging only)."""
    t = torch.tensor(list(values), dtype=torch.float64, device=device)
    if dist.is_initialized():
        if device.type == "cuda":
            t = t.float()
        dist.all_reduce(t)
    return t.tolist()
