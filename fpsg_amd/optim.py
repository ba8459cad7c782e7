"""Adam over flat buffers (K7): the optimizer step of the train loop as one HBM stream.

``FlatAdam`` is ``torch.optim.Adam(lr, betas, eps)`` as the reference configures it
(``src/trainNetwork.py:118-123``: no weight decay, no amsgrad) for a model whose parameters all
live on one ROCm device in fp32.  At construction the parameters are re-pointed into ONE flat
buffer (``p.data`` becomes a view; values, ``Parameter`` objects and state-dict keys are
untouched), the two moments are flat buffers of the same layout, and ``step()`` is a single
launch of ``fpsg_adam_step`` (4 reads + 3 writes per element) instead of a multi-tensor sweep
over ~600 tensors.  The layout is that of ``fpsg_amd.dist.FlatGradBuckets`` (``flat_layout``), so
the step's flat gradient buffer is consumed in place.  Gradients that are NOT views of a bound flat
buffer -- a step of one episode on one rank leaves them where autograd put them -- are read through
a table of per-parameter pointers (``fpsg_adam_step_segments``: no 310 MB gather); only
non-contiguous gradients are first gathered with one multi-tensor copy.

It is a ``torch.optim.Optimizer``: ``param_groups[0]['lr']`` (StepLR), ``state_dict()`` /
``load_state_dict()`` (per-parameter ``step`` / ``exp_avg`` / ``exp_avg_sq`` entries, i.e. the
format of ``torch.optim.Adam``) keep working.  A parameter without a gradient counts as a zero
gradient (torch skips it): with the flat gradient buffer of the train step every parameter has one.
"""
from __future__ import annotations

import torch
from torch.optim import Optimizer

from . import _hip


def layout_order(params):
    """The parameters in the order of the flat buffers: reverse registration order (the order in which gradients
    become ready in backward), except that a *stack group* -- parameters tagged ``_fpsg_stack = (group, index, n)``
    by a module that evaluates them as one stacked tensor (``PCDecoder``: the same layer of its 16 patch MLPs) -- is
    placed whole, in index order, where its first member falls.  Densely packed, the members of a group are then the
    rows of a contiguous ``[n, ...]`` tensor inside the flat buffer: stacking them is a view.  A group that is
    incomplete among ``params`` (or whose members differ in shape) is not kept together."""
    params = list(params)
    groups = {}
    for p in params:
        tag = getattr(p, "_fpsg_stack", None)
        if tag is not None:
            groups.setdefault(tag[0], []).append((tag[1], p))
    out, placed = [], set()
    for p in reversed(params):
        if id(p) in placed:
            continue
        members = [p]
        tag = getattr(p, "_fpsg_stack", None)
        if tag is not None:
            g = sorted(groups[tag[0]], key=lambda t: t[0])
            if len(g) == tag[2] and [i for i, _ in g] == list(range(tag[2])) and all(m.shape == p.shape for _, m in g):
                members = [m for _, m in g]
        for m in members:
            out.append(m)
            placed.add(id(m))
    return out


def flat_layout(params):
    """``[(param, offset, numel)]`` of the flat buffers: ``layout_order``, densely packed."""
    out, off = [], 0
    for p in layout_order(params):
        out.append((p, off, p.numel()))
        off += p.numel()
    return out, off


class FlatAdam(Optimizer):
    def __init__(self, params, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8):
        if lr < 0 or eps < 0 or not (0 <= betas[0] < 1) or not (0 <= betas[1] < 1):
            raise ValueError("FlatAdam: invalid lr / eps / betas")
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps))
        if len(self.param_groups) != 1:
            raise ValueError("FlatAdam: one parameter group (the reference's configuration)")
        ps = [p for p in self.param_groups[0]["params"] if p.requires_grad]
        if not ps:
            raise ValueError("FlatAdam: no trainable parameters")
        dev = ps[0].device
        if dev.type != "cuda" or any(p.device != dev or p.dtype != torch.float32 for p in ps):
            raise ValueError("FlatAdam: all parameters must be fp32 on one ROCm device")
        self._layout, total = flat_layout(ps)
        self.flat_param = torch.empty(total, dtype=torch.float32, device=dev)
        self.flat_exp_avg = torch.zeros(total, dtype=torch.float32, device=dev)
        self.flat_exp_avg_sq = torch.zeros(total, dtype=torch.float32, device=dev)
        self._gather = None                    # private flat gradient buffer (non-contiguous gradients)
        self._bound = None                     # the train step's flat gradient buffer
        offs = [off for _, off, _ in self._layout] + [total]
        self._seg_off = torch.tensor(offs, dtype=torch.int64, device=dev)
        self._gtab = torch.zeros(len(self._layout), dtype=torch.int64, device=dev)
        self._gtab_host = None                 # the pointers currently in _gtab
        self._t = 0
        self.grad_scale = 1.0                  # multiplies the gradient as the step reads it (TrainStep: 1/E, then reset)
        self._step_tensor = torch.zeros((), dtype=torch.float32)       # shared by every state entry
        with torch.no_grad():
            for p, off, n in self._layout:
                view = self.flat_param[off:off + n].view(p.shape)
                view.copy_(p.data)
                p.data = view
                self.state[p] = {"step": self._step_tensor,
                                 "exp_avg": self.flat_exp_avg[off:off + n].view(p.shape),
                                 "exp_avg_sq": self.flat_exp_avg_sq[off:off + n].view(p.shape)}

    # ------------------------------------------------------------------ gradients
    def bind_gradients(self, flat_grad: torch.Tensor) -> None:
        """``flat_grad``: a flat buffer with ``flat_layout`` of the same parameters (FlatGradBuckets)."""
        if flat_grad.numel() != self.flat_param.numel() or flat_grad.device != self.flat_param.device:
            raise ValueError("FlatAdam.bind_gradients: buffer does not match the parameter layout")
        self._bound = flat_grad

    def _bound_gradient(self):
        first, off0, _ = self._layout[0]
        last, off1, _ = self._layout[-1]
        b = self._bound
        if (b is not None and first.grad is not None and last.grad is not None
                and first.grad.data_ptr() == b.data_ptr() + 4 * off0 and last.grad.data_ptr() == b.data_ptr() + 4 * off1):
            return b
        return None

    def _pointer_table(self):
        """Device table of the parameters' gradient pointers (0 = no gradient), or None when a
        gradient cannot be read in place (not contiguous fp32 on the parameters' device)."""
        dev = self.flat_param.device
        ptrs = []
        for p, _, n in self._layout:
            g = p.grad
            if g is None:
                ptrs.append(0)
            elif g.dtype == torch.float32 and g.device == dev and g.is_contiguous() and g.numel() == n:
                ptrs.append(g.data_ptr())
            else:
                return None
        if ptrs != self._gtab_host:
            # a fresh pinned tensor per change: the caching host allocator keeps it until the copy ran
            self._gtab.copy_(torch.tensor(ptrs, dtype=torch.int64).pin_memory(), non_blocking=True)
            self._gtab_host = ptrs
        return self._gtab

    def _flat_gradient(self) -> torch.Tensor:
        b = self._bound_gradient()
        if b is not None:
            return b
        if self._gather is None:
            self._gather = torch.zeros_like(self.flat_param)
        dst, src, missing = [], [], []
        for p, off, n in self._layout:
            view = self._gather[off:off + n].view(p.shape)
            (missing if p.grad is None else dst).append(view)
            if p.grad is not None:
                src.append(p.grad)
        if dst:
            torch._foreach_copy_(dst, src)
        if missing:
            torch._foreach_zero_(missing)
        return self._gather

    # ------------------------------------------------------------------------ step
    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        group = self.param_groups[0]
        first = self._layout[0][0]
        if first.data_ptr() != self.flat_param.data_ptr() + 4 * self._layout[0][1]:
            raise RuntimeError("FlatAdam: the parameters were moved off the flat buffer (model.to() / "
                               "a new .data after the optimizer was built); rebuild the optimizer")
        g = self._bound_gradient()
        table = self._pointer_table() if g is None else None
        if g is None and table is None:
            g = self._flat_gradient()
        self._t += 1
        self._step_tensor.fill_(float(self._t))
        lib = _hip.load()
        hyper = (float(group["lr"]), float(group["betas"][0]), float(group["betas"][1]), float(group["eps"]), self._t,
                 float(self.grad_scale), _hip.stream_of(self.flat_param))
        with torch.cuda.device(self.flat_param.device):
            if g is not None:
                rc = lib.fpsg_adam_step(_hip.ptr(self.flat_param), _hip.ptr(g), _hip.ptr(self.flat_exp_avg),
                                        _hip.ptr(self.flat_exp_avg_sq), self.flat_param.numel(), *hyper)
            else:
                rc = lib.fpsg_adam_step_segments(_hip.ptr(self.flat_param), _hip.ptr(table), _hip.ptr(self._seg_off),
                                                 len(self._layout), _hip.ptr(self.flat_exp_avg),
                                                 _hip.ptr(self.flat_exp_avg_sq), self.flat_param.numel(), *hyper)
        _hip.check(rc, "fpsg_adam_step")
        return loss

    # ------------------------------------------------------------------ state dict
    def load_state_dict(self, state_dict) -> None:
        super().load_state_dict(state_dict)
        t = 0
        with torch.no_grad():
            for p, off, n in self._layout:
                st = self.state.get(p, {})
                for key, flat in (("exp_avg", self.flat_exp_avg), ("exp_avg_sq", self.flat_exp_avg_sq)):
                    view = flat[off:off + n].view(p.shape)
                    if key in st:
                        view.copy_(st[key])
                    else:
                        view.zero_()
                    st[key] = view
                if "step" in st:
                    t = max(t, int(float(st["step"])))
                st["step"] = self._step_tensor
                self.state[p] = st
        self._t = t
        self._step_tensor.fill_(float(t))
