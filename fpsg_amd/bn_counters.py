"""``num_batches_tracked`` bookkeeping of the fused BatchNorm paths.

``nn.BatchNorm*`` adds 1 to its ``num_batches_tracked`` buffer in every training-mode forward
(``torch/nn/modules/batchnorm.py``); the fused paths (K5, the fold into K6 / K6f) do the same so
that checkpoints carry the counters of the reference.  One episode runs ~20 BatchNorm layers, i.e.
~20 one-element kernels; inside ``deferred()`` the increments are collected and applied as
multi-tensor adds when the block ends (same final values)."""
from __future__ import annotations

import contextlib

import torch

_pending = None      # id(tensor) -> [tensor, count] while a ``deferred()`` block is open


def count_batch(bn) -> None:
    """``bn.num_batches_tracked += 1`` (now, or when the enclosing ``deferred()`` block ends)."""
    t = bn.num_batches_tracked
    if t is None:
        return
    if _pending is None:
        t += 1
        return
    entry = _pending.get(id(t))
    if entry is None:
        _pending[id(t)] = [t, 1]
    else:
        entry[1] += 1


@contextlib.contextmanager
def deferred():
    global _pending
    if _pending is not None:         # nested: the outer block applies everything
        yield
        return
    _pending = {}
    try:
        yield
    finally:
        items, _pending = list(_pending.values()), None
        by_count = {}
        for t, c in items:
            by_count.setdefault((c, t.device), []).append(t)
        with torch.no_grad():
            for (c, _dev), tensors in by_count.items():
                torch._foreach_add_(tensors, c)
