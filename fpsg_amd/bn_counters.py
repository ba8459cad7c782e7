"""Bookkeeping of the fused BatchNorm paths: ``num_batches_tracked`` and the decoder's running statistics.

``nn.BatchNorm*`` adds 1 to its ``num_batches_tracked`` buffer in every training-mode forward
(``torch/nn/modules/batchnorm.py``); the fused paths (K5, the fold into K6 / K6f) do the same so
that checkpoints carry the counters of the reference.  One episode runs ~20 BatchNorm layers, i.e.
~20 one-element kernels; the decoder's grouped BatchNorm additionally updates the running mean / variance of its
16-64 modules per layer with multi-tensor ops, ~50 launches per episode.  Inside ``deferred()`` both are collected
and applied when the block ends: the counters as one multi-tensor add per increment value, the running statistics
as one ``mul`` + ``add`` pair per round (a module that ran twice in the block gets its updates in order) -- the same
operations on the same values, so the results are bit-identical."""
from __future__ import annotations

import contextlib
import threading

import torch


class _State(threading.local):
    """Per thread: a ``deferred()`` block collects the updates of the forward passes of ITS thread only (the process
    also runs the episode prefetcher and the RCCL watchdog; a second thread's training-mode forward must neither join
    nor lose this block's updates)."""
    pending = None       # id(tensor) -> [tensor, count] while a ``deferred()`` block is open
    stats = None         # [(tensors, scale, addends, alpha)] in call order while a block is open


_tls = _State()


def count_batch(bn, times: int = 1) -> None:
    """``bn.num_batches_tracked += times`` (now, or when the enclosing ``deferred()`` block ends)."""
    t = bn.num_batches_tracked
    if t is None:
        return
    if _tls.pending is None:
        t += times
        return
    entry = _tls.pending.get(id(t))
    if entry is None:
        _tls.pending[id(t)] = [t, times]
    else:
        entry[1] += times


def update_running(tensors, scale: float, addends, alpha: float = 1.0) -> None:
    """``t <- t * scale + alpha * a`` for every (t, a) of ``tensors`` / ``addends`` (running statistics), as the two
    multi-tensor ops ``_foreach_mul_`` / ``_foreach_add_(alpha=)`` -- now, or batched with the other updates of the
    enclosing ``deferred()`` block."""
    tensors, addends = list(tensors), list(addends)
    if _tls.stats is None:
        torch._foreach_mul_(tensors, scale)
        torch._foreach_add_(tensors, addends, alpha=alpha)
    else:
        _tls.stats.append((tensors, float(scale), addends, float(alpha)))


@contextlib.contextmanager
def deferred():
    if _tls.pending is not None:     # nested: the outer block applies everything
        yield
        return
    _tls.pending, _tls.stats = {}, []
    try:
        yield
    finally:
        items, _tls.pending = list(_tls.pending.values()), None
        stats, _tls.stats = _tls.stats, None
        by_count = {}
        for t, c in items:
            by_count.setdefault((c, t.device), []).append(t)
        # running statistics: round k holds every tensor's k-th update of the block, so updates of one tensor stay in order
        seen, rounds = {}, {}
        for tensors, scale, addends, alpha in stats:
            for t, a in zip(tensors, addends):
                k = seen.get(id(t), 0)
                seen[id(t)] = k + 1
                ts, scs, adds = rounds.setdefault((k, t.device, alpha), ([], [], []))
                ts.append(t)
                scs.append(scale)
                adds.append(a)
        with torch.no_grad():
            for (c, _dev), tensors in by_count.items():
                torch._foreach_add_(tensors, c)
            for key in sorted(rounds, key=lambda kd: kd[0]):
                ts, scs, adds = rounds[key]
                if all(sc == scs[0] for sc in scs):
                    torch._foreach_mul_(ts, scs[0])
                else:
                    torch._foreach_mul_(ts, scs)
                torch._foreach_add_(ts, adds, alpha=key[2])
