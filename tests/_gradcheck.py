"""Gradient comparison against a float64 evaluation, with the reference's own fp32 arithmetic as
the yardstick.

Training-mode BatchNorm over a handful of samples and ReLU kinks make single gradient tensors of
these networks ill-conditioned in fp32: the REFERENCE formulation itself, run in fp32 on the CPU,
deviates from a float64 run of the same network by up to several percent on individual tensors
(a pre-activation that rounds to the other side of zero flips a ReLU mask) while the typical
tensor agrees to 1e-6.  A fixed per-tensor tolerance is therefore either loose enough to hide a
wrong gradient or fails at random.  The criterion used here:

  deviation(t) = max|g(t) - g64(t)| / max|g64(t)|            for every tensor t
  the HIP path's deviations must be distributed like those of the reference-arithmetic fp32 run
  (median, 90th percentile and maximum within `factor` of the yardstick's, plus small floors),
  and no tensor may deviate by more than `hard_max`.

A wrong gradient for one parameter group (missing term, wrong reduction) shows up as a
deviation of order 1 and trips `hard_max`; a systematic precision loss moves the median.
Tensors whose float64 gradient is exactly cancelled (a bias in front of a training-mode
BatchNorm) hold round-off on every side; they are bounded against their parent module's scale.
"""
from __future__ import annotations

import numpy as np
import torch


def _np(t):
    if isinstance(t, torch.Tensor):
        return t.detach().cpu().double().numpy()
    return np.asarray(t, dtype=np.float64)


def _parent(name: str) -> str:
    parts = name.split(".")
    return ".".join(parts[:-2]) if len(parts) > 2 else ""


def deviations(got: dict, truth64: dict, zero_rel: float = 1e-9):
    """-> (dev {name: relative deviation}, cancelled {name: max|got| / parent scale})."""
    truth = {k: _np(v) for k, v in truth64.items()}
    gmax = max(float(np.abs(v).max()) for v in truth.values())
    pscale = {}
    for k, v in truth.items():
        p = _parent(k)
        pscale[p] = max(pscale.get(p, 0.0), float(np.abs(v).max()))
    dev, cancelled = {}, {}
    for k, t in truth.items():
        g = _np(got[k])
        if g.shape != t.shape:                       # goldens store the first 64 rows of large tensors
            g = g[: t.shape[0]]
        s = float(np.abs(t).max())
        if s <= zero_rel * gmax:
            cancelled[k] = float(np.abs(g).max()) / max(pscale[_parent(k)], 1e-30)
        else:
            dev[k] = float(np.abs(g - t).max()) / s
    return dev, cancelled


def summarize(dev: dict) -> dict:
    v = np.array(sorted(dev.values()))
    worst = max(dev.items(), key=lambda kv: kv[1])
    return {"median": float(np.median(v)), "p90": float(np.quantile(v, 0.9)), "max": float(v[-1]),
            "worst": worst[0], "n": int(v.size)}


def assert_like_yardstick(got: dict, yard: dict, truth64: dict, what: str, factor: float = 4.0,
                          hard_max: float = 0.15, cancelled_max: float = 2e-2, ill_conditioned_ok: bool = False,
                          set_aside: dict | None = None):
    """`got` (HIP path, fp32), `yard` (reference arithmetic, fp32, CPU), `truth64` (float64).
    `ill_conditioned_ok`: tensors on which the REFERENCE arithmetic itself misses the float64 gradient by more than
    `hard_max` are set aside -- with a BatchNorm over a batch of two (1-shot episodes) the normalised values are +-1
    whatever the input, the true gradient behind it is O(eps) and fp32 holds round-off there on every side; for those the
    HIP path's values must stay on the scale of the reference arithmetic's."""
    dg, cg = deviations(got, truth64)
    dy, _ = deviations(yard, truth64)
    if ill_conditioned_ok:
        ill = sorted(k for k in dy if dy[k] > hard_max)
        for k in ill:
            a, b = float(np.abs(_np(got[k])).max()), float(np.abs(_np(yard[k])).max())
            assert a <= 10.0 * b + 1e-30, (what, "ill-conditioned tensor off the reference arithmetic's scale", k, a, b)
            dg.pop(k, None)
            dy.pop(k, None)
        # the exactly cancelled ones likewise: their parent module's true gradient is O(eps) too, so the parent scale is
        # no yardstick; the reference arithmetic's own round-off is
        for k in sorted(cg):
            a, b = float(np.abs(_np(got[k])).max()), float(np.abs(_np(yard[k])).max())
            assert a <= 10.0 * b + 1e-30, (what, "cancelled tensor off the reference arithmetic's scale", k, a, b)
        if set_aside is not None:       # the caller pins WHICH tensors may be set aside (a regression must not hide there)
            set_aside["ill_conditioned"] = list(ill)
            set_aside["cancelled"] = sorted(cg)
        print(f"{what}: {len(ill)} + {len(cg)} tensors set aside (the reference fp32 arithmetic itself is > {hard_max} from "
              f"float64 on them / their float64 gradient is zero)")
        cg = {}
    sg, sy = summarize(dg), summarize(dy)
    print(f"{what}: gradient deviation from float64 -- HIP path {sg} | reference fp32 arithmetic {sy}")
    assert sg["median"] <= factor * sy["median"] + 2e-6, (what, sg, sy)
    assert sg["p90"] <= factor * sy["p90"] + 2e-5, (what, sg, sy)
    assert sg["max"] <= max(factor * sy["max"], 2e-3), (what, sg, sy)
    assert sg["max"] <= hard_max, (what, sg)
    bad = {k: v for k, v in cg.items() if v > cancelled_max}
    assert not bad, (what, "cancelled-gradient tensors above round-off", bad)
    return sg, sy
