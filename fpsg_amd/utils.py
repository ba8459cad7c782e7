"""Template samplers and the EMD wrapper -- the live parts of reference
``src/models/utils.py`` (``emd_wrapper :12-13``, ``SquareTemplate.get_random_points
:51-54``, ``ShpereTemplate.get_random_points :28-33``, ``get_template :90-96``).

Differences that are deliberate:
  * samplers draw on the device they are asked for (the reference hard-codes
    ``torch.cuda.FloatTensor``) and accept a ``torch.Generator`` so that a run can be
    reproduced; with no generator they use the device's global RNG like the reference;
  * the pymesh-based ``get_regular_points`` variants are not on any training / evaluation
    path of the reference and are out of scope.
"""
from __future__ import annotations

import torch


def emd_wrapper(pc1: torch.Tensor, pc2: torch.Tensor) -> torch.Tensor:
    """Reference ``emd_wrapper(pc1, pc2)`` = ``emd_loss(pc1, pc2, reduce='sum', sinkhorn=True)``:
    Sinkhorn-divergence EMD summed over the batch, 0-dim."""
    from .metrics import emd_loss
    return emd_loss(pc1, pc2, reduce="sum", sinkhorn=True)


class Template:
    dim = 0

    def __init__(self, device=0):
        self.device = device
        self.npoints = 0

    def get_random_points(self, shape, device="cuda", generator=None):
        raise NotImplementedError


class SquareTemplate(Template):
    """Uniform samples of the unit square: ``shape = [B, 2, P]`` -> U(0,1)."""
    dim = 2

    def get_random_points(self, shape, device="cuda", generator=None):
        grid = torch.empty(tuple(shape), dtype=torch.float32, device=device)
        return grid.uniform_(0, 1, generator=generator)


class ShpereTemplate(Template):
    """Gaussian samples in R^3: ``shape = [B, 3, P]``.  (The reference computes, but then
    discards, the projection onto the unit sphere -- ``utils.py:32`` is an expression
    statement -- so the returned points are the raw normal samples; kept as is.)"""
    dim = 3

    def get_random_points(self, shape, device="cuda", generator=None):
        assert shape[1] == 3, f"3 is expected in dimension 1, while got {shape}"
        grid = torch.empty(tuple(shape), dtype=torch.float32, device=device)
        return grid.normal_(0, 1, generator=generator)


def get_template(template_type: str, device=0) -> Template:
    kinds = {"SQUARE": SquareTemplate, "SPHERE": ShpereTemplate}
    if template_type not in kinds:
        raise ValueError(f"Invalid template: {template_type}")
    return kinds[template_type](device=device)
