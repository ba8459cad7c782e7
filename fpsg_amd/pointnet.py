"""PointNet global-feature encoder (``[B,3,N] -> [B,1024]``) with its 3x3 input transform.

Follows reference ``src/pointnet/model.py:11-45`` (``STN3d``) and ``:199-239``
(``PointNetfeat``).  Attribute names are kept (``stn.conv1 .. stn.fc3``, ``conv1..3``,
``bn1..3``) so that the shipped ``pretrained_pcencoder_pointnet.pt`` (58 tensors) loads key
for key.  Only the global-feature configuration used by the reference entry points is
provided (``global_feat=True, feature_transform=False``); the unused classification /
segmentation heads of the upstream file are out of scope.

All layers are 1x1 convolutions = GEMMs over ``B*N`` points; they run on MFMA through the
ROCm libraries.  Pinned by goldens generated from the reference module itself
(``tests/golden/pointnet_*.npz``).
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

from .fused_bn import conv_bn_act, conv_bn_act_max


class _PointTrunk(nn.Module):
    """3 -> 64 -> 128 -> 1024 shared-MLP (1x1 conv + BN) followed by a max over points."""

    def _make_trunk(self):
        self.conv1 = nn.Conv1d(3, 64, 1)
        self.conv2 = nn.Conv1d(64, 128, 1)
        self.conv3 = nn.Conv1d(128, 1024, 1)
        self.bn1 = nn.BatchNorm1d(64)
        self.bn2 = nn.BatchNorm1d(128)
        self.bn3 = nn.BatchNorm1d(1024)


_EYE9: dict = {}


def _identity9(like: torch.Tensor) -> torch.Tensor:
    """The flattened 3x3 identity on ``like``'s device (built once per device and dtype: two launches per call otherwise)."""
    key = (like.device, like.dtype)
    eye = _EYE9.get(key)
    if eye is None:
        eye = torch.eye(3, dtype=like.dtype, device=like.device).reshape(1, 9)
        if not (like.is_cuda and torch.cuda.is_current_stream_capturing()):      # a capture's allocations belong to its graph
            _EYE9[key] = eye
    return eye


class STN3d(_PointTrunk):
    """Predicts a 3x3 alignment matrix per cloud (identity + learned residual)."""

    def __init__(self):
        super().__init__()
        self._make_trunk()
        self.fc1 = nn.Linear(1024, 512)
        self.fc2 = nn.Linear(512, 256)
        self.fc3 = nn.Linear(256, 9)
        self.relu = nn.ReLU()  # parameter-free; present in the reference module
        self.bn4 = nn.BatchNorm1d(512)
        self.bn5 = nn.BatchNorm1d(256)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        h = conv_bn_act(self.conv1, self.bn1, x, "relu")
        h = conv_bn_act(self.conv2, self.bn2, h, "relu")
        h = conv_bn_act_max(self.conv3, self.bn3, h, "relu")     # BatchNorm + ReLU + max over points (K5)
        h = F.relu(self.bn4(self.fc1(h)))
        h = F.relu(self.bn5(self.fc2(h)))
        h = self.fc3(h)
        return (h + _identity9(h)).view(-1, 3, 3)


class PointNetfeat(_PointTrunk):
    """``forward(x[B,3,N]) -> (feat[B,1024], trans[B,3,3], None)``: aligned points through
    the shared MLP (no ReLU after the last BN) and a max over the N points."""

    def __init__(self, global_feat: bool = True, feature_transform: bool = False):
        super().__init__()
        if not global_feat or feature_transform:
            raise NotImplementedError(
                "only the global-feature PointNet used by FPSG's entry points is provided")
        self.stn = STN3d()
        self._make_trunk()
        self.global_feat = global_feat
        self.feature_transform = feature_transform

    def forward(self, x: torch.Tensor):
        trans = self.stn(x)
        # (x^T @ trans)^T == trans^T @ x : one small batched GEMM, no transposed copies
        h = torch.bmm(trans.transpose(1, 2), x)
        h = conv_bn_act(self.conv1, self.bn1, h, "relu")
        h = conv_bn_act(self.conv2, self.bn2, h, "relu")
        h = conv_bn_act_max(self.conv3, self.bn3, h, None)       # BatchNorm + max over points (K5)
        return h, trans, None
