"""``AuxClassifier``: the classification head used only by the encoder pre-training script
(reference ``src/models/support_models.py:6-30``).  The reference's two mask allocators in
the same file are never instantiated (``mask_learner`` is always ``None``) and are omitted."""
from __future__ import annotations

import torch.nn as nn
import torch.nn.functional as F


class AuxClassifier(nn.Module):
    """``1024 -> 512 -> 256 -> n_classes`` with BN, dropout(0.3) before the second BN, and
    log-softmax output; Xavier-normal weights."""

    def __init__(self, in_dim: int, out_dim: int, num_layer: int = 3):
        super().__init__()
        self.fc1 = nn.Linear(in_dim, 512)
        self.fc2 = nn.Linear(512, 256)
        self.fc3 = nn.Linear(256, out_dim)
        self.dropout = nn.Dropout(p=0.3)
        self.bn1 = nn.BatchNorm1d(512)
        self.bn2 = nn.BatchNorm1d(256)
        for fc in (self.fc1, self.fc2, self.fc3):
            nn.init.xavier_normal_(fc.weight)

    def forward(self, x):
        x = F.relu(self.bn1(self.fc1(x)))
        x = F.relu(self.bn2(self.dropout(self.fc2(x))))
        return F.log_softmax(self.fc3(x), dim=1)
