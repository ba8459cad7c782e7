"""Times the in-edge-list construction of the EdgeConv backward at c4's shape (64 clouds x 2048 points x 20
neighbours): fpsg_edgeconv_reverse_graph against the stable torch sort it replaced."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fpsg_amd.dgcnn import _reverse_graph, _reverse_graph_sorted
idx = torch.randint(0, 2048, (64, 2048, 20), dtype=torch.int32, device="cuda")
for name, fn in (("kernel", _reverse_graph), ("torch sort", _reverse_graph_sorted)):
    for _ in range(5): fn(idx)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): fn(idx)
    e1.record(); torch.cuda.synchronize()
    print(name, e0.elapsed_time(e1) / 50 * 1e3, "us")
