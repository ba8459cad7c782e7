"""Which Python lines call a torch function during one episode step (forward call sites only; autograd's own calls do
not pass through the Python name):  python tools/who_calls.py [workload] [function ...]"""
import collections
import os
import sys
import traceback

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from fpsg_amd.engine import TrainStep, build_model, build_optimizer, default_options  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "c5"
names = sys.argv[2:] or ["cat", "stack", "zeros", "zeros_like", "empty_like", "sum", "mean", "clone"]
dev = torch.device("cuda:0")
S, Q, intra, encoder, epr, _ = bench.WORKLOADS[wl]
opt = default_options(device="cuda", intra_recon=intra, pc_encoder=encoder, n_shot=S, n_query=Q)
torch.manual_seed(0)
model = build_model(opt).to(dev).train()
optimizer, _ = build_optimizer(model, opt)
step = TrainStep(model, optimizer, world=1, graph=False)
episodes = bench.make_episodes(S, Q, 1, seed=1, device=dev)
for _ in range(2):
    step(episodes, n_episodes_global=1)
counts = collections.Counter()
originals = {}
for name in names:
    fn = getattr(torch, name)
    originals[name] = fn

    def wrapper(*a, _fn=fn, _name=name, **k):
        fr = traceback.extract_stack(limit=3)[0]
        counts[(_name, f"{os.path.basename(fr.filename)}:{fr.lineno}")] += 1
        return _fn(*a, **k)
    setattr(torch, name, wrapper)
step(episodes, n_episodes_global=1)
torch.cuda.synchronize()
for name, fn in originals.items():
    setattr(torch, name, fn)
for (name, where), n in sorted(counts.items(), key=lambda kv: (kv[0][0], -kv[1])):
    print(f"{n:4d}  torch.{name:12s} {where}")
