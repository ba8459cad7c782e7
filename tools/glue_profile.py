"""Which PyTorch ops make up the ~1.4 ms of small-kernel 'glue' of an episode step: torch.profiler over two c5
episodes, aten ops (GEMMs left out) grouped by input shapes (python stacks are not recorded on this build).
    python tools/glue_profile.py > gpurun_out/glue_profile.txt"""
import collections
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from fpsg_amd.engine import TrainStep, build_model, build_optimizer, default_options  # noqa: E402

dev = torch.device("cuda:0")
S, Q, intra, encoder, epr, _ = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "c5"]
opt = default_options(device="cuda", intra_recon=intra, pc_encoder=encoder, n_shot=S, n_query=Q)
torch.manual_seed(0)
model = build_model(opt).to(dev).train()
optimizer, _ = build_optimizer(model, opt)
step = TrainStep(model, optimizer, world=1, graph=False)
episodes = bench.make_episodes(S, Q, 2, seed=1, device=dev)
for _ in range(3):
    step(episodes, n_episodes_global=2)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    step(episodes, n_episodes_global=2)
    torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0.0])
for ev in prof.key_averages(group_by_input_shape=True):
    t = getattr(ev, "self_device_time_total", 0)
    if not t or not ev.key.startswith("aten::") or ev.key in ("aten::bmm", "aten::baddbmm", "aten::mm", "aten::addmm"):
        continue
    agg[(ev.key, str(ev.input_shapes)[:110])][0] += ev.count
    agg[(ev.key, str(ev.input_shapes)[:110])][1] += t
rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
print("# small aten ops of one 2-episode step by self device time: calls, us, op, input shapes")
for (name, shp), (n, t) in rows[:70]:
    print(f"{n:5d} {t:9.1f}  {name:28s} {shp}")
