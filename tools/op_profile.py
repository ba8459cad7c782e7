"""Which PyTorch operators the glue kernels of an episode step come from (torch.profiler).
    python tools/op_profile.py [workload] > gpurun_out/op_profile.txt"""
import sys
import torch
sys.path.insert(0, ".")
import bench  # noqa: E402
from fpsg_amd import gemm_tuning  # noqa: E402
from fpsg_amd.engine import TrainStep, build_model, build_optimizer, default_options  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
S, Q, intra, encoder, epr, _ = bench.WORKLOADS[wl]
gemm_tuning.enable()
opt = default_options(device="cuda", intra_recon=intra, pc_encoder=encoder, n_shot=S, n_query=Q)
torch.manual_seed(0)
model = build_model(opt).to("cuda").train()
optimizer, _ = build_optimizer(model, opt)
step = TrainStep(model, optimizer)
eps = bench.make_episodes(S, Q, 2, seed=1, device=torch.device("cuda"))
for _ in range(3):
    step(eps)
torch.cuda.synchronize()
from torch.profiler import ProfilerActivity, profile
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    step(eps)
    torch.cuda.synchronize()
print(prof.key_averages(group_by_input_shape=True).table(sort_by="self_cuda_time_total", row_limit=90,
                                                         max_name_column_width=48, max_shapes_column_width=70))
