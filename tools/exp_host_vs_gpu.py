"""How far ahead of the GPU does the host run in the default bench step?  Times, per optimizer step, the host's
enqueue time (step() returns without a sync) against the time until the GPU has drained.  Run on the GPU box:
    python tools/exp_host_vs_gpu.py [workload]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from fpsg_amd import gemm_tuning  # noqa: E402
from fpsg_amd.engine import TrainStep, build_model, build_optimizer, default_options  # noqa: E402


def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else "c5"
    S, Q, intra, encoder, epr, _ = bench.WORKLOADS[wl]
    dev = torch.device("cuda:0")
    gemm_tuning.enable()
    opt = default_options(device="cuda", intra_recon=intra, pc_encoder=encoder, n_shot=S, n_query=Q)
    torch.manual_seed(0)
    model = build_model(opt).to(dev).train()
    model.overlap_encoders = True
    optimizer, _ = build_optimizer(model, opt)
    step = TrainStep(model, optimizer, world=1)
    eps = bench.make_episodes(S, Q, epr, seed=1234, device=dev)
    for _ in range(3):
        step(eps, n_episodes_global=epr)
    torch.cuda.synchronize()
    rows = []
    for _ in range(6):
        t0 = time.perf_counter()
        step(eps, n_episodes_global=epr)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        rows.append(((t1 - t0) * 1e3, (t2 - t0) * 1e3))
    for h, g in rows:
        print(f"host enqueue {h:7.1f} ms   until drained {g:7.1f} ms   host share {h / g:.2f}")
    # one episode at a time: host time of forward and backward separately
    sample = eps[0]
    for _ in range(2):
        t0 = time.perf_counter()
        out = model.loss(sample)
        t1 = time.perf_counter()
        out["ttl_loss"].sum().backward()
        t2 = time.perf_counter()
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        print(f"episode: host forward {1e3 * (t1 - t0):.2f} ms, host backward {1e3 * (t2 - t1):.2f} ms, drained after {1e3 * (t3 - t0):.2f} ms")


if __name__ == "__main__":
    main()
