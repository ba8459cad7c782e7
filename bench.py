#!/usr/bin/env python3
"""bench.py -- episodes/sec of the FPSG hot path + Chamfer-kernel roofline on MI355X.

Contract: ``python bench.py --gpus N --steps K --warmup W`` (for N>1 launched by
``python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...``).
A *step* is one optimizer step: every rank runs ``--episodes-per-rank`` independent
few-shot episodes (forward, Chamfer loss, backward), gradients are all-reduced over RCCL,
Adam updates the 77 M parameters.  Weak scaling: per-rank work is fixed.

Workloads (BASELINE.json ``configs``; synthetic data, random-init weights, fp32):
  c5 (default)  configs[4]'s per-GPU share: 1-way 32-shot 5-query, --intra_recon, PointNet
                encoder, 8 episodes per rank per step (64 over 8 GPUs)
  c3            configs[2]: the same episode, 1 episode per step
  c2            configs[1]: 1-way 1-shot 1-query, 1 episode per step
  c4            configs[3]: c3 with the DGCNN encoder

Rank 0 prints ONE JSON line with ``roofline`` (the K1 Chamfer forward launches of the timed
region, bracketed with HIP events on the launch stream) and ``cpu_baseline`` (the same
episode step on the host cores through the CPU oracle; N=1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from fpsg_amd import dist as fdist  # noqa: E402
from fpsg_amd import gemm_tuning  # noqa: E402
from fpsg_amd import metrics  # noqa: E402
from fpsg_amd.engine import TrainStep, build_model, build_optimizer, default_options  # noqa: E402
from fpsg_amd.episodes import synthetic_episode  # noqa: E402

WORKLOADS = {
    #        S   Q  intra  encoder     episodes/rank  description
    "c5": (32, 5, True, "pointnet", 8,
           "configs[4] per-GPU share: 1-way 32-shot 5-query intra_recon PointNet, "
           "8 episodes/rank/step, 224x224 images, 2048-pt clouds"),
    "c3": (32, 5, True, "pointnet", 1,
           "configs[2]: 1-way 32-shot 5-query intra_recon PointNet, 224x224, 2048-pt"),
    "c2": (1, 1, False, "pointnet", 1,
           "configs[1]: 1-way 1-shot 1-query PointNet, 224x224, 2048-pt"),
    "c4": (32, 5, True, "dgcnn", 1,
           "configs[3]: 1-way 32-shot 5-query intra_recon DGCNN(k=20), 224x224, 2048-pt"),
}

# one two-sided Chamfer forward on a 2048-point cloud pair = 81,920 algorithmic
# bytes (2*2048*12 B read + 2*2048*(4+4) B written) and 2*N*M pair evaluations x 8 flop
HBM_PEAK = 8.0e12                                # B/s  (MI355X_MICROARCH.md)
F32_PEAK = 157.3e12                              # FLOP/s, fp32 vector == fp32-input MFMA


class EventProbe:
    """Brackets every K1 launch with HIP events on the stream it is enqueued on."""

    def __init__(self):
        self.records = []
        self.enabled = False

    class _Ctx:
        def __init__(self, owner, kind, B, N, M):
            self.o, self.meta = owner, (kind, B, N, M)

        def __enter__(self):
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e1 = torch.cuda.Event(enable_timing=True)
            self.e0.record()
            return self

        def __exit__(self, *exc):
            self.e1.record()
            self.o.records.append((self.meta, self.e0, self.e1))
            return False

    def __call__(self, kind, B, N, M):
        if not self.enabled:
            return metrics._NoProbe()
        return EventProbe._Ctx(self, kind, B, N, M)

    def summary(self, kind):
        """Launch-weighted totals for `kind` -> (launches, seconds, cloud_pairs, bytes, flops)."""
        n, sec, pairs, nbytes, flops = 0, 0.0, 0, 0.0, 0.0
        for (k, B, N, M), e0, e1 in self.records:
            if k != kind:
                continue
            n += 1
            sec += e0.elapsed_time(e1) * 1e-3
            pairs += B
            nbytes += B * ((N + M) * 12 + (N + M) * 8)   # read both clouds; write dist+idx
            flops += B * 2.0 * N * M * 8.0                # 2*N*M pair evaluations x 8 flop
        return n, sec, pairs, nbytes, flops


def pmc_traffic(kind_prefix, probe, kind):
    """HBM bytes per launch from the committed rocprofv3 PMC summary (profiles/k1_traffic.json,
    made by tools/pmc_summary.py on this workload's launches); None when a launch shape of this
    run has no PMC record."""
    path = os.path.join(ROOT, "profiles", "k1_traffic.json")
    if not os.path.exists(path):
        return None
    table = {}
    for rec in json.load(open(path)).get("launches", []):
        if rec["kernel"].startswith(kind_prefix) and "hbm_bytes_per_launch" in rec:
            table[rec["cloud_pairs"]] = rec["hbm_bytes_per_launch"]
    total, n = 0.0, 0
    for (k, B, N, M), _, _ in probe.records:
        if k != kind:
            continue
        if N != 2048 or M != 2048 or B not in table:
            return None
        total += table[B]
        n += 1
    return total / n if n else None


def make_episodes(S, Q, count, seed, device):
    return [synthetic_episode(S, Q, n_pts=2048, img_size=224, seed=seed * 1000 + i, device=device)
            for i in range(count)]


def usable_cores() -> int:
    """Cores this process may really use: affinity mask and cgroup CPU quota, not the host's
    core count (the GPU box exposes every host core but grants a 16-core share)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except AttributeError:
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, int(os.environ.get("FPSG_CPU_THREADS", "16"))))


def cpu_baseline(S, Q, intra, encoder, budget_s):
    """The same episode step (fwd + Chamfer + bwd + Adam) on the host cores: the model code
    in PyTorch-CPU with the C oracle's Chamfer.  The reference has no CPU path (hard
    .cuda()), so this is a port.  Bounded: repeats whole episodes until
    ``budget_s`` is used (at least one)."""
    import oracle
    if encoder != "pointnet":
        return None
    cores = usable_cores()
    torch.set_num_threads(cores)
    opt = default_options(device="cpu", intra_recon=intra, pc_encoder=encoder)
    torch.manual_seed(0)
    model = build_model(opt)
    model.pc_metric = oracle.make_torch_chamfer()
    optimizer, _ = build_optimizer(model, opt)
    step = TrainStep(model, optimizer)
    ep = synthetic_episode(S, Q, seed=99, device="cpu")
    n, t0 = 0, time.perf_counter()
    while True:
        step([ep])
        n += 1
        el = time.perf_counter() - t0
        print(f"[cpu_baseline] {n} episode(s) in {el:.1f} s on {cores} threads", file=sys.stderr, flush=True)
        if el > budget_s or el + el / n > 1.5 * budget_s:
            break
    return {"value": n / el, "unit": "episodes/s", "cores": cores, "kind": "port",
            "sample": f"{n} episode step(s) of the same workload ({S}-shot {Q}-query"
                      f"{' intra_recon' if intra else ''}), PyTorch-CPU model + C-oracle Chamfer, "
                      f"{el:.1f} s, includes first-call warm-up"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="c5")
    ap.add_argument("--episodes-per-rank", type=int, default=None)
    ap.add_argument("--cpu-baseline-seconds", type=float, default=20.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--bucket-mb", type=float, default=80.0)
    ap.add_argument("--graph", dest="graph", action="store_true", default=None,
                    help="replay each episode's fwd+bwd as a captured hipGraph [default: only for the "
                         "launch-bound 1-shot workload c2 on one GPU; the 32-shot steps are GPU-bound]")
    ap.add_argument("--no-graph", dest="graph", action="store_false")
    ap.add_argument("--overlap", dest="overlap", action="store_true", default=False,
                    help="run the point encoder on a second stream beside the image trunk (+2.5%% on c3)")
    ap.add_argument("--gemm-tuning", choices=["file", "off", "tune"], default=None,
                    help="library GEMM kernel selection (fpsg_amd.gemm_tuning): recorded choices from "
                         "fpsg_amd/tuning/gemm_gfx950.csv [default], the libraries' heuristic, or time "
                         "unknown shapes now and append them to --gemm-records")
    ap.add_argument("--gemm-records", default=None, help="records file for --gemm-tuning tune / file")
    ap.add_argument("--channels-last", action="store_true", help="experiment: NHWC image trunk")
    ap.add_argument("--miopen-benchmark", action="store_true", help="experiment: MIOpen find mode")
    args = ap.parse_args()

    S, Q, intra, encoder, epr, desc = WORKLOADS[args.workload]
    if args.episodes_per_rank:
        epr = args.episodes_per_rank
    steps = args.steps if args.steps is not None else (5 if epr > 1 else 20)
    warmup = args.warmup if args.warmup is not None else (2 if epr > 1 else 5)


    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm GPU (the hot path has no CPU fallback)")
    rank, world, device = fdist.init_distributed("cuda")
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if args.graph is None:
        args.graph = args.workload == "c2" and not torch.distributed.is_initialized()

    if args.graph:
        warmup = max(warmup, 1 if epr >= 4 else 4)   # 2 eager uses + the capture before timing
    torch.backends.cudnn.benchmark = bool(args.miopen_benchmark)
    if args.gemm_tuning is not None:
        os.environ["FPSG_GEMM_TUNING"] = {"off": "0"}.get(args.gemm_tuning, args.gemm_tuning)
    gemm_info = gemm_tuning.enable(path=args.gemm_records)
    opt = default_options(device="cuda", intra_recon=intra, pc_encoder=encoder, n_shot=S, n_query=Q)
    torch.manual_seed(0)                      # identical initial weights on every rank
    model = build_model(opt).to(device)
    if args.channels_last:
        model.img_encoder.to(memory_format=torch.channels_last)
    model.train()
    model.overlap_encoders = bool(args.overlap)
    optimizer, _ = build_optimizer(model, opt)
    step = TrainStep(model, optimizer, world=world, bucket_mb=args.bucket_mb, graph=args.graph)
    episodes = make_episodes(S, Q, epr, seed=1234 + rank, device=device)   # resident in HBM
    if args.channels_last:
        for ep in episodes:
            for key in ("xs", "xq", "xad"):
                ep[key] = ep[key].squeeze(0).contiguous(memory_format=torch.channels_last).unsqueeze(0)

    probe = EventProbe()
    metrics.set_launch_probe(probe)

    def barrier():
        if world > 1:
            torch.distributed.barrier()

    for _ in range(warmup):
        step(episodes, n_episodes_global=epr * world)
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    probe.enabled = not args.graph
    t0 = time.perf_counter()
    for _ in range(steps):
        out = step(episodes, n_episodes_global=epr * world)
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if args.graph:
        # HIP events cannot be timed inside a captured graph: the K1 launches are bracketed in
        # an eager replica of the same episodes (same tensors, same neighbouring kernels) run
        # right after the timed region; profiles/ holds the rocprofv3 durations of the replays.
        probe.enabled = True
        for k, ep in enumerate(episodes):
            step._episode(ep, first=(k == 0))
        torch.cuda.synchronize()
    probe.enabled = False

    t = torch.tensor([elapsed], dtype=torch.float64, device=device)
    if world > 1:
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
    elapsed = float(t.item())
    loss = float(out[-1]["ttl_loss"].sum().item())

    n_l, sec, pairs, nbytes, flops = probe.summary("chamfer_fwd")
    nb_l, bsec, _, _, _ = probe.summary("chamfer_bwd")

    if rank == 0:
        total_eps = steps * epr * world
        res = {
            "metric": "episodes/sec (+ Chamfer-kernel HBM GB/s in roofline.hbm), 2048-pt clouds, "
                      "224x224 images; episode = fwd + Chamfer + bwd, step = E episodes + all-reduce + Adam",
            "value": total_eps / elapsed,
            "unit": "episodes/s",
            "n_gpus": world, "steps": steps, "warmup": warmup,
            "ms_per_step": elapsed / steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic (U(-1,1) images, unit-ball clouds), random-init weights",
            "config": {"workload": desc, "id": args.workload, "n_shot": S, "n_query": Q,
                       "intra_recon": intra, "pc_encoder": encoder, "episodes_per_rank_per_step": epr,
                       "episodes_per_step_global": epr * world, "parallelism": f"dp{world}",
                       "params": sum(p.numel() for p in model.parameters()),
                       "hip_graph": bool(args.graph), **gemm_info},
            "final_loss": loss,
            "hbm_peak_allocated_gb": round(torch.cuda.max_memory_allocated(device) / 1e9, 2),
        }
        if n_l:
            achieved_flops = flops / sec
            res["roofline"] = {
                "kernel": "chamfer_fwd_kernel (K1, two-sided nearest neighbour + argmin)",
                "bound": "mfma",   # fp32 compute bound: fp32 vector peak == fp32-input MFMA peak
                "achieved": achieved_flops / 1e12, "peak": F32_PEAK / 1e12, "unit": "TFLOP/s",
                "frac": achieved_flops / F32_PEAK,
                "traffic": pmc_traffic("chamfer_fwd_kernel", probe, "chamfer_fwd"),
                "traffic_unit": "HBM bytes per launch, rocprofv3 PMC (2*FETCH_SIZE+WRITE_SIZE)*1024, "
                                "profiles/k1_traffic.json",
                "algorithmic_bytes_per_launch": nbytes / n_l,
                "launches": n_l, "avg_launch_us": sec / n_l * 1e6,
                "cloud_pairs_per_launch": pairs / n_l,
                "hbm": {"achieved": nbytes / sec / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                        "frac": nbytes / sec / HBM_PEAK,
                        "note": "algorithmic bytes (81,920 B per 2048x2048 cloud pair); the kernel "
                                "is fp32-VALU bound: arithmetic intensity ~820 flop/B vs ridge ~20"},
            }
            if nb_l:
                res["roofline"]["bwd_avg_launch_us"] = bsec / nb_l * 1e6
        if world == 1 and not args.no_cpu_baseline:
            try:
                res["cpu_baseline"] = cpu_baseline(S, Q, intra, encoder, args.cpu_baseline_seconds)
            except Exception as e:  # the GPU numbers stay valid without the CPU leg
                res["cpu_baseline"] = {"error": repr(e)}
        print(json.dumps(res), flush=True)
    fdist.shutdown()


if __name__ == "__main__":
    main()
