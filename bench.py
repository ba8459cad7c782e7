#!/usr/bin/env python3
"""bench.py -- episodes/sec of the FPSG hot path + Chamfer-kernel roofline on MI355X.

Contract: ``python bench.py --gpus N --steps K --warmup W``.  For N>1 either launched by
``python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py
--gpus N ...`` or typed as is: without a torchrun environment (no WORLD_SIZE) bench.py starts that launcher itself
as a child process, before any GPU call, and relays rank 0's line and the exit code.
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
episode step on the host cores through the CPU oracle; N=1 only).  At N=1 the line also carries
  ``configs``   short legs of the other BASELINE workloads (c2, c3, c4: episodes/s each) and ``eval``: items/s of the
                evaluation loop (``_return_reconstruction``: forward + Chamfer + Sinkhorn-form EMD) on the configs[2]
                episode with K1's and K2b's event-timed shares,
  ``kernels``   event-timed rooflines of the other hand-written distance / graph kernels at the
                BASELINE shapes (K1 backward in-step; K2, K2b, K3, K4b as micro-legs), each with its
                bound (valu / mfma / hbm), algorithmic work and fraction of peak,
  ``mfu_direct_equivalent``  the step's FLOPs counted as the REFERENCE formulation would spend them
                (SURVEY.md 3.1 MAC counts, forward x 3) over the fp32 peak;
``--no-extra`` skips these legs.  At N>1 ``allreduce`` times each gradient bucket's collective.
"""
from __future__ import annotations

import argparse
import json
import os

os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")   # before the HIP runtime initialises (fpsg_amd/__init__.py)
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
# bytes (2*2048*12 B read + 2*2048*(4+4) B written) and N*M distances x 8 flop when one pass serves both sides
HBM_PEAK = 8.0e12                                # B/s  (MI355X_MICROARCH.md)
F32_PEAK = 157.3e12                              # FLOP/s, fp32 vector == fp32-input MFMA

# forward multiply-accumulates of the REFERENCE formulation (direct 3x3
# convolutions, repeat + concat decoder), per item
VGG_MAC_PER_IMAGE = 15.5e9
DECODER_MAC_PER_CLOUD = 2048 * 3.85e6
POINTNET_MAC_PER_CLOUD = 2048 * 0.28e6
DGCNN_MAC_PER_CLOUD = 0.27e12 / 64


def direct_equivalent_flops(S, Q, intra, encoder):
    """FLOPs of one episode step (forward + backward = 3 x forward MACs x 2) as the reference's
    own formulation would spend them."""
    images = S + Q
    clouds_dec = Q + (S if intra else 0)
    clouds_enc = 2 * S
    enc = DGCNN_MAC_PER_CLOUD if encoder == "dgcnn" else POINTNET_MAC_PER_CLOUD
    mac = images * VGG_MAC_PER_IMAGE + clouds_dec * DECODER_MAC_PER_CLOUD + clouds_enc * enc
    return 3.0 * 2.0 * mac


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
            flops += B * 1.0 * N * M * 8.0                # N*M distances EVALUATED (each serves both directions) x 8 flop
        return n, sec, pairs, nbytes, flops


def pmc_traffic(probe, kind):
    """HBM bytes per op from the committed rocprofv3 PMC summary (profiles/k1_traffic.json, made by
    tools/pmc_k1.sh + tools/pmc_summary.py on this workload's launches); None when a launch shape of
    this run has no PMC record."""
    path = os.path.join(ROOT, "profiles", "k1_traffic.json")
    if not os.path.exists(path):
        return None
    table = {rec["cloud_pairs"]: rec["hbm_bytes_per_op"] for rec in json.load(open(path)).get("ops", [])
             if rec.get("op") == kind and "hbm_bytes_per_op" in rec}
    total, n = 0.0, 0
    for (k, B, N, M), _, _ in probe.records:
        if k != kind:
            continue
        if N != 2048 or M != 2048 or B not in table:
            return None
        total += table[B]
        n += 1
    return total / n if n else None


_SCLK_FILE = None


def gpu_clock_mhz(device):
    """The shader clock level the card reports RIGHT NOW (`pp_dpm_sclk`'s starred line, MHz), or None when sysfs does not
    show it.  Called while a step's kernels are still executing (after its enqueue, before its synchronize), on the
    last warm-up step and on one extra untimed step behind the timed region -- never inside the timed region -- so that
    a bench line records the clock its box held under this load (box-to-box spread)."""
    global _SCLK_FILE
    import glob
    import re
    try:
        if _SCLK_FILE is None:
            cands = sorted(glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk"))
            want = None
            try:
                pr = torch.cuda.get_device_properties(device)
                want = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}"
            except Exception:
                pass
            pick = [c for c in cands if want and want in os.path.realpath(os.path.dirname(c))]
            _SCLK_FILE = (pick or cands or [""])[0]
        if not _SCLK_FILE:
            return None
        txt = open(_SCLK_FILE).read()
        m = re.search(r"(\d+)\s*[Mm][Hh]z\s*\*", txt)
        return int(m.group(1)) if m else None
    except Exception:
        return None


def make_episodes(S, Q, count, seed, device):
    return [synthetic_episode(S, Q, n_pts=2048, img_size=224, seed=seed * 1000 + i, device=device)
            for i in range(count)]


from fpsg_amd.cli import usable_cores  # noqa: E402  (affinity mask + cgroup quota, capped by FPSG_CPU_THREADS)


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
    t0 = time.perf_counter()
    step([ep])                                   # first call: thread pools, oneDNN primitive caches -- not timed
    warm = time.perf_counter() - t0
    print(f"[cpu_baseline] warm-up episode {warm:.1f} s on {cores} threads (not counted)", file=sys.stderr, flush=True)
    n, t0 = 0, time.perf_counter()
    while True:
        step([ep])
        n += 1
        el = time.perf_counter() - t0
        print(f"[cpu_baseline] {n} episode(s) in {el:.1f} s on {cores} threads", file=sys.stderr, flush=True)
        if el > budget_s or el + el / n > 1.3 * budget_s:
            break
    return {"value": n / el, "unit": "episodes/s", "cores": cores, "kind": "port",
            "sample": f"{n} episode step(s) of the same workload ({S}-shot {Q}-query"
                      f"{' intra_recon' if intra else ''}), PyTorch-CPU model + C-oracle Chamfer, "
                      f"{el:.1f} s after one untimed warm-up episode ({warm:.1f} s)"}


def _event_time(fn, reps, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / reps


def kernel_rooflines(device):
    """Event-timed micro-legs of the hand-written distance / graph kernels that the default (PointNet,
    Chamfer) step does not launch, at the BASELINE shapes; algorithmic work as in DESIGN.md section 3."""
    from fpsg_amd import _hip
    from fpsg_amd.dgcnn import _reverse_graph, knn_int32
    from fpsg_amd.metrics import emd_approx, sinkhorn_divergence, sinkhorn_epsilons
    out = {}
    N = 2048
    g = torch.Generator(device="cpu").manual_seed(7)
    x = (torch.rand(5, N, 3, generator=g) * 2 - 1).to(device)
    y = torch.tanh(torch.randn(5, N, 3, generator=g)).to(device)

    def entry(bound, achieved, peak, unit, t, shape, work, **more):
        return {"bound": bound, "achieved": achieved, "peak": peak, "unit": unit, "frac": achieved / peak,
                "us": t * 1e6, "shape": shape, "work": work, **more}

    # K1 sweep: N = M = 2048, B in {1, 5, 32, 37, 256, 2048}, forward and backward as the product path
    # runs them (metrics._sided_forward / _sided_backward: the two-pass kernel below ~7 pairs, the one-pass tiled form
    # above), event-timed back to back.  frac by EXECUTED work (one-pass: N*M distances per pair serve both directions;
    # two-pass: 2*N*M), hbm_GBps by algorithmic bytes (81,920 B forward, 131,072 B backward per pair).  B = 2048 is the
    # kernel's own ceiling; the in-step B = 37 figure stays the headline `roofline`.
    from fpsg_amd import metrics as _m
    lib = _hip.load()
    for B1 in (1, 5, 32, 37, 256, 2048):
        p1 = (torch.rand(B1, N, 3, generator=g) * 2 - 1).to(device)
        p2 = torch.tanh(torch.randn(B1, N, 3, generator=g)).to(device)
        one_pass = lib.fpsg_chamfer_workspace_bytes(B1, N, N, -1) > 0 and _m._tiled_enabled()
        reps = 200 if B1 <= 256 else 20
        # (the median of three rounds: one-off stalls of a first-seen size -- allocator, code-object load -- landed in single
        # rounds of this leg: 385 us where the steady state is 31)
        t = sorted(_event_time(lambda: _m._sided_forward(p1, p2), reps, warm=20) for _ in range(3))[1]
        flop = B1 * float(N) * N * 8 * (1 if one_pass else 2)
        out[f"K1_chamfer_fwd_B{B1}"] = entry(
            "valu", flop / t / 1e12, F32_PEAK / 1e12, "TFLOP/s", t, f"B={B1} N=M=2048",
            ("one-pass tiled form: N*M distances per pair x 8 flop, tiles + finalize launches" if one_pass else
             "two-pass form: 2*N*M distances per pair x 8 flop, one launch"),
            hbm_GBps=B1 * 81920.0 / t / 1e9, hbm_frac=B1 * 81920.0 / t / HBM_PEAK,
            frac_2NM_convention=B1 * 2.0 * N * N * 8 / t / F32_PEAK)
        d1, d2, i1, i2 = _m._sided_forward(p1, p2)
        g1, g2 = torch.randn(B1, N, generator=g).to(device), torch.randn(B1, N, generator=g).to(device)
        t = sorted(_event_time(lambda: _m._sided_backward(p1, p2, i1, i2, g1, g2), reps, warm=20) for _ in range(3))[1]
        out[f"K1_chamfer_bwd_B{B1}"] = entry(
            "hbm", B1 * 131072.0 / t / 1e9, HBM_PEAK / 1e9, "GB/s", t, f"B={B1} N=M=2048",
            "algorithmic bytes: idx + grad read, clouds re-read, gradients written (131,072 B per pair); in-degree "
            "gathers are served by L2")
        del p1, p2, d1, d2, i1, i2, g1, g2
    # K2: approximate-assignment EMD, forward (evaluation): 10 levels x 3 sweeps of N*M pairs; a pair of the
    # assignment sweep = distance (8 flop) + exp + sqrt + 4 mul/add, of the two normaliser sweeps = 8 + exp + 3
    t = _event_time(lambda: emd_approx(x, y), 20)
    flop = 5 * N * N * 10.0 * (2 * 12 + 14)
    out["K2_emd_approx_fwd"] = entry("valu", flop / t / 1e12, F32_PEAK / 1e12, "TFLOP/s", t, "B=5 N=M=2048",
                                     "30 sweeps x N*M pairs x 12-14 flop (v_exp_f32 / v_sqrt_f32 counted as 1)",
                                     pair_sweeps_per_s=5 * 30.0 * N * N / t)
    # K2b: Sinkhorn divergence (what emd_wrapper calls): (n_eps + 2) steps x 4 soft-mins x N*M pairs; a pair =
    # distance (8 flop) + fma + max + sub + exp + add = 13
    n_eps = len(sinkhorn_epsilons(3.5))
    t = _event_time(lambda: sinkhorn_divergence(x, y, diameter=3.5), 20)
    flop = 5 * 4.0 * (n_eps + 2) * N * N * 13.0
    out["K2b_sinkhorn_divergence"] = entry("valu", flop / t / 1e12, F32_PEAK / 1e12, "TFLOP/s", t,
                                           f"B=5 N=M=2048, {n_eps} epsilons",
                                           f"{4 * (n_eps + 2)} soft-mins x N*M pairs x 13 flop (v_exp_f32 counted as 1)",
                                           launches=n_eps + 3)
    # K3 / K4b at DGCNN's layer shapes: B = 64 clouds (2 x 32-shot), N = 2048, k = 20
    B, k = 64, 20
    for C, Co in ((3, 64), (64, 64), (64, 128), (128, 256)):
        xc = torch.randn(B, C, N, generator=g).to(device)
        t = _event_time(lambda: knn_int32(xc, k), 10)
        flop = B * 2.0 * C * N * N
        if f"K3_knn_C{C}" in out:
            pass             # layers 2 and 3 share the 64-channel graph shape
        elif C >= 64:        # the x^T x term is a real GEMM on fp32-input MFMA
            out[f"K3_knn_C{C}"] = entry(
                "mfma", flop / t / 1e12, F32_PEAK / 1e12, "TFLOP/s", t, f"B=64 C={C} N=2048 k=20",
                "2*C*N*N flop per cloud (x^T x) + top-20 selection", pair_scores_per_s=B * float(N) * N / t)
        else:            # 3 channels: the top-k selection rounds bind (VALU / DPP)
            out["K3_knn_C3"] = entry("valu", B * float(N) * N * 8 / t / 1e12, F32_PEAK / 1e12, "TFLOP/s", t,
                                     "B=64 C=3 N=2048 k=20", "N*N pair scores x 8 flop + top-20 selection",
                                     pair_scores_per_s=B * float(N) * N / t)
        # K4b: the fused EdgeConv kernels themselves (C ABI), on this layer's PQ = x^T [W1 ; W2-W1]^T
        lib = _hip.load()
        idx = knn_int32(xc, k)
        PQ = torch.randn(B, N, 2 * Co, generator=g).to(device)
        sgn = torch.where(torch.randn(Co, generator=g) < 0, -1.0, 1.0).to(device)
        ysel = torch.empty(B, N, Co, device=device)
        jsel = torch.empty(B, N, Co, dtype=torch.uint8, device=device)
        s1 = torch.empty(B, N, Co, device=device)
        part = torch.empty(lib.fpsg_edgeconv_blocks(B, N, Co), 2, Co, device=device)
        st = torch.cuda.current_stream().cuda_stream
        tf = _event_time(lambda: lib.fpsg_edgeconv_fwd(PQ.data_ptr(), idx.data_ptr(), sgn.data_ptr(), B, N, k, Co,
                                                       ysel.data_ptr(), jsel.data_ptr(), s1.data_ptr(), part.data_ptr(), st), 20)
        rev, off = _reverse_graph(idx)
        dzs = torch.randn(B, N, Co, generator=g).to(device)
        coef = torch.randn(3, Co, generator=g).to(device)
        dPQ = torch.empty_like(PQ)
        tb = _event_time(lambda: lib.fpsg_edgeconv_bwd(dzs.data_ptr(), jsel.data_ptr(), PQ.data_ptr(), s1.data_ptr(),
                                                       rev.data_ptr(), off.data_ptr(), coef.data_ptr(), B, N, k, Co,
                                                       dPQ.data_ptr(), st), 20)
        # compulsory HBM bytes: forward reads PQ [B,N,2Co] and idx, writes ysel (4 B) + jsel (1 B) + s1 (4 B) per
        # [B,N,Co]; backward reads dzs, jsel, s1, PQ, the reverse graph (rev + idx-sized offsets) and writes dPQ
        fb = B * N * (2 * Co * 4 + k * 4 + Co * 9)
        bb = B * N * (Co * 4 + Co + Co * 4 + 2 * Co * 4 + k * 4 + 4 + 2 * Co * 4)
        key = f"K4b_edgeconv_{C}to{Co}"
        out[key + "_fwd"] = entry("hbm", fb / tf / 1e9, HBM_PEAK / 1e9, "GB/s", tf, f"B=64 N=2048 k=20 Co={Co}",
                                  "compulsory bytes: PQ + idx read, ysel / jsel / s1 written; the k-fold neighbour gather "
                                  "is served by L2", gathered_GBps=B * N * k * Co * 4 / tf / 1e9)
        out[key + "_bwd"] = entry("hbm", bb / tb / 1e9, HBM_PEAK / 1e9, "GB/s", tb, f"B=64 N=2048 k=20 Co={Co}",
                                  "compulsory bytes: dzs, jsel, s1, PQ, reverse graph read, dPQ written; in-edge gathers "
                                  "are served by L2")
    out.update(trunk_kernel_rooflines(device, entry))
    return out


def trunk_kernel_rooflines(device, entry):
    """The hand-written kernels of the image branch and the optimizer at the step's shapes (37 images): K5 / K6
    transforms / K7 / K8 / K8f against the HBM peak with their algorithmic bytes, K6f against
    the fp32 MFMA peak in the Winograd domain."""
    import torch.nn as nn
    from fpsg_amd import winograd as wg
    from fpsg_amd.conv_first import _forward as first_fwd, conv3x3_first
    from fpsg_amd.fused_bn import bn_act
    from fpsg_amd.optim import FlatAdam
    out = {}
    n_img = 37
    hbm = lambda nbytes, t, shape, work, **more: entry("hbm", nbytes / t / 1e9, HBM_PEAK / 1e9, "GB/s", t, shape, work, **more)
    # K6f: both 64-input-channel layers; flop in the Winograd domain (36 products of [K x 64] . [64 x tiles])
    for K, H in ((64, 224), (128, 112)):
        x = torch.randn(n_img, 64, H, H, device=device)
        U = wg._filter(4, torch.randn(K, 64, 3, 3, device=device) * 0.05, False)
        t = _event_time(lambda: wg._fused(x, U), 10)
        flop = 2.0 * 36 * K * 64 * n_img * (H // 4) ** 2
        out[f"K6f_fused_conv_64to{K}"] = entry(
            "mfma", flop / t / 1e12, F32_PEAK / 1e12, "TFLOP/s", t, f"({n_img},64,{H},{H})",
            "36 x [K x 64].[64 x tiles] products (Winograd domain; x4 = direct-equivalent); fp32 MFMA and fp32 VALU "
            "do not overlap on gfx950, so the transform's vector instructions add to the MFMA time (DESIGN.md K6f)",
            hbm_GBps=(x.numel() + n_img * K * H * H) * 4 / t / 1e9)
        gy = torch.randn(n_img, K, H, H, device=device)
        t = _event_time(lambda: wg._fused_dw(x, None, None, gy), 10)
        out[f"K6w_fused_dw_64to{K}"] = entry(
            "mfma", flop / t / 1e12, F32_PEAK / 1e12, "TFLOP/s", t, f"x ({n_img},64,{H},{H}), dy {K} channels",
            "the same 36 products reduced over the tiles, both operands transformed in registers (two transforms per "
            "lane and step add to the MFMA time)")
        del x, U, gy
    # K6 transforms at the largest three-kernel layer (128 channels @112): (1 + 2.25) x the activation tensor
    x = torch.randn(n_img, 128, 112, 112, device=device)
    V = wg._input(4, x)
    nb = 3.25 * x.numel() * 4
    for name, fn in (("input", lambda: wg._input(4, x)), ("output", lambda: wg._output(4, V, n_img, 112, 112)),
                     ("grad_output", lambda: wg._grad_output(4, x))):
        t = _event_time(fn, 10)
        out[f"K6_{name}_transform"] = hbm(nb, t, f"({n_img},128,112,112) m=4", "1 read + 2.25 writes (or the reverse) of the tensor")
    del V
    # K6g: the 36 batched products of the layers that stay on the three-kernel form, as the step runs them
    # (torch.bmm = the library's fp32-MFMA batched GEMM): forward U.V and the weight gradient gM.V^T
    for C, K, H in ((128, 128, 112), (256, 256, 56), (512, 512, 28)):
        P = n_img * (H // 4) ** 2
        Ps = wg.row_stride(P)        # the step's tensors: rows padded to whole 128-byte lines, zero pad columns
        U = torch.randn(36, K, C, device=device)
        V = torch.randn(36, C, Ps, device=device)
        gM = torch.randn(36, K, Ps, device=device)
        V[:, :, P:] = 0
        gM[:, :, P:] = 0
        flop = 2.0 * 36 * K * C * P      # the P real tiles; the products run over the Ps columns
        for leg, fn in (("fwd", lambda: torch.bmm(U, V)), ("dw", lambda: torch.bmm(gM, V.transpose(1, 2)))):
            t = _event_time(fn, 10)
            out[f"K6g_batched_gemm_{C}to{K}_{leg}"] = entry(
                "mfma", flop / t / 1e12, F32_PEAK / 1e12, "TFLOP/s", t, f"36 x [{K} x {C}].[{C} x {P}] (row stride {Ps})",
                "library GEMM (hipBLASLt / rocBLAS through torch.bmm, tuning records of fpsg_amd/tuning); "
                "Winograd-domain flop", hbm_GBps=(U.numel() + V.numel() + gM.numel()) * 4 / t / 1e9)
        del U, V, gM
    # K5: BatchNorm + ReLU on the same tensor: forward 2 reads + 1 write, backward 4 reads + 1 write
    bn = nn.BatchNorm2d(128).to(device).train()
    xr = x.requires_grad_()
    g = torch.randn_like(x)
    t = _event_time(lambda: bn_act(bn, xr, "relu"), 10)
    out["K5_bn_relu_fwd"] = hbm(3.0 * x.numel() * 4, t, f"({n_img},128,112,112)", "statistics pass + apply pass: 2R + 1W")
    y = bn_act(bn, xr, "relu")
    t = _event_time(lambda: torch.autograd.grad(y, xr, g, retain_graph=True), 10)
    out["K5_bn_relu_bwd"] = hbm(5.0 * x.numel() * 4, t, f"({n_img},128,112,112)", "sums pass (x, dy) + dx pass (x, dy -> dx): 4R + 1W")
    del x, xr, g, y
    # K8f / K8: the first convolution (3 -> 64 @224): forward bound by the write of y, weight gradient by the read of dy
    x = torch.randn(n_img, 3, 224, 224, device=device)
    w = (torch.randn(64, 3, 3, 3, device=device) * 0.2).requires_grad_()
    b = torch.zeros(64, device=device)
    t = _event_time(lambda: first_fwd(x, w.detach(), b, True), 10)
    ybytes = n_img * 64 * 224 * 224 * 4
    out["K8f_first_conv_fwd"] = hbm(ybytes + x.numel() * 4, t, f"({n_img},3,224,224) -> 64 channels",
                                    "x read, y written once; BatchNorm partial sums in the epilogue")
    yv = conv3x3_first(x, w)
    gy = torch.randn_like(yv)
    t = _event_time(lambda: torch.autograd.grad(yv, w, gy, retain_graph=True), 10)
    out["K8_first_conv_dw"] = hbm(ybytes + x.numel() * 4, t, f"dy ({n_img},64,224,224)", "dy and x read once")
    del x, yv, gy
    # K7: Adam over the model's 77.4 M parameters: 4 reads + 3 writes
    n = 77445125
    p = nn.Parameter(torch.randn(n, device=device))
    opt = FlatAdam([p], lr=1e-3)
    p.grad = torch.randn(n, device=device)
    opt.bind_gradients(p.grad)
    t = _event_time(lambda: opt.step(), 10)
    out["K7_adam_step"] = hbm(28.0 * n, t, f"{n} parameters", "param, grad, exp_avg, exp_avg_sq read; param and both moments written")
    del p, opt
    torch.cuda.empty_cache()
    return out


def split_gemm_error(device):
    """K10 and the library fp32 GEMM against a float64 product of the same operands at the 256 -> 256 @56 products of the
    step (three of the 36 batch entries): max and median error over the product's RMS, and K10's time against the
    library's (tuned record) on this box."""
    from fpsg_amd.gemm_split import bmm_split
    g = torch.Generator(device="cpu").manual_seed(11)
    from fpsg_amd.winograd import row_stride
    P = 37 * 196
    U = torch.randn(36, 256, 256, generator=g).to(device)
    V = torch.zeros(36, 256, row_stride(P), device=device)         # the step's layout: rows padded to whole lines, zero pads
    V[:, :, :P] = torch.randn(36, 256, P, generator=g).to(device)
    ref = torch.bmm(U[:3].double(), V[:3].double())
    scale = ref.pow(2).mean().sqrt()

    def err(C):
        d = (C[:3].double() - ref).abs()
        return float(d.max() / scale), float(d.flatten()[::7].median() / scale)

    e_lib, e_split = err(torch.bmm(U, V)), err(bmm_split(U, V))
    t_lib = _event_time(lambda: torch.bmm(U, V), 10)
    t_split = _event_time(lambda: bmm_split(U, V), 10)
    return {"shape": f"36 x [256x256].[256x{P}] (row stride {V.shape[2]})", "library_fp32_max": e_lib[0], "library_fp32_median": e_lib[1],
            "split_max": e_split[0], "split_median": e_split[1], "max_ratio": e_split[0] / e_lib[0],
            "median_ratio": e_split[1] / e_lib[1], "library_us": t_lib * 1e6, "split_us": t_split * 1e6,
            "speedup": t_lib / t_split,
            "gate": "VERDICT r4 item 1: >= 1.6x and error <= 1.5x the library's; the error bound holds, the speed-up "
                    "does not (DESIGN.md K10): wired as an opt-in, the headline stays on the fp32 MFMA"}


def eval_leg(device, probe, items=20, warmup=3):
    """The evaluation hot loop (reference src/evaluate_Network.py:107-118 around few_shot.py:131-176) on the configs[2]
    episode: eval mode, no_grad, per item ``_return_reconstruction`` (image + point encoders, query decode, K1 Chamfer,
    K2b Sinkhorn-form EMD = what ``emd_wrapper`` calls) and the loop's two ``.item()`` reads.  items/s is wall clock over
    the loop; K1's and K2b's shares come from HIP events around their C calls."""
    S, Q = 32, 5
    opt = default_options(device="cuda", intra_recon=True, pc_encoder="pointnet", n_shot=S, n_query=Q)
    torch.manual_seed(0)
    model = build_model(opt).to(device).eval()
    eps = make_episodes(S, Q, 4, seed=77, device=device)
    keep, keep_enabled = list(probe.records), probe.enabled
    probe.records, probe.enabled = [], False
    from fpsg_amd.engine import EvalItem
    with EvalItem(model) as run_item:                      # as evaluate_Network.main runs its loop
        for i in range(max(warmup, 4)):                    # (two eager items, the capture, one replay)
            out = run_item(eps[i % len(eps)])
            out["cd_loss"].item(), out["emd_loss"].item()
        torch.cuda.synchronize()
        probe.enabled = True
        t0 = time.perf_counter()
        for i in range(items):
            out = run_item(eps[i % len(eps)])
            cd, emd = out["cd_loss"].item() / Q, out["emd_loss"].item() / Q
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        graphed = bool(run_item._graphs)
    probe.enabled = False
    sec = {}
    for (k, B, N, M), e0, e1 in probe.records:
        sec[k] = sec.get(k, 0.0) + e0.elapsed_time(e1) * 1e-3
    probe.records, probe.enabled = keep, keep_enabled
    del model
    torch.cuda.empty_cache()
    return {"items_per_s": items / el, "ms_per_item": el / items * 1e3, "items": items,
            "workload": "evaluation loop, configs[2] episode: 32-shot 5-query PointNet, eval mode, no_grad; per item "
                        "_return_reconstruction (5 query images + 32 support clouds encoded -- the reference also encodes "
                        "the 32 ad images and 32 ad clouds and drops their features unused; in eval mode they do not "
                        "influence the outputs --, 5 query clouds decoded, Chamfer K1 + Sinkhorn-form EMD K2b on "
                        "B=5 x 2048 x 2048) + two .item() reads; FPSG_EVAL_PRUNE=0: the reference's full forward",
            "eval_prune": os.environ.get("FPSG_EVAL_PRUNE", "1") != "0",
            "hip_graph": graphed,
            # (inside the replayed graph the Chamfer call cannot be bracketed by events: null there; the Sinkhorn form runs
            # eagerly behind the graph)
            "k1_chamfer_us_per_item": sec["chamfer_fwd"] / items * 1e6 if "chamfer_fwd" in sec else None,
            "k2b_sinkhorn_us_per_item": sec.get("sinkhorn", 0.0) / items * 1e6,
            "k2b_share_of_item": sec.get("sinkhorn", 0.0) / el,
            "last_cd_per_query": cd, "last_emd_per_query": emd}


def allreduce_probe(step, world):
    """Per-bucket all-reduce of the flat gradient buffer, one collective at a time (event-timed on the
    current stream after a barrier): bucket bytes, time, algorithmic and bus bandwidth."""
    if not torch.distributed.is_initialized():
        return None
    res = []
    for (s, e) in step.buckets.buckets:
        buf = step.buckets.flat[s:e]
        for _ in range(2):
            torch.distributed.all_reduce(buf)
        torch.cuda.synchronize()
        torch.distributed.barrier()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            torch.distributed.all_reduce(buf)
        e1.record()
        torch.cuda.synchronize()
        t = e0.elapsed_time(e1) * 1e-3 / 5
        nbytes = (e - s) * 4
        res.append({"mb": nbytes / 1e6, "ms": t * 1e3, "algbw_GBps": nbytes / t / 1e9,
                    "busbw_GBps": nbytes / t / 1e9 * 2 * (world - 1) / max(world, 1)})
    step.buckets.flat.zero_()
    return res


def run_workload(wl, args, rank, world, device, steps, warmup, probe=None, graph=None, epr=None):
    """Builds the model of workload ``wl`` and times ``steps`` optimizer steps after ``warmup`` (the contract's
    timed region: barrier + synchronize on both sides, MAX over ranks).  Returns a dict of results."""
    S, Q, intra, encoder, epr_default, desc = WORKLOADS[wl]
    epr = epr or epr_default
    if graph is None:
        graph = wl == "c2" and not torch.distributed.is_initialized()
    if graph:
        warmup = max(warmup, 1 if epr >= 4 else 4)   # 2 eager uses + the capture before timing
    opt = default_options(device="cuda", intra_recon=intra, pc_encoder=encoder, n_shot=S, n_query=Q)
    torch.manual_seed(0)                      # identical initial weights on every rank
    model = build_model(opt).to(device)
    if args.channels_last:
        model.img_encoder.to(memory_format=torch.channels_last)
    model.train()
    model.overlap_encoders = bool(args.overlap)
    optimizer, _ = build_optimizer(model, opt)
    step = TrainStep(model, optimizer, world=world, bucket_mb=args.bucket_mb, graph=graph)
    episodes = make_episodes(S, Q, epr, seed=1234 + rank, device=device)   # resident in HBM
    if args.channels_last:
        for ep in episodes:
            for key in ("xs", "xq", "xad"):
                ep[key] = ep[key].squeeze(0).contiguous(memory_format=torch.channels_last).unsqueeze(0)

    def barrier():
        if world > 1:
            torch.distributed.barrier()

    for _ in range(warmup):
        step(episodes, n_episodes_global=epr * world)
    sclk_before = gpu_clock_mhz(device)       # the last warm-up step is still executing
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    if probe is not None:
        probe.enabled = not graph
    t0 = time.perf_counter()
    for _ in range(steps):
        out = step(episodes, n_episodes_global=epr * world)
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    sclk_after = None
    if probe is not None and world == 1:      # the headline run on one GPU: one extra untimed step, clock read under it
        was = probe.enabled
        probe.enabled = False
        step(episodes, n_episodes_global=epr * world)
        sclk_after = gpu_clock_mhz(device)
        torch.cuda.synchronize()
        probe.enabled = was
    if probe is not None and graph:
        # HIP events cannot be timed inside a captured graph: the K1 launches are bracketed in
        # an eager replica of the same episodes (same tensors, same neighbouring kernels) run
        # right after the timed region; profiles/ holds the rocprofv3 durations of the replays.
        probe.enabled = True
        for k, ep in enumerate(episodes):
            step._episode(ep, first=(k == 0))
        torch.cuda.synchronize()
    if probe is not None:
        probe.enabled = False
    t = torch.tensor([elapsed], dtype=torch.float64, device=device)
    ranks_seen = 1
    if world > 1:
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        one = torch.ones(1, dtype=torch.float32, device=device)        # every rank adds 1 through the same collective path
        torch.distributed.all_reduce(one)
        ranks_seen = int(one.item())
    elapsed = float(t.item())
    eps_per_s = steps * epr * world / elapsed
    flops = direct_equivalent_flops(S, Q, intra, encoder)
    return {"value": eps_per_s, "elapsed": elapsed, "steps": steps, "warmup": warmup, "epr": epr, "graph": bool(graph),
            "loss": float(out[-1]["ttl_loss"].sum().item()), "desc": desc, "S": S, "Q": Q, "intra": intra,
            "encoder": encoder, "params": sum(p.numel() for p in model.parameters()), "step": step,
            "ranks_seen": ranks_seen,
            "clock": {"sclk_mhz_under_last_warmup_step": sclk_before, "sclk_mhz_under_a_step_behind_the_timed_region": sclk_after,
                      "source": _SCLK_FILE or "unavailable",
                      "note": "pp_dpm_sclk's current level, read on the host while the step's kernels execute; outside the "
                              "timed region"},
            "mfu": {"direct_equivalent_tflop_per_episode": flops / 1e12,
                    "direct_equivalent_tflops": eps_per_s / world * flops / 1e12,
                    "fraction_of_fp32_peak": eps_per_s / world * flops / F32_PEAK,
                    "note": "FLOPs the REFERENCE formulation spends per episode step (SURVEY.md 3.1: direct 3x3 "
                            "convolutions, repeat+concat decoder; forward MACs x 3 x 2) times the measured episodes/s "
                            "per GPU, over the 157.3 TFLOP/s fp32 peak; above 1 is possible because Winograd F(4x4,3x3) "
                            "and the decoder's split first layer do the same arithmetic with 4x / ~100x fewer multiplies"}}


def self_launch(n_gpus):
    """``python bench.py --gpus N`` (N > 1) without a torchrun environment: runs
    ``python -m torch.distributed.run --nnodes=1 --nproc-per-node N --rdzv-backend c10d --rdzv-endpoint 127.0.0.1:0
    --local-addr 127.0.0.1 bench.py <the same arguments>`` as a child process, passes its stdout / stderr through (rank 0
    prints the one JSON line) and returns its exit code.  The rendezvous store binds port 0 itself (the kernel picks a
    free port while the socket is held: no window between choosing and using it, concurrent runs cannot collide); the
    workers get MASTER_ADDR / MASTER_PORT from it.  Called before anything touches the GPU."""
    import subprocess
    import uuid
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL's intra-node transport needs it on this driver
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or n_gpus) // n_gpus)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_gpus),
           "--rdzv-backend", "c10d", "--rdzv-endpoint", "127.0.0.1:0", "--rdzv-id", uuid.uuid4().hex,
           "--local-addr", "127.0.0.1", os.path.abspath(__file__)] + sys.argv[1:]
    print("[bench] no torchrun environment: launching " + " ".join(cmd), file=sys.stderr, flush=True)
    return subprocess.run(cmd, env=env, cwd=ROOT).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="c5")
    ap.add_argument("--episodes-per-rank", type=int, default=None)
    ap.add_argument("--cpu-baseline-seconds", type=float, default=15.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true",
                    help="skip the N=1 extra legs (other workloads' episodes/s, per-kernel rooflines)")
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
    # the driver's form: `python bench.py --gpus 1 --steps 20 --warmup 5`; the bare command is the same run
    steps = args.steps if args.steps is not None else 20
    warmup = args.warmup if args.warmup is not None else 5

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # typed as `python bench.py --gpus N` with no launcher around it: start one rank per GPU as CHILD processes of
        # torch.distributed.run and relay rank 0's line.  This process has made no HIP call up to here and makes none
        # afterwards (it never replaces itself either: an exec from a GPU-initialised process is what the pool forbids).
        raise SystemExit(self_launch(args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm GPU (the hot path has no CPU fallback)")
    rank, world, device = fdist.init_distributed("cuda")
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the launcher's --nproc-per-node and --gpus differ")
    torch.backends.cudnn.benchmark = bool(args.miopen_benchmark)
    if args.gemm_tuning is not None:
        os.environ["FPSG_GEMM_TUNING"] = {"off": "0"}.get(args.gemm_tuning, args.gemm_tuning)
    gemm_info = gemm_tuning.enable(path=args.gemm_records)

    probe = EventProbe()
    metrics.set_launch_probe(probe)
    main_run = run_workload(args.workload, args, rank, world, device, steps, warmup, probe=probe, graph=args.graph,
                            epr=epr)
    steps, warmup = main_run["steps"], main_run["warmup"]
    elapsed = main_run["elapsed"]
    allreduce = allreduce_probe(main_run["step"], world) if world > 1 or os.environ.get("FPSG_FORCE_DIST") else None

    n_l, sec, pairs, nbytes, flops = probe.summary("chamfer_fwd")
    nb_l, bsec, bpairs, _, _ = probe.summary("chamfer_bwd")

    if rank == 0:
        res = {
            "metric": "episodes/sec (+ Chamfer-kernel HBM GB/s in roofline.hbm), 2048-pt clouds, "
                      "224x224 images; episode = fwd + Chamfer + bwd, step = E episodes + all-reduce + Adam",
            "value": main_run["value"],
            "unit": "episodes/s",
            "n_gpus": world, "steps": steps, "warmup": warmup,
            "ms_per_step": elapsed / steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic (U(-1,1) images, unit-ball clouds), random-init weights",
            "config": {"workload": desc, "id": args.workload, "n_shot": S, "n_query": Q,
                       "intra_recon": intra, "pc_encoder": encoder, "episodes_per_rank_per_step": epr,
                       "episodes_per_step_global": epr * world, "parallelism": f"dp{world}",
                       "params": main_run["params"],
                       "hip_graph": main_run["graph"], **gemm_info},
            "final_loss": main_run["loss"],
            "hbm_peak_allocated_gb": round(torch.cuda.max_memory_allocated(device) / 1e9, 2),
            "mfu_direct_equivalent": main_run["mfu"],
            "clock": main_run["clock"],
        }
        if torch.distributed.is_initialized():
            # what the process group itself reports (N > 1: RCCL over xGMI), so the line shows that N ranks took part
            res["distributed"] = {"world_size": torch.distributed.get_world_size(),
                                  "backend": torch.distributed.get_backend(),
                                  "ranks_that_reported": main_run.get("ranks_seen")}
        if n_l:
            achieved_flops = flops / sec
            res["roofline"] = {
                "kernel": "K1 Chamfer forward of the timed steps: chamfer_tile_kernel + chamfer_finalize_kernel (two-sided "
                          "nearest neighbour + argmin, every pair distance evaluated once) + the loss sums' last stage "
                          "(chamfer_loss_reduce_kernel), one C call bracketed by HIP events",
                "bound": "valu",   # fp32 vector-ALU bound (no MFMA instruction in it); same 157.3 TFLOP/s peak
                "achieved": achieved_flops / 1e12, "peak": F32_PEAK / 1e12, "unit": "TFLOP/s",
                "frac": achieved_flops / F32_PEAK,
                "flop_convention": "EXECUTED work: N*M distances per cloud pair x 8 flop (one evaluation serves both "
                                   "directions; SURVEY.md 8(d)'s 4,194,304 pair evaluations 'if one fused pass yields "
                                   "both sides')",
                "frac_directed_pair_convention": 2.0 * achieved_flops / F32_PEAK,
                "directed_pair_convention": "SURVEY.md 8(d)'s other count, 2*N*M directed pair evaluations x 8 flop, which "
                                            "a two-pass kernel executes and this one does not",
                "traffic": pmc_traffic(probe, "chamfer_fwd"),
                "traffic_unit": "HBM bytes per op (tiles + finalize + sums), rocprofv3 PMC (2*FETCH_SIZE+WRITE_SIZE)*1024, "
                                "profiles/k1_traffic.json",
                "algorithmic_bytes_per_launch": nbytes / n_l,
                "launches": n_l, "avg_launch_us": sec / n_l * 1e6,
                "cloud_pairs_per_launch": pairs / n_l,
                "hbm": {"achieved": nbytes / sec / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                        "frac": nbytes / sec / HBM_PEAK,
                        "note": "algorithmic bytes (81,920 B per 2048x2048 cloud pair); the kernel "
                                "is fp32-VALU bound: arithmetic intensity ~410 flop/B vs ridge ~20"},
            }
            if nb_l:
                res["roofline"]["bwd_avg_launch_us"] = bsec / nb_l * 1e6
        kernels = {}
        if nb_l:
            # K1 backward in the step (fpsg_chamfer_bwd_losses): SURVEY.md 8(d)'s 131,072 algorithmic bytes per 2048x2048
            # cloud pair less the two upstream-gradient rows (16,384 B) that the per-pair constants replace; the kernel
            # is a latency chain (sort + runs), not a stream
            bb = bpairs * (131072.0 - 16384.0)
            kernels["K1_chamfer_bwd_in_step"] = {
                "bound": "hbm", "achieved": bb / bsec / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                "frac": bb / bsec / HBM_PEAK, "us": bsec / nb_l * 1e6, "shape": f"B={bpairs / nb_l:.0f} N=M=2048",
                "work": "114,688 algorithmic bytes per cloud pair (idx + clouds read, gradients written; the loss sums' per-pair "
                        "constants are formed in the kernel); one workgroup per (pair, side): LDS sort of the "
                        "argmin list + blocked ascending sums, time independent of the in-degree distribution"}
        if world == 1 and not args.no_extra:
            del main_run["step"]
            torch.cuda.empty_cache()
            configs = {}
            for wl in ("c2", "c3", "c4"):
                if wl == args.workload:
                    continue
                try:
                    _, _, _, _, e1, _ = WORKLOADS[wl]
                    r = run_workload(wl, args, rank, world, device, steps=10 if wl != "c2" else 30, warmup=3)
                    configs[wl] = {"episodes_per_s": r["value"], "ms_per_episode": 1e3 / r["value"],
                                   "workload": r["desc"], "hip_graph": r["graph"], "steps": r["steps"],
                                   "mfu_direct_equivalent": {k: v for k, v in r["mfu"].items() if k != "note"}}
                    del r
                    torch.cuda.empty_cache()
                except Exception as e:      # the headline numbers stay valid without a leg
                    configs[wl] = {"error": repr(e)}
            # K10 (opt-in, NOT the headline): the same c5 step with the trunk's transform-domain products on the bf16 matrix
            # pipe, fp32 operands split exactly into three bf16 pieces (FPSG_GEMM_SPLIT=1); its own error beside it
            if args.workload == "c5":
                try:
                    os.environ["FPSG_GEMM_SPLIT"] = "1"
                    r = run_workload("c5", args, rank, world, device, steps=10, warmup=3)
                    configs["c5_split"] = {
                        "episodes_per_s": r["value"], "ms_per_episode": 1e3 / r["value"], "steps": r["steps"],
                        "vs_headline": r["value"] / main_run["value"],
                        "arithmetic": "FPSG_GEMM_SPLIT=1: Winograd-domain products of the >= 128-channel layers and "
                                      "the three products of the decoder's wide layers (1539 -> 769, 769 -> 384) by "
                                      "fpsg_gemm_split -- fp32 operands split exactly into 3 x bf16, the six products of "
                                      "order <= 2^-18 on v_mfma_f32_32x32x16_bf16, fp32 accumulation; everything else as "
                                      "the headline (fp32 MFMA / VALU).  Opt-in: the headline line stays on the fp32 MFMA",
                        "final_loss": r["loss"], "gemm_error": split_gemm_error(device)}
                    del r
                except Exception as e:
                    configs["c5_split"] = {"error": repr(e)}
                finally:
                    os.environ.pop("FPSG_GEMM_SPLIT", None)
                    torch.cuda.empty_cache()
            try:
                configs["eval"] = eval_leg(device, probe)
            except Exception as e:
                configs["eval"] = {"error": repr(e)}
            res["configs"] = configs
            try:
                kernels.update(kernel_rooflines(device))
            except Exception as e:
                kernels["error"] = repr(e)
        if kernels:
            res["kernels"] = kernels
        if allreduce is not None:
            res["allreduce"] = {"buckets": allreduce, "note": "one blocking all_reduce per gradient bucket after the "
                                "timed region (in the step they are launched from autograd hooks and overlap backward)"}
        if world == 1 and not args.no_cpu_baseline:
            try:
                res["cpu_baseline"] = cpu_baseline(S, Q, intra, encoder, args.cpu_baseline_seconds)
            except Exception as e:  # the GPU numbers stay valid without the CPU leg
                res["cpu_baseline"] = {"error": repr(e)}
        print(json.dumps(res), flush=True)
    fdist.shutdown()


if __name__ == "__main__":
    main()
