"""Where the entry point's loop loses time against bench.py's: the same c5-shaped steps (32-shot,
5-query, intra_recon, 8 episodes per step) timed with the episodes (a) already assembled (bench.py's condition),
(b) drawn from the resident corpora by the main thread, (c) by EpisodePrefetcher, (d) from host corpora with the
prefetcher -- plus the per-episode host time of drawing an episode.  Usage (GPU box): python tools/entry_vs_bench.py"""
import itertools
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fpsg_amd import cli, gemm_tuning  # noqa: E402
from fpsg_amd.engine import TrainStep, build_model, build_optimizer, default_options, to_device  # noqa: E402
from fpsg_amd.episodes import EpisodePrefetcher, SyntheticFewShot, collate_episode  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    print("host threads:", torch.get_num_threads(), "->", cli.limit_host_threads(), flush=True)
    gemm_tuning.enable()
    opt = default_options(device="cuda", intra_recon=True)
    model = build_model(opt).to(dev).train()
    optimizer, _ = build_optimizer(model, opt)
    step = TrainStep(model, optimizer)
    S, Q, E = 32, 5, 8
    steps = 6
    res = {}
    for where in ("resident", "host"):
        ds = SyntheticFewShot(n_classes=4, per_class=40, n_support=S, n_query=Q, device=dev if where == "resident" else "cpu")

        def draw(n):
            for _ in range(n):
                yield collate_episode(ds[int(torch.randint(len(ds), (1,)))])

        t0 = time.perf_counter()
        for _ in draw(16):
            pass
        torch.cuda.synchronize()
        res[f"{where}: host time to draw one episode (ms)"] = (time.perf_counter() - t0) / 16 * 1e3
        fixed = [to_device(ep, dev) for ep in draw(E)]
        modes = {"fixed episodes (bench.py)": None, "main thread draws": "main", "EpisodePrefetcher": "prefetch"}
        for name, mode in modes.items():
            if where == "host" and mode is None:
                continue
            for timed in (False, True):
                n = steps if timed else 2
                it = None
                if mode == "prefetch":
                    it = EpisodePrefetcher(draw(n * E), dev)
                elif mode == "main":
                    it = draw(n * E)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                sums = torch.zeros(2, dtype=torch.float64, device=dev)
                for _ in range(n):
                    local = fixed if it is None else [to_device(next(it), dev) for _ in range(E)]
                    for out in step(local, n_episodes_global=E):
                        sums[0] += out["query_rec_loss"].sum() / Q
                        sums[1] += out["support_rec_loss"].sum() / S
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
                if mode == "prefetch":
                    it.close()
            res[f"{where}: {name} (episodes/s)"] = n * E / dt
    for k, v in res.items():
        print(f"{k:60s} {v:8.2f}", flush=True)


if __name__ == "__main__":
    main()
