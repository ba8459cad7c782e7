#!/usr/bin/env python3
"""Few-shot image -> point-cloud training on MI355X (entry point of the hot path).

Same command line, log lines and checkpoint files as reference ``src/trainNetwork.py``
(``main :67-208``); the per-step work is ``fpsg_amd.engine.TrainStep`` (HIP Chamfer loss,
PyTorch-ROCm networks).  Extras: ``--synthetic`` data, and data-parallel episodes when
launched with ``python -m torch.distributed.run --nproc-per-node N trainNetwork.py ...``
(every optimizer step then covers ``--episodes_per_step`` episodes, default one per rank;
gradients are averaged with an RCCL all-reduce; the BatchNorm buffers of the ranks are reconciled before an
evaluation or a save (``fpsg_amd.dist.BufferSync``); every rank evaluates a shard of the test items; rank 0 logs and
saves).

    python trainNetwork.py --synthetic --n_shot 1 --n_query 1 --epoch 2 --n_episode 10 \
        --pc_encoder_path tests/golden/pretrained_pcencoder_pointnet.pt --model_path /tmp/ckpt
"""
from __future__ import annotations

import itertools
import os

os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")   # before the HIP runtime initialises (fpsg_amd/__init__.py)
import statistics
import time
from collections import defaultdict

import torch

from fpsg_amd import cli, winograd
from fpsg_amd import dist as fdist
from fpsg_amd.engine import TrainStep, build_model, build_optimizer, to_device
from fpsg_amd.episodes import EpisodePrefetcher


def evaluate(model, dl_test, n_query, n_shot, device, log, rank: int = 0, world: int = 1):
    """Per-class Chamfer report (reference :157-189); returns the per-item values.

    With ``world`` ranks every rank evaluates the test items ``rank, rank + world, ...`` (every rank walks the same
    loader, so the items and their order are the single-process ones) and the values are gathered: no rank sits in a
    collective while another evaluates, and the report lines are those of one process."""
    model.eval()
    mine = []
    with torch.no_grad(), winograd.weights_frozen():     # no parameter changes while evaluating: filters / stacks made once
        for n, sample in enumerate(dl_test):
            if n % world != rank:
                continue
            out = model.loss(to_device(sample, device))
            mine.append((n, sample["class"][0], out["query_rec_loss"].item() / n_query))
    per_class = defaultdict(list)
    every = []
    for _, name, cd in sorted(itertools.chain.from_iterable(fdist.gather_objects(mine))):
        per_class[name].append(cd)
        every.append(cd)
    for name in sorted(per_class):
        vals = per_class[name]
        spread = statistics.stdev(vals) if len(vals) > 1 else 0.0
        log(f"Class: {name} -- Rec CD: {statistics.mean(vals)} ({spread})")
    model.train()
    return every


def main(opt):
    cli.validate(opt)
    n_query = opt.n_shot if opt.n_query == 0 else opt.n_query
    rank, world, device = fdist.init_distributed("cuda" if opt.device.startswith("cuda") else "cpu")
    cli.limit_host_threads(world)
    if world == 1:
        device = cli.pick_device(opt)
    elif device.type == "cuda":
        from fpsg_amd import gemm_tuning
        gemm_tuning.enable()
    is_main = rank == 0

    timestamp = time.strftime("%m_%d_%H_%M")
    checkpoint_path = os.path.join(opt.model_path, opt.name)
    checkpoint_imgs = os.path.join(checkpoint_path, "images")
    checkpoint_logs = os.path.join(checkpoint_path, f"log_{timestamp}.txt")
    if is_main:
        os.makedirs(checkpoint_imgs, exist_ok=True)

    ds, ds_test = cli.build_datasets(opt, n_query, device)
    dl, dl_test = cli.build_loaders(opt, ds, ds_test)

    model = build_model(opt)
    start_epoch = 1
    state_path = None
    if opt.resume > 0:
        if is_main:
            print(f"Resume previous training, start from epoch {opt.resume}, loading previous model")   # reference :106
        start_epoch = opt.resume
        resume_path = os.path.join(checkpoint_path, f"model_epoch_{start_epoch}.pt")
        # every rank reads the weights itself; all ranks learn whether all succeeded before anyone enters a collective
        err = None
        try:
            if not os.path.exists(resume_path):
                raise FileNotFoundError(f"{resume_path} does not exist, loading failed")
            model.load_state_dict(torch.load(resume_path, map_location="cpu", weights_only=True))
        except Exception as e:          # noqa: BLE001 -- reported on every rank by agree()
            err = f"{type(e).__name__}: {e}"
        fdist.agree(err, "resume (weights)")
        # Extension: the reference saves weights only, so a resumed run restarts
        # Adam's moments and the LR schedule.  A sidecar file keeps them; the weights file keeps
        # the reference's format and name.  Rank 0 decides whether the sidecar is used (ranks that saw different
        # directories would run different epoch counts and hang in the collectives).
        state_path = os.path.join(checkpoint_path, f"train_state_epoch_{start_epoch}.pt")
        if not fdist.broadcast_object(os.path.exists(state_path) if is_main else None):
            state_path = None
    model = model.to(device).train()
    fdist.broadcast_parameters(model)

    optimizer, scheduler = build_optimizer(model, opt)
    if state_path is not None:
        err = None
        try:
            state = torch.load(state_path, map_location=device, weights_only=True)
            optimizer.load_state_dict(state["optimizer"])
            scheduler.load_state_dict(state["scheduler"])
            # the sidecar holds the state AFTER epoch N's scheduler.step(): training continues with epoch
            # N+1 (the weights-only resume of the reference re-runs epoch N from fresh moments; doing that
            # here would step the LR schedule twice for epoch N and reuse post-epoch-N moments)
            start_epoch = int(state.get("epoch", start_epoch)) + 1
        except Exception as e:          # noqa: BLE001
            err = f"{type(e).__name__}: {e}"
        fdist.agree(err, "resume (optimizer / scheduler state)")
    start_epoch = int(fdist.broadcast_object(start_epoch))
    if opt.resume > 0 and is_main:
        how = (f"optimizer / scheduler state restored from {state_path}" if state_path is not None
               else "weights only, fresh optimizer state as in the reference")
        print(f"Resumed from model_epoch_{opt.resume}.pt ({how}): the next epoch is {start_epoch}")
    if start_epoch > opt.epoch:
        raise RuntimeError(f"nothing to do: the resumed run would start at epoch {start_epoch} but --epoch is "
                           f"{opt.epoch}")
    step = TrainStep(model, optimizer, world=world)
    buffers = fdist.BufferSync(model)
    eps_per_step = opt.episodes_per_step or world
    local_n = len(range(rank, eps_per_step, world))

    pending: list[str] = []

    def log(line: str) -> None:
        if is_main:
            print(line)
            pending.append(line)

    for epoch in range(start_epoch, opt.epoch + 1):
        # every rank draws its own episodes; different seeds per (epoch, rank)
        torch.manual_seed(1000003 * epoch + rank)
        sums = torch.zeros(2, dtype=torch.float64, device=device)
        n_steps = max(1, opt.n_episode // eps_per_step)
        # exactly the episodes this rank uses (the worker must not draw one more from the global RNG: the
        # evaluation below and the next epoch's seed share it), drawn and uploaded behind the step
        it = EpisodePrefetcher(itertools.islice(iter(dl), n_steps * local_n), device)
        t0 = time.perf_counter()
        try:
            for _ in range(n_steps):
                local = []
                for _ in range(local_n):
                    try:
                        local.append(to_device(next(it), device))
                    except StopIteration:
                        raise RuntimeError(
                            f"the loader ran out of episodes: --n_episode {opt.n_episode} gives this rank "
                            f"fewer than {n_steps} x {local_n} episodes (episodes_per_step {eps_per_step}, "
                            f"{world} rank(s))") from None
                for out in step(local, n_episodes_global=eps_per_step):
                    sums[0] += out["query_rec_loss"].sum() / n_query
                    sums[1] += out["support_rec_loss"].sum() / opt.n_shot
        finally:
            it.close()
        q_sum, s_sum = fdist.all_reduce_scalars(sums.tolist(), device)   # one host sync per epoch
        done = n_steps * eps_per_step
        dt = time.perf_counter() - t0
        log(f"Training Results for Epoch -- {epoch} are: Query_rec: {q_sum / done}, "
            f"Support_rec: {s_sum / done}")
        if is_main:
            print(f"  [{done / dt:.2f} episodes/s over {world} GPU(s)]")
        scheduler.step()

        evaluating = epoch % opt.eval_interval == 0 or epoch == opt.epoch
        saving = epoch % opt.save_interval == 0 or epoch == opt.epoch
        if evaluating or saving or epoch % opt.sample_interval == 0:
            buffers.sync()          # the BatchNorm running statistics of all ranks' episodes, as one process holds them
        if evaluating:
            if world > 1:
                torch.manual_seed(2000003 * epoch)      # every rank walks the same test items (the shards partition them)
            every = evaluate(model, dl_test, n_query, opt.n_shot, device, log, rank, world)
            spread = statistics.stdev(every) if len(every) > 1 else 0.0
            log(f"Avg testing results across all classes Epoch -- {epoch} are: "
                f"Query_rec: {sum(every) / max(len(every), 1)} ({spread})")
            for sample in dl_test if is_main else ():
                model.eval()
                model.draw_reconstruction(to_device(sample, device),
                                          os.path.join(checkpoint_imgs, f"sample_img_{epoch}_test.png"))
                model.train()
                break

        if is_main and saving:
            torch.save(model.state_dict(), os.path.join(checkpoint_path, f"model_epoch_{epoch}.pt"))
            torch.save({"optimizer": optimizer.state_dict(), "scheduler": scheduler.state_dict(),
                        "epoch": epoch}, os.path.join(checkpoint_path, f"train_state_epoch_{epoch}.pt"))
            with open(checkpoint_logs, "a") as f:
                f.writelines(f"{line}\n" for line in pending)
            pending.clear()

        if is_main and epoch % opt.sample_interval == 0:
            model.eval()
            for sample in dl:
                model.draw_reconstruction(to_device(sample, device),
                                          os.path.join(checkpoint_imgs, f"sample_img_{epoch}.png"))
                break
            model.train()
        if world > 1:
            torch.distributed.barrier()
    fdist.shutdown()


if __name__ == "__main__":
    main(cli.few_shot_parser().parse_args())
