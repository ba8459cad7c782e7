#!/usr/bin/env python3
"""Evaluation entry point: per-class Chamfer and EMD of a trained model.

Same command line and report lines as reference ``src/evaluate_Network.py`` (``main :65-123``):
loads ``<model_path>/<name>/<eval_model>``, runs ``ImgPCProtoNet._return_reconstruction`` on
every test episode (HIP Chamfer K1 + the HIP Sinkhorn divergence K2b, the form ``emd_wrapper`` calls) and prints
``Class: <c> -- Rec CD: <mean>; Rec EMD: <mean>``.  With ``--npy_folder`` the generated and
ground-truth clouds (+ a side-by-side PNG) of every item are dumped instead, which is the
reference's commented-out "OPTION 2" (``:111``).

    python evaluate_Network.py --synthetic --n_shot 1 --n_query 1 --sequential_eval \
        --model_path /tmp/ckpt --name 0 --eval_model model_epoch_2.pt
"""
from __future__ import annotations

import os

os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")   # before the HIP runtime initialises (fpsg_amd/__init__.py)
import statistics
from collections import defaultdict

import torch

from fpsg_amd import cli
from fpsg_amd.engine import EvalItem, build_model, to_device


def main(opt):
    cli.validate(opt)
    n_query = opt.n_shot if opt.n_query == 0 else opt.n_query
    device = cli.pick_device(opt)
    checkpoint_path = os.path.join(opt.model_path, opt.name)
    os.makedirs(os.path.join(checkpoint_path, "images"), exist_ok=True)

    _, ds_test = cli.build_datasets(opt, n_query, device)
    _, dl_test = cli.build_loaders(opt, ds_test, ds_test)

    model = build_model(opt)
    weights = os.path.join(checkpoint_path, opt.eval_model)
    if opt.eval_model != "NONE" or os.path.exists(weights):
        model.load_state_dict(torch.load(weights, map_location="cpu", weights_only=True))
    else:
        print("WARNING: --eval_model not given, evaluating randomly initialised weights")
    model = model.to(device).eval()

    per_class_cd, per_class_emd = defaultdict(list), defaultdict(list)
    # the weights do not change while evaluating: transformed filters, stacked decoder weights and BatchNorm coefficients
    # are made once, not per item; on a GPU the item in front of the EMD is replayed as a hipGraph (engine.EvalItem)
    with EvalItem(model) as run_item:
        for item, sample in enumerate(dl_test):
            sample = to_device(sample, device)
            if getattr(opt, "npy_folder", ""):
                os.makedirs(opt.npy_folder, exist_ok=True)
                model.draw_reconstruction(sample, [item, opt.npy_folder])
                continue
            out = run_item(sample)
            name = sample["class"][0]
            per_class_cd[name].append(out["cd_loss"].item() / n_query)
            per_class_emd[name].append(out["emd_loss"].item() / n_query)
    for name in sorted(per_class_cd):
        print(f"Class: {name} -- Rec CD: {statistics.mean(per_class_cd[name])}; "
              f"Rec EMD: {statistics.mean(per_class_emd[name])}")
    return per_class_cd, per_class_emd


if __name__ == "__main__":
    main(cli.few_shot_parser(evaluation=True).parse_args())
