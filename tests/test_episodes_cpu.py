"""Episode samplers reproduce the reference's random streams (goldens from
src/datasets/utils.py) and the sample-dict contract."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from fpsg_amd.episodes import (EpisodicBatchSampler, SequentialBatchSampler, SyntheticFewShot,
                               collate_episode, extract_episode, shard_episodes, synthetic_episode)


def test_sampler_streams_match_reference():
    g = np.load(os.path.join(GOLDEN, "episode_streams.npz"))
    torch.manual_seed(0)
    assert np.array_equal(torch.stack(list(EpisodicBatchSampler(10, 1, 6))).numpy(), g["episodic_10_1_6"])
    assert np.array_equal(torch.stack(list(EpisodicBatchSampler(40, 3, 4))).numpy(), g["episodic_40_3_4"])
    assert np.array_equal(torch.stack(list(SequentialBatchSampler(5))).numpy(), g["sequential_5"])


def test_extract_episode_matches_reference():
    g = np.load(os.path.join(GOLDEN, "episode_streams.npz"))
    torch.manual_seed(1)
    d = {"class": "chair", "img_data": torch.arange(12.0).view(12, 1), "pc_data": torch.arange(12.0).view(12, 1) + 100}
    eps = [extract_episode(4, 1, d) for _ in range(5)]
    assert np.array_equal(torch.stack([e["xs"].view(-1) for e in eps]).numpy(), g["extract_xs"])
    assert np.array_equal(torch.stack([e["xq"].view(-1) for e in eps]).numpy(), g["extract_xq"])
    assert np.array_equal(torch.stack([e["pcs"].view(-1) for e in eps]).numpy(), g["extract_pcs"])
    assert np.array_equal(np.array([e["tmp"] for e in eps]), g["extract_tmp"])


def test_multi_query_episode_does_not_raise():
    """The reference raises for n_query != 1 (`query_idx.item()`, datasets/utils.py:27)."""
    d = {"class": "c", "img_data": torch.zeros(9, 1), "pc_data": torch.zeros(9, 1)}
    e = extract_episode(3, 5, d)
    assert e["xq"].shape[0] == 5 and isinstance(e["tmp"], int)
    assert extract_episode(3, -1, d)["xq"].shape[0] == 6


def test_synthetic_dataset_layout():
    ds = SyntheticFewShot(n_classes=2, per_class=6, n_support=3, n_query=2, n_pts=64, img_size=32)
    s = collate_episode(ds[7])
    assert s["xs"].shape == (1, 3, 3, 32, 32) and s["xq"].shape == (1, 2, 3, 32, 32)
    assert s["xad"].shape == (1, 3, 3, 32, 32) and s["pcad"].shape == (1, 3, 64, 3)
    assert s["pcs"].shape == (1, 3, 64, 3) and s["pcq"].shape == (1, 2, 64, 3)
    assert s["class"] == ["class01"]
    pc = s["pcs"][0]
    assert torch.allclose(pc.norm(dim=-1).amax(dim=-1), torch.ones(3), atol=1e-6)  # unit ball
    assert pc.mean(dim=1).abs().max() < 0.2
    assert -1 <= float(s["xs"].min()) and float(s["xs"].max()) <= 1


def test_episode_sharding_is_world_size_independent():
    for world in (1, 2, 4, 8):
        seen = sorted(i for r in range(world) for i in shard_episodes(64, r, world))
        assert seen == list(range(64))
    e = synthetic_episode(2, 1, n_pts=32, img_size=32, seed=5)
    f = synthetic_episode(2, 1, n_pts=32, img_size=32, seed=5)
    assert torch.equal(e["pcs"], f["pcs"]) and torch.equal(e["xq"], f["xq"])


def test_prefetcher_yields_the_same_episodes_in_order():
    from fpsg_amd.episodes import EpisodePrefetcher, synthetic_episode
    eps = [synthetic_episode(2, 1, n_pts=64, img_size=8, seed=s) for s in range(5)]
    got = list(EpisodePrefetcher(iter(eps), "cpu", depth=2))
    assert len(got) == 5
    for a, b in zip(eps, got):
        assert all(torch.equal(a[k], b[k]) for k in ("xs", "xq", "xad", "pcs", "pcq", "pcad"))

    def broken():
        yield eps[0]
        raise RuntimeError("loader failed")
    it = EpisodePrefetcher(broken(), "cpu")
    next(it)
    with pytest.raises(RuntimeError, match="loader failed"):
        next(it)
