"""Episode assembly: the sample-dict contract the hot path consumes.

Mirrors reference ``src/datasets/utils.py`` (``extract_episode :4-28``,
``EpisodicBatchSampler :31-42``, ``SequentialBatchSampler :45-54``) and the episode layout
of ``FewShotModelNet.__getitem__`` (``src/datasets/modelnet.py:110-128``; ShapeNet is the
same, ``shapenet.py:130-145``): support / query split of one class corpus by a random
permutation, plus ``n_support`` random (image, cloud) "ad" pairs from the whole corpus.
The random streams are the reference's (same ``torch.randperm`` calls in the same order),
pinned by ``tests/golden/episode_streams.npz``.

Fix of a reference defect: ``'tmp'`` is the index of the FIRST query item
instead of ``query_idx.item()``, which raises for ``n_query != 1``.

``SyntheticFewShot`` provides corpora of the reference's shapes without any files
(SURVEY.md 8d); with ``device='cuda'`` the corpora live in HBM and an episode is assembled
by on-device indexing -- no per-step host->device copy of the ~42 MB of images the
reference re-uploads every step.
"""
from __future__ import annotations

import math
import threading

import torch
from torch.utils.data import Dataset

_staging = threading.local()      # .ring: the pinned staging ring of the thread that is assembling episodes, if any


class _PinnedRing:
    """Pinned host buffers handed out to ``take_rows`` while ``EpisodePrefetcher``'s worker assembles episodes:
    ``slots`` generations of buffers (one generation = the tensors of one episode), reused round-robin.  A generation
    is taken again only after the upload that read it has finished (its event), so ``slots`` must exceed the number
    of episodes in flight (the prefetcher's queue depth + the one being assembled + the one being consumed)."""

    def __init__(self, slots: int):
        self.slots = slots
        self.gen = 0
        self.bufs = [dict() for _ in range(slots)]        # per generation: (shape, dtype, ordinal) -> pinned tensor
        self.events = [None] * slots
        self.ordinal = 0

    def next_episode(self):
        self.gen = (self.gen + 1) % self.slots
        self.ordinal = 0
        ev = self.events[self.gen]
        if ev is not None:
            ev.synchronize()                                # the upload of the episode that used these buffers

    def take(self, shape, dtype):
        key = (tuple(shape), dtype, self.ordinal)
        self.ordinal += 1
        buf = self.bufs[self.gen].get(key)
        if buf is None:
            buf = torch.empty(shape, dtype=dtype).pin_memory()
            self.bufs[self.gen][key] = buf
        return buf

    def uploaded(self, event):
        self.events[self.gen] = event


def take_rows(corpus: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
    """``corpus[idx]`` (rows of a corpus, as ``modelnet.py:117-120`` indexes them).  On the host ``index_select`` is a
    row-wise memcpy -- the advanced-indexing kernel behind ``corpus[idx]`` moves the 69 images of a 32-shot episode
    (42 MB) several times slower once the intra-op pool matches the cores the process really has
    (``cli.limit_host_threads``), which made episode assembly the bottleneck of host-resident corpora; and when the calling
    thread has a pinned staging ring open (``EpisodePrefetcher``) the rows land in pinned memory at once, ready for an
    asynchronous upload (no second 42 MB ``pin_memory()`` copy)."""
    if corpus.device.type != "cpu":
        return corpus[idx.to(corpus.device)]               # resident corpora: the device's gather kernel
    ring = getattr(_staging, "ring", None)
    if ring is not None:
        out = ring.take((idx.numel(),) + tuple(corpus.shape[1:]), corpus.dtype)
        return torch.index_select(corpus, 0, idx, out=out)
    return torch.index_select(corpus, 0, idx)


def extract_episode(n_support: int, n_query: int, d: dict) -> dict:
    n_examples = d["img_data"].size(0)
    if n_query == -1:
        n_query = n_examples - n_support
    example_idx = torch.randperm(n_examples)[:(n_support + n_query)]
    support_idx = example_idx[:n_support]
    query_idx = example_idx[n_support:]
    return {
        "class": d["class"],
        "xs": take_rows(d["img_data"], support_idx),
        "xq": take_rows(d["img_data"], query_idx),
        "pcs": take_rows(d["pc_data"], support_idx),
        "pcq": take_rows(d["pc_data"], query_idx),
        "tmp": int(query_idx[0]) if query_idx.numel() else -1,
    }


class EpisodicBatchSampler:
    """``n_episodes`` batches of ``n_way`` random class indices."""

    def __init__(self, n_classes: int, n_way: int, n_episodes: int):
        self.n_classes, self.n_way, self.n_episodes = n_classes, n_way, n_episodes

    def __len__(self):
        return self.n_episodes

    def __iter__(self):
        for _ in range(self.n_episodes):
            yield torch.randperm(self.n_classes)[:self.n_way]


class SequentialBatchSampler:
    """Every item once, in order, one per batch."""

    def __init__(self, n_classes: int):
        self.n_classes = n_classes

    def __len__(self):
        return self.n_classes

    def __iter__(self):
        for i in range(self.n_classes):
            yield torch.LongTensor([i])


SequentialBatchSamplerV2 = SequentialBatchSampler  # identical in the reference (:57-65)


def normalize_unit_ball(points: torch.Tensor) -> torch.Tensor:
    """Centre on the mean and divide by the largest norm
    (``src/datasets/modelnet.py:66-69``); ``points [..., N, 3]``."""
    points = points - points.mean(dim=-2, keepdim=True)
    return points / points.norm(dim=-1).amax(dim=-1)[..., None, None]


def synthetic_clouds(n: int, n_pts: int, generator: torch.Generator) -> torch.Tensor:
    """Uniform-in-ball clouds, normalised like the reference loader."""
    direction = torch.randn(n, n_pts, 3, generator=generator)
    direction = direction / direction.norm(dim=-1, keepdim=True)
    radius = torch.rand(n, n_pts, 1, generator=generator) ** (1.0 / 3.0)
    return normalize_unit_ball(direction * radius).contiguous()


def synthetic_images(n: int, size: int, generator: torch.Generator) -> torch.Tensor:
    """U(-1,1) images: the range of ``Normalize(.5,.5)`` output (trainNetwork.py:22-27)."""
    return torch.rand(n, 3, size, size, generator=generator) * 2 - 1


class SyntheticFewShot(Dataset):
    """File-free stand-in for ``FewShotModelNet`` / ``FewShotShapeNet`` with the same item
    layout.  ``len()`` = number of corpus items; item ``i`` yields an episode of item ``i``'s
    class, as the reference datasets do."""

    def __init__(self, n_classes: int = 4, per_class: int = 40, n_support: int = 1,
                 n_query: int = 1, n_pts: int = 2048, img_size: int = 224, seed: int = 1234,
                 device: str | torch.device = "cpu"):
        g = torch.Generator().manual_seed(seed)
        self.n_support, self.n_query, self.n_way = n_support, n_query, 1
        if per_class < n_support + max(n_query, 0):
            raise ValueError("per_class must cover n_support + n_query")
        self.reference = {}
        imgs, pcs = [], []
        for c in range(n_classes):
            name = f"class{c:02d}"
            im = synthetic_images(per_class, img_size, g).to(device)
            pc = synthetic_clouds(per_class, n_pts, g).to(device)
            self.reference[name] = {"imgs": im, "pcs": pc}
            imgs.append(im)
            pcs.append(pc)
        self.classes = list(self.reference)
        self.img_corpus = torch.cat(imgs, dim=0)
        self.pc_corpus = torch.cat(pcs, dim=0)
        self.per_class = per_class
        self.item_len = self.img_corpus.size(0)

    def __len__(self):
        return self.item_len

    def __getitem__(self, index):
        name = self.classes[int(index) // self.per_class]
        ans = extract_episode(self.n_support, self.n_query, {
            "class": name,
            "img_data": self.reference[name]["imgs"],
            "pc_data": self.reference[name]["pcs"],
        })
        ad_idx = torch.randperm(self.item_len)[:self.n_support]
        ans["xad"] = take_rows(self.img_corpus, ad_idx)
        ans["pcad"] = take_rows(self.pc_corpus, ad_idx)
        return ans


def collate_episode(sample: dict) -> dict:
    """What ``DataLoader(batch_sampler=...)`` does to one episode: a leading batch axis of 1
    on every tensor, ``class`` wrapped in a list."""
    out = {}
    for k, v in sample.items():
        if torch.is_tensor(v):
            out[k] = v.unsqueeze(0)
        elif k == "class":
            out[k] = [v]
        else:
            out[k] = torch.tensor([v])
    return out


def synthetic_episode(n_support: int, n_query: int, n_pts: int = 2048, img_size: int = 224,
                      seed: int = 1234, device="cpu") -> dict:
    """One collated episode of fresh synthetic data (benchmarks, smoke tests)."""
    g = torch.Generator().manual_seed(seed)
    s = {
        "class": "synthetic",
        "xs": synthetic_images(n_support, img_size, g),
        "xq": synthetic_images(n_query, img_size, g),
        "xad": synthetic_images(n_support, img_size, g),
        "pcs": synthetic_clouds(n_support, n_pts, g),
        "pcq": synthetic_clouds(n_query, n_pts, g),
        "pcad": synthetic_clouds(n_support, n_pts, g),
        "tmp": 0,
    }
    s = collate_episode(s)
    return {k: (v.to(device) if torch.is_tensor(v) else v) for k, v in s.items()}


def shard_episodes(n_episodes: int, rank: int, world: int) -> range:
    """Episode ids of a global step handled by ``rank``: ``rank, rank+world, ...`` so the
    episode SET of a step does not depend on the world size."""
    return range(rank, n_episodes, world)


def episodes_per_rank(n_episodes: int, world: int) -> int:
    return math.ceil(n_episodes / world)


class EpisodePrefetcher:
    """Iterator over ``it`` whose episodes arrive on ``device`` ahead of their use.

    The reference assembles an episode on the host and uploads it inside the step
    (``trainNetwork.py:37-43,141``: ``to_cuda``) -- with 32-shot episodes that is 69 images of CPU
    indexing plus 42 MB over PCIe per episode, serial with the GPU work.  Here a background thread
    draws the next episodes (same order, same RNG stream: it is the only consumer of the loader while
    it is open -- pass it exactly the episodes that will be used and ``close()`` it before anything
    else touches the global RNG),
    stages them in pinned memory and uploads them on a side stream; ``next()`` makes the compute
    stream wait on the upload's event, not the host.  On a CPU device it is a plain pass-through."""

    _KEYS = ("xs", "xq", "xad", "pcs", "pcq", "pcad")

    def __init__(self, it, device, depth: int = 2):
        import queue
        import threading
        self._it, self._device = it, torch.device(device)
        self._cuda = self._device.type == "cuda"
        if self._cuda and self._device.index is None:
            self._device = torch.device("cuda", torch.cuda.current_device())
        self._q = queue.Queue(maxsize=max(1, depth))
        self._stop = False
        self._stream = torch.cuda.Stream(device=self._device) if self._cuda else None
        self._thread = threading.Thread(target=self._work, daemon=True)
        self._thread.start()

    def _work(self):
        try:
            ring = None
            if self._cuda:
                torch.cuda.set_device(self._device)
                # rows of host corpora are gathered straight into pinned buffers (take_rows): one 42 MB copy per
                # 32-shot episode instead of three (advanced indexing, pin_memory(), upload staging)
                ring = _staging.ring = _PinnedRing(self._q.maxsize + 3)
            it = iter(self._it)
            while not self._stop:
                if ring is not None:
                    ring.next_episode()
                try:
                    sample = next(it)
                except StopIteration:
                    break
                event = None
                if self._cuda:
                    with torch.cuda.stream(self._stream):
                        for k in self._KEYS:
                            t = sample.get(k)
                            if torch.is_tensor(t) and t.device != self._device:
                                src = t if t.is_pinned() else t.pin_memory()
                                sample[k] = src.to(self._device, non_blocking=True)
                        event = torch.cuda.Event()
                        event.record(self._stream)
                    ring.uploaded(event)
                self._q.put((sample, event))
            _staging.ring = None
            self._q.put(None)
        except BaseException as exc:          # surfaces in the consumer
            self._q.put(exc)

    def __iter__(self):
        return self

    def __next__(self):
        item = self._q.get()
        if item is None:
            raise StopIteration
        if isinstance(item, BaseException):
            raise item
        sample, event = item
        if event is not None:
            cur = torch.cuda.current_stream(self._device)
            cur.wait_event(event)
            for k in self._KEYS:
                t = sample.get(k)
                if torch.is_tensor(t) and t.is_cuda:
                    t.record_stream(cur)
        return sample

    def close(self):
        """Stops the background thread and WAITS for it (episodes already staged are dropped): after
        ``close()`` nothing draws from the loader -- or from the global RNG behind it -- any more.  Give
        the prefetcher exactly the episodes it will be asked for (``itertools.islice``), so that the
        worker never draws one the consumer does not use."""
        import queue
        self._stop = True
        while self._thread.is_alive():
            try:                               # a worker blocked in put() needs room to see the flag
                self._q.get(timeout=0.05)
            except queue.Empty:
                pass
        self._thread.join()
        try:
            while True:
                self._q.get_nowait()
        except queue.Empty:
            pass
