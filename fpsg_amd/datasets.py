"""File-backed episode datasets with the reference's list-file and asset formats
(SURVEY.md 8f-N2): host-side I/O only, no kernels.

Follows reference ``src/datasets/modelnet.py`` (``ply_reader :15-29``,
``FewShotSubModelNet :31-82``, ``FewShotModelNet :85-153``), ``src/datasets/shapenet.py``
(``FewShotSubShapeNet :31-101``, ``FewShotShapeNet :103-171``) and, for the pre-training
script, ``src/datasets/mv_dataset.py`` (``MultiViewDataSet :55-142``):

  * ModelNet list line: ``<image path>\\t<ASCII .ply path>``; the class is the 4th path
    component from the end of the image path; at most 2048 vertices are read, short clouds
    are padded with random repeats;
  * ShapeNet list line: ``<item dir>`` holding ``npy_file.npy`` (15000 points, subsampled to
    2048 once at load) and ``images/`` (first view used); the class id is a path component
    (the reference hard-codes component 5; here: the component that is a known synset id);
  * per-class auxiliary lists ``<dataset>+<class>.txt`` in ``auxiliary_dir``;
  * clouds are centred and scaled into the unit ball (``modelnet.py:66-69``);
  * images: centre crop (550 ModelNet / 256 ShapeNet) -> resize 224 -> [0,1] -> (x-.5)/.5
    (``trainNetwork.py:22-34``), implemented with PIL (torchvision is not required).

Whole class corpora are kept as tensors, as in the reference; ``to(device)`` moves them to
HBM so that episodes are assembled by on-device indexing.
"""
from __future__ import annotations

import collections
import os

import numpy as np
import torch
from PIL import Image
from torch.utils.data import Dataset

from .episodes import extract_episode, take_rows

SHAPENET_ID2NAME = {
    "02691156": "airplane", "02773838": "bag", "02801938": "basket", "02808440": "bathtub",
    "02818832": "bed", "02828884": "bench", "02876657": "bottle", "02880940": "bowl",
    "02924116": "bus", "02933112": "cabinet", "02747177": "can", "02942699": "camera",
    "02954340": "cap", "02958343": "car", "03001627": "chair", "03046257": "clock",
    "03207941": "dishwasher", "03211117": "monitor", "04379243": "table", "04401088": "telephone",
    "02946921": "tin_can", "04460130": "tower", "04468005": "train", "03085013": "keyboard",
    "03261776": "earphone", "03325088": "faucet", "03337140": "file", "03467517": "guitar",
    "03513137": "helmet", "03593526": "jar", "03624134": "knife", "03636649": "lamp",
    "03642806": "laptop", "03691459": "speaker", "03710193": "mailbox", "03759954": "microphone",
    "03761084": "microwave", "03790512": "motorcycle", "03797390": "mug", "03928116": "piano",
    "03938244": "pillow", "03948459": "pistol", "03991062": "pot", "04004475": "printer",
    "04074963": "remote_control", "04090263": "rifle", "04099429": "rocket", "04225987": "skateboard",
    "04256520": "sofa", "04330267": "stove", "04530566": "vessel", "04554684": "washer",
    "02992529": "cellphone", "02843684": "birdhouse", "02871439": "bookshelf",
}


# ----------------------------------------------------------------------------- transforms
class ImageTransform:
    """CenterCrop(crop) -> Resize(size) -> ToTensor -> Normalize(.5, .5)."""

    def __init__(self, crop: int, size: int = 224):
        self.crop, self.size = crop, size

    def __call__(self, img: Image.Image) -> torch.Tensor:
        w, h = img.size
        left, top = int(round((w - self.crop) / 2.0)), int(round((h - self.crop) / 2.0))
        img = img.crop((left, top, left + self.crop, top + self.crop))
        if w < h:   # torchvision Resize(int): the smaller edge becomes `size`
            new = (self.size, max(1, int(self.size * img.size[1] / img.size[0])))
        else:
            new = (max(1, int(self.size * img.size[0] / img.size[1])), self.size)
        img = img.resize(new, Image.BILINEAR)
        arr = np.asarray(img, dtype=np.float32) / 255.0
        return torch.from_numpy((arr - 0.5) / 0.5).permute(2, 0, 1).contiguous()


def modelnet_transform() -> ImageTransform:
    return ImageTransform(550)


def shapenet_transform() -> ImageTransform:
    return ImageTransform(256)


# -------------------------------------------------------------------------------- readers
def ply_reader(file_path: str, max_verts: int = 2048):
    """First ``min(n_vertex, max_verts)`` vertex lines of an ASCII PLY as float lists."""
    n_verts = max_verts
    with open(file_path, "r") as f:
        for line in f:
            cur = line.strip()
            if cur == "end_header":
                break
            parts = cur.split(" ")
            if len(parts) > 2 and parts[1] == "vertex":
                n_verts = min(int(parts[2]), n_verts)
        return [[float(s) for s in f.readline().strip().split(" ")] for _ in range(n_verts)]


def to_unit_ball(point_set: np.ndarray, n_pts: int, rng=np.random) -> np.ndarray:
    point_set = np.asarray(point_set, dtype=np.float32)[:, :3]
    if point_set.shape[0] < n_pts:
        extra = rng.choice(len(point_set), n_pts - point_set.shape[0], replace=True)
        point_set = np.concatenate((point_set, point_set[extra, :]))
    point_set = point_set - point_set.mean(axis=0, keepdims=True)
    return point_set / np.sqrt((point_set ** 2).sum(axis=1)).max()


def augment(point_set: np.ndarray, rng=np.random) -> np.ndarray:
    """Random rotation about y + N(0, 0.02) jitter (``modelnet.py:71-75``)."""
    theta = rng.uniform(0, np.pi * 2)
    rot = np.array([[np.cos(theta), -np.sin(theta)], [np.sin(theta), np.cos(theta)]])
    point_set[:, [0, 2]] = point_set[:, [0, 2]].dot(rot)
    return point_set + rng.normal(0, 0.02, size=point_set.shape).astype(np.float32)


# ------------------------------------------------------------------- single-class corpora
class FewShotSubModelNet(Dataset):
    def __init__(self, config_path, loader=ply_reader, transform=None, tgt_transform=None,
                 data_argument=False, n_pts=2048):
        self.imgs, self.pcs = [], []
        with open(config_path, "r") as f:
            for line in f.read().splitlines():
                if line:
                    img, pc = line.split("\t")
                    self.imgs.append(img)
                    self.pcs.append(pc)
        self.loader, self.tfs, self.tgt_tfs = loader, transform, tgt_transform
        self.data_argument, self.n_pts = data_argument, n_pts

    def __len__(self):
        return len(self.imgs)

    def __getitem__(self, index):
        img = Image.open(self.imgs[index]).convert("RGB")
        if self.tfs is not None:
            img = self.tfs(img)
        pts = to_unit_ball(np.asarray(self.loader(self.pcs[index]), dtype=np.float32), self.n_pts)
        if self.data_argument:
            pts = augment(pts)
        return img, torch.from_numpy(np.ascontiguousarray(pts, dtype=np.float32))


class FewShotSubShapeNet(Dataset):
    def __init__(self, config_path, transform=None, tgt_transform=None, data_argument=False, n_pts=2048):
        self.imgs, self.pc_data = [], []
        with open(config_path, "r") as f:
            for item_path in f.read().splitlines():
                npy_file = os.path.join(item_path, "npy_file.npy")
                view_root = os.path.join(item_path, "images")
                if not item_path or not os.path.exists(npy_file):
                    continue
                views = [os.path.join(view_root, v) for v in sorted(os.listdir(view_root))]
                pc = np.load(npy_file)
                choice = np.random.choice(pc.shape[0], n_pts)   # reference: choice(15000, n_pts)
                self.pc_data.append(pc[choice, :])
                self.imgs.append(views)
        self.tfs, self.tgt_tfs = transform, tgt_transform
        self.data_argument, self.n_pts = data_argument, n_pts

    def __len__(self):
        return len(self.imgs)

    def __getitem__(self, index):
        img = Image.open(self.imgs[index][0]).convert("RGB")
        if self.tfs is not None:
            img = self.tfs(img)
        pts = to_unit_ball(self.pc_data[index], self.n_pts)
        if self.data_argument:
            pts = augment(pts)
        return img, torch.from_numpy(np.ascontiguousarray(pts, dtype=np.float32))


def _stack(ds: Dataset):
    imgs, pcs = zip(*(ds[i] for i in range(len(ds))))
    return torch.stack(imgs), torch.stack(pcs)


# ------------------------------------------------------------------------ episode datasets
class _FewShotFiles(Dataset):
    sub_dataset = None

    def __init__(self, config_path, auxiliary_dir, n_classes, n_support, n_query, transform=None,
                 tgt_transform=None):
        print(f"Reading configuration from {config_path} ...")
        with open(config_path, "r") as f:
            self.data_corpus = [line for line in f.read().splitlines() if line]
        self.item_len = len(self.data_corpus)
        self.tfs, self.tgt_tfs = transform, tgt_transform
        self.reference = collections.defaultdict(dict)
        self.auxiliary_dir = auxiliary_dir
        self.n_way, self.n_support, self.n_query = 1, n_support, n_query
        self._build_reference()

    def _build_reference(self):
        assert self.auxiliary_dir is not None, "Auxiliary folder is not generated yet!!!"
        imgs, pcs = [], []
        for name in sorted(os.listdir(self.auxiliary_dir)):
            if not name.endswith(".txt"):
                continue
            class_name = name.split(".")[0].split("+")[1]
            print(f"Building Reference dataset for {class_name} ...")
            sub = type(self).sub_dataset(os.path.join(self.auxiliary_dir, name), transform=self.tfs,
                                         tgt_transform=self.tgt_tfs)
            im, pc = _stack(sub)
            self.reference[class_name] = {"imgs": im, "pcs": pc}
            imgs.append(im)
            pcs.append(pc)
        self.img_corpus = torch.cat(imgs, dim=0)
        self.pc_corpus = torch.cat(pcs, dim=0)

    def to(self, device):
        """Moves every corpus to ``device`` (episodes are then indexed there)."""
        for entry in self.reference.values():
            entry["imgs"], entry["pcs"] = entry["imgs"].to(device), entry["pcs"].to(device)
        self.img_corpus, self.pc_corpus = self.img_corpus.to(device), self.pc_corpus.to(device)
        return self

    def class_of(self, line: str):
        raise NotImplementedError

    def __len__(self):
        return self.item_len

    def __getitem__(self, index):
        key, shown = self.class_of(self.data_corpus[int(index)])
        ans = extract_episode(self.n_support, self.n_query, {
            "class": shown, "img_data": self.reference[key]["imgs"], "pc_data": self.reference[key]["pcs"]})
        ad = torch.randperm(self.img_corpus.size(0))[:self.n_support]
        ans["xad"] = take_rows(self.img_corpus, ad)
        ans["pcad"] = take_rows(self.pc_corpus, ad)
        return ans


class FewShotModelNet(_FewShotFiles):
    sub_dataset = FewShotSubModelNet

    def class_of(self, line):
        name = line.split("\t")[0].split("/")[-4]
        return name, name


class FewShotShapeNet(_FewShotFiles):
    sub_dataset = FewShotSubShapeNet

    def class_of(self, line):
        for part in line.split("/"):
            if part in SHAPENET_ID2NAME:
                return part, SHAPENET_ID2NAME[part]
        raise KeyError(f"no ShapeNet synset id in path: {line}")


# -------------------------------------------------- labelled clouds for trainPointAE.py
class MultiViewClouds(Dataset):
    """``root/<label>/<train|test>/<item>/<view>.png`` + ``ply_root/<label>/<split>/<item>.ply``
    -> (cloud [N,3], label) pairs (the parts of ``MultiViewDataSet`` the pre-training uses)."""

    def __init__(self, root, ply_root, data_type, sub_cat=None, number_of_points=2048, data_augment=False):
        labels = sorted(sub_cat) if sub_cat else sorted(d for d in os.listdir(root) if os.path.isdir(os.path.join(root, d)))
        self.class_to_idx = {c: i for i, c in enumerate(labels)}
        self.items = []
        for label in labels:
            c_path = os.path.join(root, label, data_type)
            if not os.path.isdir(c_path):
                continue
            for item in sorted(os.listdir(c_path)):
                self.items.append((os.path.join(ply_root, label, data_type, f"{item}.ply"), self.class_to_idx[label]))
        self.n_pts, self.data_augment = number_of_points, data_augment

    def __len__(self):
        return len(self.items)

    def __getitem__(self, index):
        path, label = self.items[index]
        pts = to_unit_ball(np.asarray(ply_reader(path), dtype=np.float32), self.n_pts)
        if self.data_augment:
            pts = augment(pts)
        return torch.from_numpy(np.ascontiguousarray(pts, dtype=np.float32)), label


MODELNET_PRETRAIN_CLASSES = None  # all sub-directories of --root


def multiview_datasets(opt):
    if opt.dataset != "modelnet":
        raise SystemExit("trainPointAE.py: file-backed pre-training is provided for the ModelNet layout; "
                         "use --synthetic otherwise")
    train = MultiViewClouds(opt.root, opt.proot, "train", MODELNET_PRETRAIN_CLASSES, opt.n_pts)
    test = MultiViewClouds(opt.root, opt.proot, "test", MODELNET_PRETRAIN_CLASSES, opt.n_pts)
    return train, test, len(train.class_to_idx)
