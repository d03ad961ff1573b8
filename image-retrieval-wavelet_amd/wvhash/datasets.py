"""Readers of the reference's on-disk retrieval splits (``train.txt`` / ``test.txt`` / ``database.txt``:
one line per image, ``<relative path> <0/1 label bits ...>``) with the item layout the evaluation engine
expects: ``{"image", "label", "path"}``.

Reference: MIRFlickrHashing / COCOHashing, /root/reference/main/datasets/flikr_coco.py:7-63, 65-124.
Evaluation-side only (no multi-crop training augmentation).
"""
import os
from collections import defaultdict

import torch
from PIL import Image
from torch.utils.data import Dataset

_SPLIT_FILES = {"train": "train.txt", "query": "test.txt", "val": "test.txt", "test": "test.txt",
                "database": "database.txt", "gallery": "database.txt"}


def read_split_file(list_path):
    """-> (relative paths, float32 label matrix [n, Lc])."""
    names, rows = [], []
    with open(list_path, "r") as f:
        for line in f:
            parts = line.strip().split()
            if not parts:
                continue
            names.append(parts[0])
            rows.append([float(x) for x in parts[1:]])
    width = {len(r) for r in rows}
    if len(width) > 1:
        raise ValueError(f"{list_path}: label rows of different lengths {sorted(width)}")
    labels = torch.tensor(rows, dtype=torch.float32) if rows else torch.zeros((0, 0))
    return names, labels


class _SplitFileHashing(Dataset):
    image_subdir = ""          # MIRFLICKR keeps its files under images/, COCO's lists carry the sub-folder
    blank_size = (224, 224)    # stand-in for unreadable files (the reference substitutes a black image)

    def __init__(self, data_dir, mode="train", transform=None, **kwargs):
        if mode not in _SPLIT_FILES:
            raise ValueError(f"Mode inconnu: {mode}")
        self.data_dir, self.mode, self.transform = data_dir, mode, transform
        names, self.label_matrix = read_split_file(os.path.join(data_dir, _SPLIT_FILES[mode]))
        root = os.path.join(data_dir, self.image_subdir) if self.image_subdir else data_dir
        self.paths = [os.path.join(root, n) for n in names]
        self.labels = list(self.label_matrix)
        self.instance_dict = defaultdict(list)
        for row, col in (self.label_matrix == 1.0).nonzero().tolist():
            self.instance_dict[col].append(row)

    def __len__(self):
        return len(self.paths)

    def __getitem__(self, idx):
        path = self.paths[idx]
        try:
            img = Image.open(path).convert("RGB")
        except (OSError, ValueError):
            img = Image.new("RGB", self.blank_size, (0, 0, 0))
        if self.transform is not None:
            img = self.transform(img)
        return {"image": img, "label": self.labels[idx].clone().detach().float(), "path": path}


class MIRFlickrHashing(_SplitFileHashing):
    image_subdir = "images"


class COCOHashing(_SplitFileHashing):
    blank_size = (256, 256)
