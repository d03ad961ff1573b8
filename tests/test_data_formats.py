"""CPU: the data formats either side of the hot path -- split files (train/test/database.txt), the YAML transform
pipeline in front of the SWT (deferred: workers only size the image), and checkpoint dicts.
Reference: main/datasets/flikr_coco.py:7-124, main/getter.py:25-35, config/transform/basic_swt.yaml,
main/engine/chepoint.py:22-45."""
import numpy as np
import pytest
import torch
from PIL import Image

from wvhash.checkpoint import load_net_state, net_state_of, read_checkpoint
from wvhash.datasets import COCOHashing, MIRFlickrHashing, read_split_file
from wvhash.transforms import CenterCrop, Resize, build_transform


def _write_split(tmp, sub, names, labels, fname):
    (tmp / sub).mkdir(parents=True, exist_ok=True)
    rng = np.random.default_rng(0)
    with open(tmp / fname, "w") as f:
        for n, l in zip(names, labels):
            if not n.startswith("missing"):
                Image.fromarray(rng.integers(0, 256, (300, 400, 3), dtype=np.uint8)).save(tmp / sub / n)
            f.write(n + " " + " ".join(str(int(v)) for v in l) + "\n")
        f.write("\n")   # blank lines are skipped


def test_split_files_and_item_layout(tmp_path):
    labels = np.array([[1, 0, 1, 0], [0, 0, 0, 1], [1, 1, 0, 0]])
    _write_split(tmp_path, "images", ["im1.jpg", "im2.jpg", "missing.jpg"], labels, "database.txt")
    _write_split(tmp_path, "images", ["q1.jpg"], labels[:1], "test.txt")
    tf = build_transform({"Resize": {"size": 256}, "CenterCrop": {"size": 224},
                          "SWTTransform": {"level": 1, "wavelet": "haar"}}, defer=True)
    db = MIRFlickrHashing(str(tmp_path), mode="gallery", transform=tf)
    assert len(db) == 3 and db.paths[0].endswith("images/im1.jpg")
    assert torch.equal(db.label_matrix, torch.tensor(labels, dtype=torch.float32))
    assert dict(db.instance_dict) == {0: [0, 2], 2: [0], 3: [1], 1: [2]}
    item = db[0]
    assert set(item) == {"image", "label", "path"}
    assert item["image"].dtype == torch.uint8 and tuple(item["image"].shape) == (3, 224, 224)   # deferred SWT
    assert item["label"].dtype == torch.float32 and item["label"].tolist() == [1, 0, 1, 0]
    assert int(db[2]["image"].max()) == 0                                   # unreadable file -> black image
    q = MIRFlickrHashing(str(tmp_path), mode="query", transform=tf)
    assert len(q) == 1
    with pytest.raises(ValueError):
        MIRFlickrHashing(str(tmp_path), mode="nope")
    names, lab = read_split_file(str(tmp_path / "database.txt"))
    assert names == ["im1.jpg", "im2.jpg", "missing.jpg"] and lab.shape == (3, 4)


def test_coco_lists_carry_their_subfolder(tmp_path):
    _write_split(tmp_path, "val2014", ["a.jpg"], np.array([[0, 1]]), "tmp.txt")
    with open(tmp_path / "train.txt", "w") as f:
        f.write("val2014/a.jpg 0 1\n")
    ds = COCOHashing(str(tmp_path), mode="train")
    assert ds.paths == [str(tmp_path / "val2014" / "a.jpg")] and ds[0]["image"].size == (400, 300)


def test_resize_and_center_crop_follow_torchvision_sizes():
    img = Image.new("RGB", (400, 300))
    assert Resize(256)(img).size == (341, 256)            # shorter side -> 256, long side int(256 * 400 / 300)
    assert Resize(256)(Image.new("RGB", (300, 400))).size == (256, 341)
    assert Resize((100, 50))(img).size == (50, 100)
    assert CenterCrop(224)(Resize(256)(img)).size == (224, 224)
    small = CenterCrop(224)(Image.new("RGB", (100, 80), (255, 255, 255)))
    assert small.size == (224, 224) and small.getpixel((0, 0)) == (0, 0, 0) and small.getpixel((112, 112)) == (255, 255, 255)
    arr = np.arange(10 * 8 * 3, dtype=np.uint8).reshape(10, 8, 3)
    crop = np.array(CenterCrop((4, 6))(Image.fromarray(arr)))
    assert np.array_equal(crop, arr[3:7, 1:7])
    with pytest.raises(AttributeError):
        build_transform({"NoSuchOp": {}})


def test_checkpoint_dict_roundtrip(tmp_path):
    net = torch.nn.Sequential(torch.nn.Linear(4, 3), torch.nn.BatchNorm1d(3))
    state = {"net_state": {"module." + k: v for k, v in net.state_dict().items()},   # saved from DataParallel
             "epoch": 12, "score": 0.81, "best_model": "epoch_12.ckpt", "config": {"model": {"name": "x"}}}
    path = tmp_path / "rolling.ckpt"
    torch.save(state, path)
    other = torch.nn.Sequential(torch.nn.Linear(4, 3), torch.nn.BatchNorm1d(3))
    rest = load_net_state(other, str(path))
    assert rest["epoch"] == 12 and "net_state" not in rest
    for a, b in zip(net.state_dict().values(), other.state_dict().values()):
        assert torch.equal(a, b)
    assert set(net_state_of(read_checkpoint(str(path)))) == set(net.state_dict())
    with pytest.raises(RuntimeError):                     # strict, like evaluate.py:69
        load_net_state(torch.nn.Linear(4, 3), str(path))


class Thing:                # stands for an OmegaConf node: the weights-only loader must not unpickle it
    pass


def test_checkpoint_with_arbitrary_objects_is_refused(tmp_path):
    path = tmp_path / "bad.ckpt"
    torch.save({"net_state": {}, "config": Thing()}, path)
    with pytest.raises(RuntimeError, match="weights-only"):
        read_checkpoint(str(path))
