"""Host twins of the transforms (wv_*_forward_cpu) and the plugin dispatch around them -- runs without a GPU.

The reference calls its transform inside forked DataLoader workers (flikr_coco.py:59-60 -> custom_transforms.py:145-157);
with an unchanged YAML node and num_workers > 0 the plugin's __call__ lands in the library's host twin.  These tests
pin the twin to the oracle (tolerance as for the device kernel: fp32, fused multiply-adds, the axes' passes in the
kernel's order -- |twin - oracle| <= 4e-6 * 2^level) and to the committed golden vectors, and check that the plugin
never changes device silently.
"""
import numpy as np
import pytest
import torch
from PIL import Image
from torch.utils.data import DataLoader, Dataset

from oracle import swt_np
from wvhash import _lib, synth
from wvhash.transforms import (DWTTransform, RawStackTransform, SWTTransform, build_transform, dwt2d_host,
                               rawstack_host, swt2d_host)

CASES = [("haar", 1, 32, 32), ("haar", 2, 64, 48), ("haar", 3, 224, 224), ("db2", 1, 40, 56), ("db2", 3, 224, 224),
         ("db2", 3, 64, 256), ("db4", 1, 224, 224), ("db4", 2, 96, 96), ("bior4.4", 1, 224, 224),
         ("bior4.4", 2, 64, 32), ("db2", 3, 8, 8), ("haar", 1, 2, 4), ("db4", 3, 32, 32), ("db2", 2, 36, 44)]


def tol(level):
    return 4e-6 * 2 ** level


@pytest.mark.parametrize("wl,lev,H,W", CASES)
@pytest.mark.parametrize("channels_last", [False, True])
def test_swt_twin_matches_oracle(wl, lev, H, W, channels_last):
    img = synth.natural_images(2, H, W, seed=H * 7 + W + lev)
    ref = swt_np.c_transform_batch(img, wl, lev)
    x = torch.from_numpy(img)
    if not channels_last:
        x = x.permute(0, 3, 1, 2).contiguous()
    got = swt2d_host(x, wl, lev, channels_last=channels_last).numpy()
    assert got.shape == (2, 3, 4, H, W) and got.dtype == np.float32
    assert np.abs(got - ref).max() <= tol(lev)


def test_swt_twin_float_input_is_used_as_is():
    img = synth.noise_images(2, 64, 48, seed=3)
    x = torch.from_numpy(img).permute(0, 3, 1, 2).contiguous()
    a = swt2d_host(x, "db2", 2)
    b = swt2d_host(x.float() / 255.0, "db2", 2)
    assert torch.equal(a, b)                                   # u8 is divided by 255 in fp32, exactly like astype/255
    c = swt2d_host(x.float(), "db2", 2)                        # float input is NOT rescaled
    assert np.abs(c.numpy() / 255.0 - a.numpy()).max() < 1e-4


def test_swt_twin_matches_golden(golden_dir):
    g = np.load(f"{golden_dir}/swt_golden.npz")
    names = sorted({k.split("/")[0] for k in g.files if k.endswith("/img")})
    assert names
    for name in names:
        img, out = g[name + "/img"], g[name + "/out"]
        wl, lev = bytes(g[name + "/wavelet"]).decode(), int(g[name + "/meta"][0])
        got = swt2d_host(torch.from_numpy(img), wl, lev, channels_last=True).numpy()
        assert np.abs(got - out).max() <= tol(lev), name
    got = swt2d_host(torch.from_numpy(synth.natural_images(1, 224, 224, seed=1234)), "db2", 3, channels_last=True).numpy()
    assert np.abs(got.reshape(-1)[::9973] - g["db2_l3_224/samples"]).max() <= tol(3)


@pytest.mark.parametrize("wl,lev", [("haar", 1), ("db2", 2), ("bior4.4", 1), ("db4", 2)])
def test_dwt_twin_matches_oracle(wl, lev):
    img = synth.natural_images(2, 50, 38, seed=lev)
    got = dwt2d_host(torch.from_numpy(img), wl, lev, channels_last=True).numpy()
    ref = np.stack([np.stack([swt_np.wavedec2_coarsest(img[b, :, :, c].astype(np.float32) / 255.0, wl, lev)
                              for c in range(3)]) for b in range(2)])
    assert got.shape == ref.shape
    assert np.abs(got - ref).max() <= tol(lev)


def test_rawstack_twin_is_exact():
    img = synth.noise_images(2, 24, 40, seed=9)
    got = rawstack_host(torch.from_numpy(img), 5, channels_last=True)
    want = (torch.from_numpy(img).permute(0, 3, 1, 2).float() / 255.0).unsqueeze(2).expand(-1, -1, 5, -1, -1)
    assert torch.equal(got, want)


def test_twin_argument_errors():
    x = torch.zeros((1, 3, 20, 20), dtype=torch.uint8)
    with pytest.raises(ValueError, match="divisible"):
        swt2d_host(x, "db2", 3)
    with pytest.raises(TypeError):
        swt2d_host(x.double(), "haar", 1)
    assert swt2d_host(x[:0], "haar", 1).shape == (0, 3, 4, 20, 20)


# ---------------------------------------------------------------------------------- plugin dispatch
def _pil(seed=0, h=224, w=224):
    return Image.fromarray(synth.natural_images(1, h, w, seed=seed)[0])


def test_plugin_on_cpu_device_matches_reference_contract():
    tf = SWTTransform(level=3, wavelet="db2", device="cpu")
    img = _pil(4, 230, 250)                                      # not a multiple of 8: fix_size resizes (BICUBIC)
    out = tf(img)
    sized = swt_np.fix_size(img, 3)
    ref = swt_np.transform_image(np.array(sized), "db2", 3)
    assert out.dtype == torch.float32 and tuple(out.shape) == ref.shape == (3, 4, 232, 256)
    assert np.abs(out.numpy() - ref).max() <= tol(3)
    raw = RawStackTransform(copies=4, device="cpu")(_pil(5))
    assert torch.equal(raw[:, 0], raw[:, 3]) and tuple(raw.shape) == (3, 4, 224, 224)
    d = DWTTransform(level=2, wavelet="db2", device="cpu")(_pil(6))
    assert tuple(d.shape) == (3, 4, 58, 58)


@pytest.mark.skipif(torch.cuda.is_available(), reason="the main process owns a GPU here")
def test_plugin_without_gpu_fails_loudly_in_the_main_process():
    with pytest.raises(_lib.WvhashUnavailable, match="device='cpu'"):
        SWTTransform(level=1, wavelet="haar")(_pil())


class _Items(Dataset):
    def __init__(self, n, transform):
        self.imgs = synth.natural_images(n, 256, 256, seed=2)
        self.transform = transform

    def __len__(self):
        return len(self.imgs)

    def __getitem__(self, i):
        return {"image": self.transform(Image.fromarray(self.imgs[i])), "label": torch.zeros(3), "path": str(i)}


@pytest.mark.parametrize("node", [{"Resize": {"size": 256}, "CenterCrop": {"size": 224},
                                   "SWTTransform": {"level": 3, "wavelet": "db2"}},
                                  {"Resize": {"size": 256}, "CenterCrop": {"size": 224},
                                   "SWTTransform": {"level": 1, "wavelet": "bior4.4"}}])
def test_unedited_yaml_node_in_forked_dataloader_workers(node):
    """getter.py:25-35 builds the pipeline from the YAML node; flikr_coco.py:59-60 calls it inside worker processes.
    No defer, no device kwarg: the workers take the host twin by themselves, the batch is the reference's tensor."""
    tf = build_transform(node)
    ds = _Items(6, tf)
    batches = [b["image"] for b in DataLoader(ds, batch_size=3, num_workers=2, shuffle=False)]
    got = torch.cat(batches)
    assert tuple(got.shape) == (6, 3, 4, 224, 224) and got.dtype == torch.float32
    wl, lev = node["SWTTransform"]["wavelet"], node["SWTTransform"]["level"]
    crop = np.stack([np.array(tf.transforms[1](tf.transforms[0](Image.fromarray(im)))) for im in ds.imgs])
    ref = swt_np.c_transform_batch(crop, wl, lev)
    assert np.abs(got.numpy() - ref).max() <= tol(lev)


def test_model_refuses_a_raw_batch_without_a_bound_transform():
    from wvhash.models import SharedDinoHashing
    from wvhash.models.vit import tiny_vit
    fusion = {"type": "cross_attention_advanced", "output_dim": 384, "num_heads": 8, "num_queries": 4}
    net = SharedDinoHashing({"name": "dinov2_vits14"}, fusion, {"nbits": 16}, backbone=tiny_vit()).eval()
    with pytest.raises(RuntimeError, match="never guessed"):
        net(torch.zeros((2, 3, 224, 224), dtype=torch.uint8))
    with pytest.raises(ValueError, match="no SWTTransform"):
        net.bind_transform(object())
    net.bind_transform(build_transform({"Resize": {"size": 256}, "SWTTransform": {"level": 3, "wavelet": "db2"}},
                                       defer=True))
    assert net._bound_transform.wavelet == "db2" and net._bound_transform.level == 3
