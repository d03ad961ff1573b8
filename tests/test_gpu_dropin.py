"""The drop-in boundary the way the reference really runs it:

* an UNEDITED transform YAML node (getter.py:25-35) inside forked DataLoader workers with pinned memory
  (flikr_coco.py:52-63, evaluate.py:79-91) feeding a model built from cfg.model kwargs -> the library's host twin in
  the workers, codes identical to the model fed with the kernel-made 5-D tensor;
* the deferred (batched) form of the same node: raw uint8 batches, the transform object bound to the model by the
  engine, band-major kernel output -- no wavelet to guess, no second copy of the sub-band tensor;
* host twin == device kernel bit for bit on the sliding kernel's shapes; band-major == permuted reference layout.
"""
import numpy as np
import pytest
import torch
from PIL import Image
from torch.utils.data import Dataset

from oracle import ranking, swt_np
from wvhash import synth
from wvhash.engine import evaluate
from wvhash.engine.evaluate import get_tester
from wvhash.models import MultiDinoHashing, SharedDinoHashing
from wvhash.models.vit import tiny_vit
from wvhash.transforms import RawStackTransform, SWTTransform, build_transform, rawstack, swt2d, swt2d_host

pytestmark = pytest.mark.gpu

NODES = [{"Resize": {"size": 256}, "CenterCrop": {"size": 224}, "SWTTransform": {"level": 3, "wavelet": "db2"}},
         {"Resize": {"size": 256}, "CenterCrop": {"size": 224}, "SWTTransform": {"level": 1, "wavelet": "bior4.4"}}]
MODEL_KWARGS = {"backbone_config": {"name": "dinov2_vits14", "frozen": True},
                "fusion_config": {"type": "cross_attention_advanced", "output_dim": 384, "num_heads": 8, "dropout": 0.1,
                                  "num_queries": 4, "sub_band_dropout_p": 0.3, "ortho_weight": 0.1},
                "binary_config": {"nbits": 64}, "modelhooks": None, "with_autocast": True}


class SynthHashing(Dataset):
    """{"image","label","path"} items like MIRFlickrHashing.__getitem__ (flikr_coco.py:52-63)."""

    def __init__(self, n, seed, transform, size=(300, 280), classes=38):
        self.imgs = synth.natural_images(n, size[0], size[1], seed=seed)
        self.labels = synth.multi_hot_labels(n, classes, 0.10, seed)
        self.transform = transform

    def __len__(self):
        return len(self.imgs)

    def __getitem__(self, i):
        return {"image": self.transform(Image.fromarray(self.imgs[i])), "label": self.labels[i], "path": str(i)}


def make_net(cls=SharedDinoHashing, nbits=64, num_queries=4):
    torch.manual_seed(0)
    kw = dict(MODEL_KWARGS, binary_config={"nbits": nbits})
    kw["fusion_config"] = dict(kw["fusion_config"], num_queries=num_queries)
    if cls is MultiDinoHashing:
        kw["backbones_config"] = [kw.pop("backbone_config")] * 4
        net = cls(backbones=[tiny_vit() for _ in range(4)], **kw)
    else:
        net = cls(backbone=tiny_vit(), **kw)
    net.hash_fc.weight.data.mul_(50)                      # spread the logits away from 0
    for name, prm in net.named_parameters():              # DINOv2 initialises LayerScale at 1e-5: with random weights the
        if name.endswith(".gamma"):                       # CLS token would barely depend on the image; make it depend
            prm.data.fill_(1.0)
    return net.cuda().eval()


def sized_images(ds, tf):
    """The uint8 images after the host-side sizing steps of the pipeline (everything before the wavelet plugin)."""
    out = []
    for im in ds.imgs:
        pil = Image.fromarray(im)
        for step in tf.transforms[:-1]:
            pil = step(pil)
        out.append(np.array(pil))
    return np.stack(out)


@pytest.mark.parametrize("wl,lev", [("db2", 3), ("haar", 1), ("bior4.4", 1), ("db4", 1), ("haar", 2), ("db2", 1)])
@pytest.mark.parametrize("channels_last", [False, True])
def test_host_twin_equals_kernel_bit_for_bit(wl, lev, channels_last):
    img = synth.natural_images(3, 224, 224, seed=lev + len(wl))
    x = torch.from_numpy(img)
    if not channels_last:
        x = x.permute(0, 3, 1, 2).contiguous()
    host = swt2d_host(x, wl, lev, channels_last=channels_last)
    dev = swt2d(x.cuda(), wl, lev, channels_last=channels_last).cpu()
    assert torch.equal(host, dev), float((host - dev).abs().max())


@pytest.mark.parametrize("wl,lev,H,W", [("db2", 3, 224, 224), ("haar", 1, 224, 224), ("bior4.4", 1, 56, 256),
                                        ("db2", 3, 512, 384), ("db4", 2, 96, 96)])   # the last two: not the sliding kernel
@pytest.mark.parametrize("out_dtype", [torch.float32, torch.bfloat16])
def test_band_major_output_is_the_permuted_reference_layout(wl, lev, H, W, out_dtype):
    x = torch.from_numpy(synth.natural_images(5, H, W, seed=W)).permute(0, 3, 1, 2).contiguous().cuda()
    inner = swt2d(x, wl, lev, out_dtype=out_dtype)
    outer = swt2d(x, wl, lev, out_dtype=out_dtype, band_major=True)
    assert tuple(outer.shape) == (4, 5, 3, H, W) and outer.is_contiguous()
    assert torch.equal(outer, inner.permute(2, 0, 1, 3, 4))
    # a preallocated band-major buffer larger than one call's batch: chunks written in place
    buf = torch.zeros((4, 5, 3, H, W), dtype=out_dtype, device="cuda")
    swt2d(x, wl, lev, out_dtype=out_dtype, band_major=True, out=buf)
    assert torch.equal(buf, outer)


@pytest.mark.parametrize("node", NODES)
def test_unedited_node_with_dataloader_workers_gives_the_5d_path_codes(node):
    tf = build_transform(node)                                  # exactly Getter.get_transform's result: no defer
    dts = {"test": SynthHashing(10, 1, tf), "gallery": SynthHashing(40, 2, tf)}
    net = make_net()
    tester = get_tester(batch_size=8, num_workers=2, k=20, distance_metric="hamming")
    wl, lev = node["SWTTransform"]["wavelet"], node["SWTTransform"]["level"]
    codes = {}
    for name, ds in dts.items():
        got, _ = tester.get_all_embeddings(ds, net)             # DataLoader(num_workers=2, pin_memory=True)
        with torch.no_grad():
            x = torch.from_numpy(sized_images(ds, tf)).cuda()
            want = net(swt2d(x, wl, lev, channels_last=True))   # the reference's 5-D input, kernel-made
        assert set(got.unique().tolist()) <= {-1.0, 1.0}
        assert torch.equal(got, want), f"{name}: {(got != want).sum().item()} code bits differ"
        codes[name] = got.cpu()
    m = evaluate(net, test_dataset=dts, epoch=1, batch_size=8, num_workers=2, k=20, distance_metric="hamming",
                 exclude=["mean_reciprocal_rank", "precision_at_1", "r_precision", "rpr", "pr", "pr_rc"])
    ref = ranking.calculate_maphashing(codes["test"], dts["test"].labels, codes["gallery"], dts["gallery"].labels, 20,
                                       stable=True)
    assert abs(m["test"]["maphashing_level0"] - ref) < 1e-6


@pytest.mark.parametrize("cls", [SharedDinoHashing, MultiDinoHashing])
@pytest.mark.parametrize("node", NODES)
def test_deferred_node_is_self_describing(node, cls):
    """defer=True: workers return raw uint8; the engine binds dataset.transform to the model -- db2/L3 and bior4.4/L1
    must come out as themselves (round 1 silently transformed them as haar/L1)."""
    tf_d = build_transform(node, defer=True)
    ds = SynthHashing(12, 3, tf_d)
    net = make_net(cls)
    tester = get_tester(batch_size=6, num_workers=2, k=5, distance_metric="hamming")
    got, _ = tester.get_all_embeddings(ds, net)
    wl, lev = node["SWTTransform"]["wavelet"], node["SWTTransform"]["level"]
    with torch.no_grad():
        x = torch.from_numpy(sized_images(ds, tf_d)).cuda()
        want = net(swt2d(x, wl, lev, channels_last=True))
        wrong = net(swt2d(x, "haar", 1, channels_last=True))
    assert torch.equal(got, want)
    assert not torch.equal(got, wrong)                           # the test can tell the wavelets apart
    fresh = make_net(cls)
    with pytest.raises(RuntimeError, match="never guessed"):
        fresh(x.permute(0, 3, 1, 2).contiguous())


def test_deferred_rawstack_and_swt_equal_their_per_image_forms():
    imgs = synth.natural_images(4, 224, 224, seed=8)
    for eager, deferred in ((SWTTransform(level=3, wavelet="db2"), SWTTransform(level=3, wavelet="db2", defer=True)),
                            (RawStackTransform(copies=4), RawStackTransform(copies=4, defer=True))):
        per_image = torch.stack([eager(Image.fromarray(im)) for im in imgs])               # GPU, one image per call
        raw = torch.stack([deferred(Image.fromarray(im)) for im in imgs])
        assert raw.dtype == torch.uint8 and tuple(raw.shape) == (4, 3, 224, 224)
        batched = deferred.apply_batch(raw.cuda()).cpu()
        assert torch.equal(per_image, batched)
        host = torch.stack([type(eager)(**{k: v for k, v in vars(eager).items() if k not in ("defer", "device")},
                                        device="cpu")(Image.fromarray(im)) for im in imgs])
        assert torch.equal(host, batched)


def test_model_path_allocates_the_sub_bands_once():
    """SharedDinoHashing on a raw batch: the kernel writes band-major, the band split is a view.  The reference (and the
    5-D input path) copies the 2.4 MB/image tensor once more (multi_dino_attention.py:818)."""
    net = make_net().bind_transform(SWTTransform(level=3, wavelet="db2", defer=True))
    B = 32
    x = torch.from_numpy(synth.noise_images(B, 224, 224, seed=1)).permute(0, 3, 1, 2).contiguous().cuda()
    band_bytes = B * 3 * 4 * 224 * 224 * 4
    seen = {}
    hook = net.shared_backbone.register_forward_pre_hook(
        lambda mod, args: seen.update(ptr=args[0].data_ptr(), contiguous=args[0].is_contiguous(), shape=tuple(args[0].shape),
                                      alloc=torch.cuda.memory_allocated()))
    with torch.no_grad():
        torch.cuda.synchronize()
        base = torch.cuda.memory_allocated()
        codes_raw = net(x)
        delta_raw = seen["alloc"] - base
        assert seen["shape"] == (4 * B, 3, 224, 224) and seen["contiguous"]
        bands = swt2d(x, "db2", 3)
        base = torch.cuda.memory_allocated()
        codes_5d = net(bands)
        delta_5d = seen["alloc"] - base
    hook.remove()
    assert torch.equal(codes_raw, codes_5d)
    assert delta_raw < 1.05 * band_bytes                          # one sub-band tensor live when the backbone starts
    assert delta_5d >= 0.95 * band_bytes                          # the 5-D path adds its permuted copy on top of `bands`


def test_bf16_bands_under_autocast():
    """c4 shape (multi-DINO per sub-band, num_queries = 8, 128-bit codes, bf16 autocast): the kernel emits bf16 sub-bands
    directly (what autocast's first cast would make), the Nq = 8 head takes the backbones' bf16 features."""
    net = make_net(MultiDinoHashing, nbits=128, num_queries=8).bind_transform(SWTTransform(level=1, wavelet="haar", defer=True))
    x = torch.from_numpy(synth.natural_images(6, 224, 224, seed=4)).permute(0, 3, 1, 2).contiguous().cuda()
    seen = []
    hooks = [b.register_forward_pre_hook(lambda mod, args: seen.append(args[0].dtype)) for b in net.backbones]
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        codes = net(x)
        ref = net(swt2d(x, "haar", 1))                           # fp32 bands, cast by autocast inside the backbone
    for h in hooks:
        h.remove()
    assert seen[:4] == [torch.bfloat16] * 4 and seen[4:] == [torch.float32] * 4
    assert tuple(codes.shape) == (6, 128) and set(codes.unique().tolist()) <= {-1.0, 1.0}
    assert (codes != ref).float().mean().item() < 0.02           # same bf16 inputs to the conv up to RN ties
    bands = swt2d(x, "haar", 1, out_dtype=torch.bfloat16)
    assert torch.equal(bands, swt2d(x, "haar", 1).to(torch.bfloat16))
    ref_np = swt_np.c_transform_batch(x.permute(0, 2, 3, 1).cpu().numpy(), "haar", 1)
    assert np.abs(bands.float().cpu().numpy() - ref_np).max() <= 2.0 * 2 ** -8
