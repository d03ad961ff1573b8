"""GPU parity of the HIP SWT path (through the C ABI) against the oracle (oracle/swt_oracle.c).

Tolerance: the kernel accumulates taps in the oracle's order (m = 0..L-1, axis 0 then axis 1) but
with fused multiply-adds, the oracle with separate multiply and add; both are fp32.  Bound used:
|hip - oracle| <= 4e-6 * 2^level absolute (values reach 2^level; measured max is ~1e-6), and the
HIP result must be at least as close to an fp64 evaluation as the fp32 oracle is (x2 slack).
"""
import numpy as np
import pytest
import torch

from oracle import swt_np
from wvhash import synth
from wvhash.transforms import SWTTransform, RawStackTransform, swt2d, rawstack

pytestmark = pytest.mark.gpu


def tol(level):
    return 4e-6 * 2 ** level


def run_hip(img_bhwc, wl, lev, channels_last, as_float=False, out_dtype=torch.float32):
    x = torch.from_numpy(img_bhwc)
    if as_float:
        x = x.float() / 255.0
    if not channels_last:
        x = x.permute(0, 3, 1, 2).contiguous()
    y = swt2d(x.cuda(), wl, lev, channels_last=channels_last, out_dtype=out_dtype)
    torch.cuda.synchronize()
    return y


CASES = [("haar", 1, 32, 32), ("haar", 2, 64, 48), ("haar", 3, 224, 224), ("db2", 1, 40, 56),
         ("db2", 2, 64, 64), ("db2", 3, 224, 224), ("db2", 3, 64, 256), ("db4", 1, 224, 224),
         ("db4", 2, 96, 96), ("bior4.4", 1, 224, 224), ("bior4.4", 2, 64, 32), ("db2", 3, 8, 8),
         ("haar", 1, 2, 4), ("db4", 3, 32, 32), ("db2", 3, 512, 384),
         # every (taps, levels) pair the sliding kernel is instantiated for, at shapes inside and at the edge of its window
         ("haar", 2, 224, 224), ("db2", 1, 224, 224), ("db2", 2, 224, 224), ("db2", 2, 40, 256), ("haar", 3, 48, 40),
         ("db4", 1, 40, 40), ("bior4.4", 1, 56, 256), ("haar", 1, 224, 224)]


@pytest.mark.parametrize("wl,lev,H,W", CASES)
@pytest.mark.parametrize("channels_last", [False, True])
def test_u8_batch_matches_oracle(wl, lev, H, W, channels_last):
    img = synth.natural_images(3, H, W, seed=H * 7 + W + lev)
    ref = swt_np.c_transform_batch(img, wl, lev)
    got = run_hip(img, wl, lev, channels_last).cpu().numpy()
    assert got.shape == (3, 3, 4, H, W) and got.dtype == np.float32
    assert np.abs(got - ref).max() <= tol(lev)


@pytest.mark.parametrize("wl,lev", [("haar", 1), ("db2", 3)])
def test_f32_input_and_fp64_accuracy(wl, lev):
    img = synth.noise_images(2, 224, 224, seed=5)
    ref32 = swt_np.c_transform_batch(img, wl, lev)
    got = run_hip(img, wl, lev, False, as_float=True).cpu().numpy()
    assert np.abs(got - ref32).max() <= tol(lev)
    ref64 = np.stack([swt_np.transform_image(i, wl, lev, dtype=np.float64) for i in img])
    # (float64 of u8/255 differs from float32 of it by <= 6e-8: part of both errors alike)
    assert np.abs(got - ref64).max() <= 2 * np.abs(ref32 - ref64).max() + 1e-7


def test_generic_fallback_path_w_not_multiple_of_4_and_odd_taps():
    img = synth.natural_images(2, 30, 30, seed=11)              # W % 4 != 0 -> generic kernels
    ref = swt_np.c_transform_batch(img, "db2", 1)
    got = run_hip(img, "db2", 1, False).cpu().numpy()
    assert np.abs(got - ref).max() <= tol(1)
    lo = [0.1, 0.2, 0.4, 0.2, 0.1, 0.05]                          # 6 taps: no tiled instantiation
    hi = [-0.05, 0.1, -0.3, 0.3, -0.1, 0.05]
    img = synth.natural_images(1, 32, 32, seed=12)
    ref = swt_np.c_transform_batch(img, (lo, hi), 2)
    got = run_hip(img, (lo, hi), 2, True).cpu().numpy()
    assert np.abs(got - ref).max() <= tol(2)


def test_golden_vectors(golden_dir):
    g = np.load(f"{golden_dir}/swt_golden.npz")
    names = sorted({k.split("/")[0] for k in g.files if k.endswith("/img")})
    for n in names:
        wl = bytes(g[n + "/wavelet"]).decode()
        lev = int(g[n + "/meta"][0])
        got = run_hip(g[n + "/img"], wl, lev, True).cpu().numpy()
        assert np.abs(got - g[n + "/out"]).max() <= tol(lev), n
    got = run_hip(synth.natural_images(1, 224, 224, seed=1234), "db2", 3, True).cpu().numpy()
    assert np.abs(got.reshape(-1)[::9973] - g["db2_l3_224/samples"]).max() <= tol(3)


def test_plugin_call_on_pil_image_matches_reference_contract():
    from PIL import Image
    arr = synth.natural_images(1, 224, 224, seed=21)[0]
    out = SWTTransform(level=1, wavelet="haar")(Image.fromarray(arr))
    assert isinstance(out, torch.Tensor) and out.dtype == torch.float32 and tuple(out.shape) == (3, 4, 224, 224)
    assert not out.is_cuda and torch.isfinite(out).all()
    assert np.abs(out.numpy() - swt_np.transform_image(arr, "haar", 1)).max() <= tol(1)
    # fix_size: 225x30 at level 3 is resized (BICUBIC) to 232x32 before the transform
    odd = Image.fromarray(synth.natural_images(1, 30, 225, seed=22)[0])
    t = SWTTransform(level=3, wavelet="db2")
    out = t(odd)
    assert tuple(out.shape) == (3, 4, 32, 232)
    sized = np.array(swt_np.fix_size(odd, 3))
    assert np.abs(out.numpy() - swt_np.transform_image(sized, "db2", 3)).max() <= tol(3)
    raw = RawStackTransform(level=1, copies=4)(Image.fromarray(arr))
    np.testing.assert_array_equal(raw.numpy(), swt_np.transform_image(arr, mode="raw"))


def test_size_not_multiple_raises_like_pywt():
    with pytest.raises(ValueError):
        swt2d(torch.zeros(1, 3, 12, 16, dtype=torch.uint8, device="cuda"), "haar", 3)


def test_full_size_properties_b256_db2_l3():
    """BASELINE c1 shape at a size the oracle would take minutes for: size-independent checks."""
    B = 256
    img = torch.from_numpy(synth.noise_images(B, 224, 224, seed=77)).cuda()
    y = swt2d(img, "db2", 3, channels_last=True)
    assert tuple(y.shape) == (B, 3, 4, 224, 224) and torch.isfinite(y).all()
    # zero-sum details (circular boundary, sum(dec_hi) = 0) and DC gain 2^3 on the approximation
    assert y[:, :, 1:].double().sum(dim=(-1, -2)).abs().max().item() < 2e-2
    x = img.permute(0, 3, 1, 2).double() / 255.0
    assert (y[:, :, 0].double().mean(dim=(-1, -2)) - 8.0 * x.mean(dim=(-1, -2))).abs().max().item() < 1e-4
    # shift equivariance on the full batch
    ys = swt2d(torch.roll(img, shifts=(16, -24), dims=(1, 2)), "db2", 3, channels_last=True)
    assert (torch.roll(y, shifts=(16, -24), dims=(3, 4)) - ys).abs().max().item() <= tol(3)
    # linearity: T(a) + T(b) = T(a + b) for float inputs
    a = torch.rand(4, 3, 224, 224, device="cuda")
    b = torch.rand(4, 3, 224, 224, device="cuda")
    lin = swt2d(a, "db2", 3) + swt2d(b, "db2", 3) - swt2d(a + b, "db2", 3)
    assert lin.abs().max().item() < 2e-5
    # spot-check 3 images of the batch against the oracle
    ref = swt_np.c_transform_batch(img[:3].cpu().numpy(), "db2", 3)
    assert np.abs(y[:3].cpu().numpy() - ref).max() <= tol(3)
    # determinism
    assert torch.equal(y, swt2d(img, "db2", 3, channels_last=True))


def test_bf16_output_is_rounded_fp32():
    img = synth.natural_images(2, 64, 64, seed=31)
    f32 = run_hip(img, "db2", 2, True)
    b16 = run_hip(img, "db2", 2, True, out_dtype=torch.bfloat16)
    assert b16.dtype == torch.bfloat16 and torch.equal(b16, f32.to(torch.bfloat16))


def test_rawstack_batched():
    img = torch.from_numpy(synth.noise_images(3, 16, 20, seed=41)).cuda()
    y = rawstack(img, copies=4, channels_last=True)
    # numpy's IEEE x/255 (torch on the GPU multiplies by a reciprocal: 1 ulp off)
    ref = img.cpu().numpy().astype(np.float32).transpose(0, 3, 1, 2) / 255.0
    assert np.array_equal(y.cpu().numpy(), np.repeat(ref[:, :, None], 4, axis=2))


def test_haar_level1_kernel_against_reference_made_lifting_vectors():
    """c0's transform (haar, level 1) tied to numbers the reference itself produced: its decimated Haar lifting output
    (tests/golden/lifting_golden.npz) is the stationary transform at even shifts, up to the lifting scale factors
    (see tests/test_oracle_lifting.py)."""
    import os
    gold = np.load(os.path.join(os.path.dirname(__file__), "golden", "lifting_golden.npz"))
    name = "haar_224"
    shape, seed = tuple(gold[f"{name}/shape"]), int(gold[f"{name}/seed"])
    x = torch.randn(shape, generator=torch.Generator().manual_seed(seed))
    y = swt2d(x.cuda(), "haar", 1).cpu().numpy()[0, 0]            # [4, 224, 224]
    ll, hi = gold[f"{name}/l0/ll"][0, 0], gold[f"{name}/l0/hi"][0, 0]
    ev = y[:, 0::2, 0::2]
    t = 2e-6 * float(x.abs().max())
    assert np.abs(ev[0] - 2 * ll).max() < t and np.abs(ev[1] + hi[0]).max() < t
    assert np.abs(ev[2] + hi[1]).max() < t and np.abs(ev[3] - hi[2] / np.sqrt(2)).max() < t


def test_placed_output_buffer_is_a_plain_result_buffer():
    """swt2d_place_output: the probe returns one of its candidate buffers, of the transform's shape, holding a valid result."""
    from wvhash.transforms import swt2d_place_output
    img = torch.from_numpy(synth.natural_images(8, 64, 64, seed=5)).cuda()
    buf, info = swt2d_place_output(img, "db2", 2, channels_last=True, candidates=3, launches=2)
    assert tuple(buf.shape) == (8, 3, 4, 64, 64) and info["candidates"] == 3 and len(info["probe_ms"]) == 3
    assert 0 <= info["picked"] < 3 and min(info["probe_ms"]) == info["probe_ms"][info["picked"]]
    assert torch.equal(buf, swt2d(img, "db2", 2, channels_last=True))
    bm, info1 = swt2d_place_output(img, "haar", 1, channels_last=True, band_major=True, candidates=1)
    assert tuple(bm.shape) == (4, 8, 3, 64, 64) and info1["probe_ms"] == []


@pytest.mark.parametrize("name,lev", [("db3", 2), ("db6", 1), ("db10", 1)])
def test_computed_daubechies_names_run_through_the_generic_kernels(name, lev):
    """db3, db5 ... db10 (computed taps, wvhash/transforms/wavelets.py) have no tiled / sliding instantiation: the generic
    kernels take them; against the oracle on the same taps."""
    from wvhash.transforms import get_filters
    img = synth.natural_images(2, 64, 64, seed=len(name) + lev)
    ref = swt_np.c_transform_batch(img, get_filters(name), lev)
    got = run_hip(img, name, lev, False).cpu().numpy()
    assert got.shape == (2, 3, 4, 64, 64) and np.abs(got - ref).max() <= tol(lev)
