"""Direct C-ABI calls on the GPU (no Python op wrappers): shapes and code paths the wrappers never take --
unaligned distance rows, 96/192/256-bit codes, tiny and ragged sizes, every SWT implementation pinned by
WV_SWT_PATH, stream argument -- each checked against the oracle."""
import ctypes
import os

import numpy as np
import pytest
import torch

from oracle import ranking, swt_np
from wvhash import _lib, synth
from wvhash.engine import hamming as H

pytestmark = pytest.mark.gpu


def sp():
    return _lib.stream_ptr()


@pytest.mark.parametrize("Q,N,nbits", [(5, 1001, 64), (3, 4097, 128), (7, 333, 192), (4, 1025, 250), (2, 17, 16)])
def test_hamming_dist_unaligned_pitch_and_wide_codes(Q, N, nbits):
    lib = _lib.require_gpu()
    q, r = synth.random_codes(Q, N, nbits, seed=N)
    qp, rp = H.pack_codes(q.cuda()), H.pack_codes(r.cuda())
    words = qp.shape[1]
    ref = ranking.hamming_matrix_u8(q, r)
    for ld in (N, N + 3, (N + 63) // 64 * 64):                  # odd pitches take the byte-store path
        out = torch.full((Q, ld), 255, dtype=torch.uint8, device="cuda")
        rc = lib.wv_hamming_dist(_lib.ptr(qp), _lib.ptr(rp), _lib.ptr(out), ld, Q, N, words, sp())
        assert rc == 0, lib.wv_last_error()
        torch.cuda.synchronize()
        assert torch.equal(out[:, :N].cpu().long(), ref)
        assert (out[:, N:] == 255).all()                       # padding untouched
        prep = H.PreparedDB(rp, nbits)
        out.fill_(255)
        rc = lib.wv_hamming_dist_prepared(_lib.ptr(qp), _lib.ptr(prep.blob), _lib.ptr(out), ld, Q, N, words, sp())
        assert rc == 0, lib.wv_last_error()
        assert torch.equal(out[:, :N].cpu().long(), ref) and (out[:, N:] == 255).all()


def test_side_stream_is_honoured():
    lib = _lib.require_gpu()
    q, r = synth.random_codes(64, 5000, 64, seed=1)
    qp, rp = H.pack_codes(q.cuda()), H.pack_codes(r.cuda())
    s = torch.cuda.Stream()
    torch.cuda.synchronize()
    with torch.cuda.stream(s):
        idx, d = H.hamming_topk(qp, rp, 64, 100)
    s.synchronize()
    ref_idx, ref_d = ranking.hamming_topk_stable(q, r, 100)
    assert torch.equal(idx.cpu().long(), ref_idx) and torch.equal(d.cpu().long(), ref_d)


@pytest.mark.parametrize("path", ["slide", "fused", "tiled", "generic"])
@pytest.mark.parametrize("wl,lev,H_,W_", [("db2", 3, 224, 224), ("haar", 1, 224, 224), ("db2", 3, 64, 96)])
def test_every_swt_implementation_agrees_with_the_oracle(path, wl, lev, H_, W_, diag):
    from wvhash.transforms import swt2d
    diag.setenv("WV_SWT_PATH", path)
    img = synth.natural_images(2, H_, W_, seed=lev + W_)
    ref = swt_np.c_transform_batch(img, wl, lev)
    x = torch.from_numpy(img).cuda()
    for cl in (True, False):
        xin = x if cl else x.permute(0, 3, 1, 2).contiguous()
        got = swt2d(xin, wl, lev, channels_last=cl)
        assert np.abs(got.cpu().numpy() - ref).max() <= 4e-6 * 2 ** lev, (path, cl)
    f = (x.float() / 255.0).permute(0, 3, 1, 2).contiguous()    # fp32 planar input
    got = swt2d(f, wl, lev)
    assert np.abs(got.cpu().numpy() - ref).max() <= 4e-6 * 2 ** lev + 3e-7 * 2 ** lev


def test_swt_into_preallocated_output_and_bad_output():
    from wvhash.transforms import swt2d
    img = torch.from_numpy(synth.noise_images(3, 224, 224, seed=4)).cuda()
    out = torch.empty((3, 3, 4, 224, 224), device="cuda")
    y = swt2d(img, "db2", 3, channels_last=True, out=out)
    assert y.data_ptr() == out.data_ptr()
    assert torch.equal(y, swt2d(img, "db2", 3, channels_last=True))
    with pytest.raises(ValueError):
        swt2d(img, "db2", 3, channels_last=True, out=torch.empty((3, 3, 4, 224, 220), device="cuda"))


def test_error_codes_on_the_gpu():
    lib = _lib.require_gpu()
    q = torch.zeros((2, 1), dtype=torch.int64, device="cuda")
    idx = torch.zeros((2, 4), dtype=torch.int32, device="cuda")
    assert lib.wv_hamming_topk(_lib.ptr(q), _lib.ptr(q), _lib.ptr(idx), None, 2, 2, 64, 4, 0, _lib.ptr(q), 8, sp()) == -22
    assert lib.wv_map_at_k(_lib.ptr(idx), 2, 0, _lib.ptr(q), _lib.ptr(q), 1, _lib.ptr(idx), None, sp()) == -22
    assert lib.wv_knn_float(_lib.ptr(q), _lib.ptr(q), 2, 2, 6, 7, 1, _lib.ptr(idx), _lib.ptr(idx), _lib.ptr(q), 8, sp()) == -22
    assert b"metric" in lib.wv_last_error()
    assert lib.wv_knn_float(_lib.ptr(q), _lib.ptr(q), 2, 2, 6, 0, 1, _lib.ptr(idx), _lib.ptr(idx), _lib.ptr(q), 8, sp()) == -12   # workspace


def test_native_consumer_without_torch_or_python():
    """tests/native/cabi_smoke: a plain HIP-runtime program linking libwvhash.so -- the boundary carries no torch
    types.  Built by __graft_entry__.build() (make -C tests/native); checks SWT, ranking and AP on the host."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "tests", "native", "cabi_smoke")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.join(root, "tests", "native")])
    res = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert res.returncode == 0, res.stdout + res.stderr
    assert "cabi_smoke ok" in res.stdout
