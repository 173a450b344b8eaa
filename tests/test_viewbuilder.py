"""View builder remainder: 5x5 bilateral depth filter, normal / depth-uncertainty images and the
UpdateView sequence (raw short depth -> metres -> 5 filter passes -> normals).

  * oracle vs the reference's ITMViewBuilder_CPU: bit-exact (same host libm);
  * HIP vs oracle: conversions, hole handling, borders and normals are exact; the filter weights use exp and the
    uncertainty uses acos, where the device library and glibc may differ in the last place -> filtered depth within
    2e-6 relative (five passes), sigmaZ within 1e-5 relative.
"""
import ctypes as C

import numpy as np
import pytest

import itm_testlib as T
from infinitam_amd import synth
from infinitam_amd.capi import DevBuffer

W, H = 160, 120
INTR = synth.intrinsics_for(W, H)


def fp(a):
    a = np.ascontiguousarray(np.asarray(a, np.float32).reshape(-1))
    return a, a.ctypes.data_as(C.POINTER(C.c_float))


def noisy_depth():
    d = synth.depth_frame(W, H, synth.parity_position(1), INTR).astype(np.float32)
    s = np.uint32(12345)
    noise = np.empty(W * H, np.float32)
    for i in range(W * H):     # LCG of SURVEY 8d, +-2 mm
        s = np.uint32((int(s) * 1664525 + 1013904223) & 0xffffffff)
        noise[i] = ((int(s) >> 8) / float(1 << 24) - 0.5) * 0.004
    d = d + noise.reshape(H, W)
    d[::7, ::5] = -1.0          # holes
    d[40:50, 60:90] = -1.0
    return d.astype(np.float32)


def raw_depth():
    d = noisy_depth()
    raw = np.where(d > 0, np.round(d * 1000.0), 0).astype(np.int16)
    raw[3, 3] = 32001           # rejected by the affine conversion
    return raw


def run_filter(be, img):
    src = be.to_backend(img)
    dst = DevBuffer(be, W * H * 4, np.float32, (H, W))
    be.check(be.fn["filter_depth"](src.ptr, dst.ptr, W, H, None), "filter_depth")
    return dst.numpy()


def run_normals(be, img):
    src = be.to_backend(img)
    nrm = be.to_backend(np.zeros((H, W, 4), np.float32))
    sig = be.to_backend(np.zeros((H, W), np.float32))
    _, ip = fp(INTR)
    be.check(be.fn["compute_normal_and_weights"](src.ptr, nrm.ptr, sig.ptr, W, H, ip, None), "normals")
    return nrm.numpy(), sig.numpy()


def run_update(be, raw, calib_type, c0, c1, bilateral, noise):
    src = be.to_backend(raw)
    depth = be.to_backend(np.zeros((H, W), np.float32))
    scratch = be.to_backend(np.zeros((H, W), np.float32))
    nrm = be.to_backend(np.zeros((H, W, 4), np.float32))
    sig = be.to_backend(np.zeros((H, W), np.float32))
    _, ip = fp(INTR)
    be.check(be.fn["update_view"](src.ptr, W, H, calib_type, c0, c1, ip, int(bilateral), int(noise), depth.ptr, scratch.ptr,
                                  nrm.ptr, sig.ptr, None), "update_view")
    return depth.numpy(), nrm.numpy(), sig.numpy()


def test_filter_oracle_vs_reference(oracle, reference):
    img = noisy_depth()
    a, b = run_filter(oracle, img), run_filter(reference, img)
    assert np.array_equal(a, b)
    assert (a[:2] == 0).all() and (a[:, :2] == 0).all() and (a[-2:] == 0).all()     # cleared border
    assert (a[2:-2, 2:-2][img[2:-2, 2:-2] < 0] == -1).all()


def test_normals_oracle_vs_reference(oracle, reference):
    img = noisy_depth()
    na, sa = run_normals(oracle, img)
    nb, sb = run_normals(reference, img)
    assert np.array_equal(na, nb) and np.array_equal(sa, sb)
    assert (na[..., 3] == 1).sum() > 0.5 * W * H


@pytest.mark.parametrize("calib", [(1, 0.001, 0.0), (0, 1135.09, 0.0819141)])
@pytest.mark.parametrize("bilateral,noise", [(False, False), (True, False), (True, True)])
def test_update_view_oracle_vs_reference(oracle, reference, calib, bilateral, noise):
    raw = raw_depth()
    if calib[0] == 0:   # encode the scene as Kinect disparity
        d = noisy_depth()
        disp = np.where(d > 0, calib[1] - 8.0 * calib[2] * INTR[0] / np.maximum(d, 1e-3), calib[1])
        raw = np.clip(np.round(disp), -32768, 32767).astype(np.int16)
    a = run_update(oracle, raw, calib[0], calib[1], calib[2], bilateral, noise)
    b = run_update(reference, raw, calib[0], calib[1], calib[2], bilateral, noise)
    assert np.array_equal(a[0], b[0])
    if noise:
        assert np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])


@pytest.mark.gpu
def test_filter_hip_vs_oracle(hip, oracle):
    img = noisy_depth()
    a, b = run_filter(hip, img), run_filter(oracle, img)
    assert np.array_equal(a <= 0, b <= 0)                      # holes and border identical
    m = b > 0
    assert np.abs(a[m] - b[m]).max() <= 1e-6 * np.abs(b[m]).max()


@pytest.mark.gpu
def test_normals_hip_vs_oracle(hip, oracle):
    img = noisy_depth()
    na, sa = run_normals(hip, img)
    nb, sb = run_normals(oracle, img)
    assert np.array_equal(na, nb)                              # cross product, IEEE sqrt and division only
    assert np.array_equal(sa < 0, sb < 0)
    m = sb > 0
    assert np.abs(sa[m] - sb[m]).max() <= 1e-5 * np.abs(sb[m]).max()


@pytest.mark.gpu
@pytest.mark.parametrize("bilateral,noise", [(False, False), (True, True)])
def test_update_view_hip_vs_oracle(hip, oracle, bilateral, noise):
    raw = raw_depth()
    a = run_update(hip, raw, 1, 0.001, 0.0, bilateral, noise)
    b = run_update(oracle, raw, 1, 0.001, 0.0, bilateral, noise)
    assert np.array_equal(a[0] <= 0, b[0] <= 0)
    m = b[0] > 0
    tol = 2e-6 if bilateral else 0.0
    assert np.abs(a[0][m] - b[0][m]).max() <= tol * np.abs(b[0][m]).max()
    if noise:
        # normals of the filtered depth inherit its last-place differences
        valid = (a[1][..., 3] == 1) & (b[1][..., 3] == 1)
        assert (a[1][..., 3] == b[1][..., 3]).mean() > 0.999
        assert np.abs(a[1][valid] - b[1][valid]).max() < 5e-3
