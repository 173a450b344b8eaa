"""On-disk formats either side of the path: 16-bit big-endian PGM depth, PPM colour, calibration text.
The product's readers / writers are checked against the reference's own Utils/FileUtils.cpp and ITMCalibIO.cpp
(through the reference shim, where /root/reference exists) in both directions, and against hand-built files."""
import os

import numpy as np
import pytest

import itm_testlib as T

W, H = 37, 23     # deliberately odd


def images():
    rng = np.random.RandomState(7)
    depth = rng.randint(-5, 32767, size=(H, W)).astype(np.int16)
    depth[0, 0] = 0x1234; depth[0, 1] = -2
    rgba = rng.randint(0, 256, size=(H, W, 4)).astype(np.uint8)
    rgba[..., 3] = 255
    return depth, rgba


def test_pgm_is_big_endian_and_round_trips(hip_host, tmp_path):
    depth, rgba = images()
    p = str(tmp_path / "d.pgm")
    hip_host.write_image(p, depth)
    raw = open(p, "rb").read()
    assert raw.startswith(b"P5\n37 23\n65535\n")
    body = raw[len(b"P5\n37 23\n65535\n"):]
    assert body[0] == 0x12 and body[1] == 0x34              # most significant byte first
    assert np.array_equal(hip_host.read_depth_image(p), depth)
    q = str(tmp_path / "c.ppm")
    hip_host.write_image(q, rgba)
    assert open(q, "rb").read().startswith(b"P6\n37 23\n255\n")
    assert np.array_equal(hip_host.read_rgb_image(q), rgba)


def test_ascii_and_rejects(hip_host, tmp_path):
    p = str(tmp_path / "a.pgm")
    open(p, "w").write("P2\n3 2\n65535\n1 2 3\n40000 5 6\n")
    got = hip_host.read_depth_image(p)
    assert got.tolist() == [[1, 2, 3], [np.int16(np.uint16(40000)), 5, 6]]
    open(p, "w").write("P5\n3 2\n255\n123456")        # 8-bit PGM is not a depth image
    with pytest.raises(Exception):
        hip_host.read_depth_image(p)
    with pytest.raises(Exception):
        hip_host.read_depth_image(str(tmp_path / "missing.pgm"))
    open(p, "wb").write(b"P5\n3 2\n65535\n\x00\x01")  # truncated
    with pytest.raises(Exception):
        hip_host.read_depth_image(p)


def test_float_depth_writer_quirk(hip_host, tmp_path):
    # SaveImageToFile(ITMFloatImage): millimetres, negative -> 0, and NO byte swap (FileUtils.cpp:305-322)
    img = np.array([[0.5, -1.0, 1.234]], np.float32)
    p = str(tmp_path / "f.pgm")
    hip_host.write_image(p, img)
    body = open(p, "rb").read().split(b"65535\n", 1)[1]
    assert np.frombuffer(body, "<u2").tolist() == [500, 0, 1234]


def test_calib_values(hip_host):
    c = hip_host.read_rgbd_calib(os.path.join(T.GOLDEN_DIR, "calib_synthetic.txt"))
    assert list(c.intr_d) == [np.float32(580.5), np.float32(581.25), np.float32(318.75), np.float32(242.5)]
    assert c.disparityType == 0 and list(c.disparityParams) == [np.float32(1090.25), np.float32(0.0778125)]
    m = np.array(c.rgb_to_depth[:], np.float32).reshape(4, 4).T     # column-major -> rows
    assert m[0, 3] == np.float32(0.0251) and m[3].tolist() == [0, 0, 0, 1]
    inv = np.array(c.rgb_to_depth_inv[:], np.float32).reshape(4, 4).T
    assert np.abs(m.astype(np.float64) @ inv.astype(np.float64) - np.eye(4)).max() < 1e-4   # calibration matrix is only ~orthonormal


def test_calib_variants(hip_host, tmp_path):
    base = open(os.path.join(T.GOLDEN_DIR, "calib_synthetic.txt")).read().rsplit("\n\n", 1)[0]
    for tail, want in (("affine 0.0002 0.01", (1, 0.0002, 0.01)), ("kinect 1090 0.08", (0, 1090.0, 0.08)), ("0 0", (1, 1.0 / 1000.0, 0.0))):
        p = str(tmp_path / "c.txt")
        open(p, "w").write(base + "\n\n" + tail + "\n")
        c = hip_host.read_rgbd_calib(p)
        assert (c.disparityType, c.disparityParams[0], c.disparityParams[1]) == (want[0], np.float32(want[1]), np.float32(want[2]))
    open(p, "w").write("640 480\n1 2\n")
    with pytest.raises(Exception):
        hip_host.read_rgbd_calib(p)


def test_against_reference_both_directions(hip_host, reference, tmp_path):
    depth, rgba = images()
    for writer, reader in ((hip_host, reference), (reference, hip_host)):
        p, q = str(tmp_path / "d.pgm"), str(tmp_path / "c.ppm")
        writer.write_image(p, depth); writer.write_image(q, rgba)
        assert np.array_equal(reader.read_depth_image(p), depth)
        assert np.array_equal(reader.read_rgb_image(q), rgba)
    a, b = str(tmp_path / "a.pgm"), str(tmp_path / "b.pgm")
    f = (np.abs(depth).astype(np.float32) / 1000.0); f[0, 0] = -1.0
    hip_host.write_image(a, f); reference.write_image(b, f)
    assert open(a, "rb").read() == open(b, "rb").read()
    hip_host.write_image(a, depth); reference.write_image(b, depth)
    assert open(a, "rb").read() == open(b, "rb").read()
    ca = hip_host.read_rgbd_calib(os.path.join(T.GOLDEN_DIR, "calib_synthetic.txt"))
    cb = reference.read_rgbd_calib(os.path.join(T.GOLDEN_DIR, "calib_synthetic.txt"))
    for f_ in ("intr_rgb", "intr_d", "rgb_to_depth", "rgb_to_depth_inv", "disparityParams"):
        assert list(getattr(ca, f_)) == list(getattr(cb, f_)), f_
    assert ca.disparityType == cb.disparityType
