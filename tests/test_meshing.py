"""Marching-cubes meshing: ITMMeshingEngine::MeshScene + ITMMesh writers.

The reference appends triangles in a defined order (table slots ascending, voxels z-y-x, case-table order) and has a defined
full-buffer behaviour, so meshes are compared as ARRAYS, bit for bit -- not as sets:
  * oracle vs the reference's ITMMeshingEngine_CPU / ITMMesh::WriteOBJ / WriteSTL: identical arrays and files (CPU);
  * HIP vs oracle (GPU): identical arrays and files, all four voxel types, the full-buffer case, dense scenes (no triangles);
  * HIP and oracle vs a committed digest of the reference's mesh (tests/golden/g_mesh.json)."""
import hashlib
import json
import os

import numpy as np
import pytest

import itm_testlib as T
from infinitam_amd.capi import Mesh
from itm_testlib import Scenario

SC = Scenario(name="mesh_micro", w=160, h=120, voxelSize=0.01, frames=3)
GOLDEN = os.path.join(T.GOLDEN_DIR, "g_mesh.json")


def build(be, sc=SC):
    ses = T.Session(be, sc)
    for k in range(sc.frames):
        ses.frame(k)
    return ses


def mesh_of(be, sc=SC, max_triangles=0):
    ses = build(be, sc)
    m = Mesh(ses.scene, max_triangles)
    m.MeshScene()
    tri, info = m.triangles(), m.info()
    return ses, m, tri, info


def digest(tri):
    return {"triangles": int(tri.shape[0]), "sha256": hashlib.sha256(np.ascontiguousarray(tri).tobytes()).hexdigest(),
            "first": np.asarray(tri[:4], np.float64).round(7).tolist(), "bbox": [np.asarray(tri.reshape(-1, 3).min(0), np.float64).round(6).tolist(),
                                                                                np.asarray(tri.reshape(-1, 3).max(0), np.float64).round(6).tolist()]}


def test_oracle_mesh_equals_reference_mesh(oracle, reference, tmp_path):
    a = mesh_of(oracle); b = mesh_of(reference)
    assert a[3] == b[3] and a[3][0] > 20000                       # (noTotalTriangles, noMaxTriangles)
    assert np.array_equal(a[2], b[2])
    # the surface is where it should be: sphere radius 0.5 at (0, 0, 1.5) and the wall z = 2.5
    pts = a[2].reshape(-1, 3).astype(np.float64)
    r = np.linalg.norm(pts - np.array([0, 0, 1.5]), axis=1)
    assert np.all((np.abs(r - 0.5) < 0.02) | (np.abs(pts[:, 2] - 2.5) < 0.02))
    for ext, fn in (("obj", "WriteOBJ"), ("stl", "WriteSTL")):
        pa, pb = str(tmp_path / ("a." + ext)), str(tmp_path / ("b." + ext))
        getattr(a[1], fn)(pa); getattr(b[1], fn)(pb)
        assert open(pa, "rb").read() == open(pb, "rb").read() and os.path.getsize(pa) > 1000
    # committed digest of the REFERENCE's mesh (regenerate with ITM_WRITE_GOLDEN=1)
    d = digest(b[2])
    if os.environ.get("ITM_WRITE_GOLDEN") == "1":
        json.dump(d, open(GOLDEN, "w"), indent=1)
    assert json.load(open(GOLDEN)) == d


@pytest.mark.parametrize("voxel", [T.VOXEL_F, T.VOXEL_S_RGB, T.VOXEL_F_RGB])
def test_oracle_mesh_equals_reference_mesh_other_voxel_types(oracle, reference, voxel):
    sc = Scenario(name="mesh_v", w=160, h=120, voxelSize=0.01, frames=2, voxelType=voxel, colour=voxel in (T.VOXEL_S_RGB, T.VOXEL_F_RGB))
    a = mesh_of(oracle, sc); b = mesh_of(reference, sc)
    assert a[3] == b[3] and np.array_equal(a[2], b[2])


def test_oracle_reproduces_the_committed_reference_digest(oracle):
    assert digest(mesh_of(oracle)[2]) == json.load(open(GOLDEN))


def test_oracle_full_buffer_and_dense(oracle):
    _, _, full, _ = mesh_of(oracle)
    ses, m, tri, (n, cap) = mesh_of(oracle, max_triangles=1000)
    assert (n, cap) == (999, 1000) and np.array_equal(tri, full[:999])      # the count stops at noMaxTriangles - 1
    dense = Scenario(name="mesh_dense", w=160, h=120, voxelSize=0.01, frames=2, indexType=T.INDEX_DENSE, denseSize=(64, 64, 64), denseOffset=(-32, -32, 95))
    assert mesh_of(oracle, dense)[3][0] == 0                                 # ITMPlainVoxelArray: MeshScene is empty in the reference


@pytest.mark.gpu
@pytest.mark.parametrize("sc", [SC, Scenario(name="mesh_vga_4mm", voxelSize=0.004, frames=3, trajectory="bench"),
                                Scenario(name="mesh_f_rgb", w=160, h=120, voxelSize=0.01, frames=2, voxelType=T.VOXEL_F_RGB, colour=True),
                                Scenario(name="mesh_s_rgb_yaw", w=320, h=240, voxelSize=0.005, frames=3, voxelType=T.VOXEL_S_RGB, colour=True, trajectory="yaw"),
                                Scenario(name="mesh_f", w=160, h=120, voxelSize=0.01, frames=2, voxelType=T.VOXEL_F)], ids=lambda s: s.name)
def test_hip_mesh_equals_oracle_mesh(hip, oracle, sc, tmp_path):
    a = mesh_of(hip, sc); b = mesh_of(oracle, sc)
    assert a[3] == b[3] and a[3][0] > 1000
    assert np.array_equal(a[2], b[2])
    if sc is SC:
        assert digest(a[2]) == json.load(open(GOLDEN))
        for ext, fn in (("obj", "WriteOBJ"), ("stl", "WriteSTL")):
            pa, pb = str(tmp_path / ("a." + ext)), str(tmp_path / ("b." + ext))
            getattr(a[1], fn)(pa); getattr(b[1], fn)(pb)
            assert open(pa, "rb").read() == open(pb, "rb").read()


@pytest.mark.gpu
def test_hip_full_buffer_dense_and_remesh(hip, oracle):
    a = mesh_of(hip, max_triangles=1000); b = mesh_of(oracle, max_triangles=1000)
    assert a[3] == b[3] == (999, 1000) and np.array_equal(a[2], b[2])
    full = mesh_of(oracle)[2]
    # a buffer with room for exactly the total + 1 holds every triangle
    n_all = full.shape[0]
    c = mesh_of(hip, max_triangles=n_all + 1)
    assert c[3][0] == n_all and np.array_equal(c[2], full)
    dense = Scenario(name="mesh_dense", w=160, h=120, voxelSize=0.01, frames=2, indexType=T.INDEX_DENSE, denseSize=(64, 64, 64), denseOffset=(-32, -32, 95))
    assert mesh_of(hip, dense)[3][0] == 0
    # meshing again after more fusion replaces the previous mesh
    ses, m, tri0, _ = mesh_of(hip)
    ses.frame(3); ses.frame(4)
    m.MeshScene()
    ref = T.Session(oracle, SC)
    for k in range(5):
        ref.frame(k)
    mo = Mesh(ref.scene); mo.MeshScene()
    assert np.array_equal(m.triangles(), mo.triangles())
