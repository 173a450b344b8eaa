"""The C++ host side above the C-ABI (include/itm_hip_engines.hpp): compiles and links against
libitmhip.so everywhere; on a GPU it runs two frames and must agree with the oracle."""
import json
import os
import subprocess

import numpy as np
import pytest

import itm_testlib as T
from infinitam_amd import capi

SRC = os.path.join(T.ROOT, "tests", "cpp", "engine_adapter_demo.cpp")
EXE = os.path.join(T.ROOT, "tests", "cpp", "engine_adapter_demo")


def build_demo():
    import infinitam_amd
    lib = infinitam_amd.lib_path()
    if not os.path.exists(lib):
        infinitam_amd.build()
    cmd = ["g++", "-std=c++14", "-O1", "-I", os.path.join(T.ROOT, "include"), SRC, "-o", EXE,
           "-L", os.path.dirname(lib), "-l:libitmhip.so", "-Wl,-rpath," + os.path.dirname(lib), "-Wl,-rpath,/opt/rocm/lib"]
    subprocess.run(cmd, check=True, capture_output=True)
    return EXE


def test_adapter_compiles_and_links():
    assert os.path.exists(build_demo())


def oracle_expectation():
    ob = T.oracle_backend()
    W, H, P = 160, 120, 160 * 120
    s = ob.create_scene(capi.VOXEL_S, capi.INDEX_HASH, capi.default_params(voxelSize=0.01))
    s.reco.ResetScene()
    rs = s.vis.CreateRenderState((W, H))
    depth = ob.to_backend(np.full((H, W), 1.5, np.float32))
    pts = capi.DevBuffer(ob, P * 16, np.float32, (P, 4)); nrm = capi.DevBuffer(ob, P * 16, np.float32, (P, 4))
    for k in range(2):
        M = np.eye(4, dtype=np.float32); M[0, 3] = np.float32(-0.01) * np.float32(k)
        v = capi.View(depth, W, H, M_d=np.ascontiguousarray(M.T).reshape(16), intr_d=(145.0, 145.0, 80.0, 60.0))
        s.process_frame(v, rs, pts, nrm)
    c = s.counters(rs)
    p = pts.numpy()
    ok = p[:, 3] > 0
    m = capi.Mesh(s)
    m.MeshScene()
    tri = m.triangles()
    return {"triangles": int(tri.shape[0]), "maxTriangles": m.info()[1], "tri_sum_z": float(tri.reshape(-1, 3)[:, 2].astype(np.float64).sum()),
            "lastFreeBlockId": c["lastFreeBlockId"], "noVisibleEntries": c["noVisibleEntries"], "valid": int(ok.sum()),
            "sum_x": float(p[ok, 0].astype(np.float64).sum()), "sum_z": float(p[ok, 2].astype(np.float64).sum())}


@pytest.mark.gpu
def test_adapter_matches_oracle():
    exe = build_demo()
    out = subprocess.run([exe], check=True, capture_output=True, text=True).stdout
    got = json.loads(out.strip().splitlines()[-1])
    want = oracle_expectation()
    assert got["age"] == 0   # -1 -> -2 -> 0, ITMTrackingController.cpp:37-38
    assert got["triangles"] > 1000
    for k in ("lastFreeBlockId", "noVisibleEntries", "valid", "triangles", "maxTriangles"):
        assert got[k] == want[k], (k, got, want)
    for k in ("sum_x", "sum_z", "tri_sum_z"):
        assert abs(got[k] - want[k]) <= 1e-6 * max(1.0, abs(want[k])), (k, got, want)
