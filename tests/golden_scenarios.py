"""Scenarios that have committed golden vectors (tests/golden/<name>.npz/.json, generated from the
reference by tests/golden/make_golden.py) and the summary / comparison code shared by the generator
and the tests."""
import hashlib
import json
import os

import numpy as np

import itm_testlib as T
from itm_testlib import Scenario

# (scenario, full): full=True stores whole maps, otherwise stride-4 samples + SHA-256 digests
GOLDEN_SCENARIOS = [
    (Scenario(name="g_micro_hash_s", w=160, h=120, voxelSize=0.01, frames=3), True),
    (Scenario(name="g_micro_hash_f_rgb", w=160, h=120, voxelSize=0.01, frames=2, voxelType=T.VOXEL_F_RGB, colour=True), True),
    (Scenario(name="g_micro_hash_s_rgb", w=160, h=120, voxelSize=0.01, frames=2, voxelType=T.VOXEL_S_RGB, colour=True), True),
    (Scenario(name="g_micro_hash_f", w=160, h=120, voxelSize=0.01, frames=2, voxelType=T.VOXEL_F), True),
    (Scenario(name="g_micro_dense_s", w=160, h=120, voxelSize=0.01, frames=2, indexType=T.INDEX_DENSE,
              denseSize=(64, 64, 64), denseOffset=(-32, -32, 95)), True),
    (Scenario(name="g_vga_hash_s_5mm", frames=3), False),
    (Scenario(name="g_vga_hash_s_4mm_bench", voxelSize=0.004, frames=3, trajectory="bench"), False),
]


def sha(a) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def summarise(res: T.RunResult, sc: Scenario, full: bool):
    """-> (arrays for the .npz, metadata for the .json)"""
    arrays, meta = {}, {"scenario": sc.name, "full": full}
    if res.hash is not None:
        used = np.nonzero(res.hash["ptr"] >= 0)[0].astype(np.int32)
        arrays["hash_slots"] = used
        arrays["hash_pos"] = res.hash["pos"][used]
        arrays["hash_ptr"] = res.hash["ptr"][used]
        arrays["hash_offset"] = res.hash["offset"][used]
        nv = res.counters[-1]["noVisibleEntries"]
        arrays["visible_ids"] = res.visible_ids[:nv]
        meta["visible_type_sha256"] = sha(res.visible_type)
        meta["hash_offset_nonzero"] = int((res.hash["offset"] != 0).sum())
    vox = res.voxels
    touched = np.nonzero(vox["w_depth"] > 0)[0]
    meta["voxels_touched"] = int(len(touched))
    meta["voxel_field_sha256"] = {n: sha(vox[n]) for n in vox.dtype.names}
    meta["sdf_sum_touched"] = float(vox["sdf"][touched].astype(np.float64).sum())
    if full:
        arrays["vox_idx"] = touched.astype(np.int32)
        for n in vox.dtype.names:
            arrays["vox_" + n] = vox[n][touched]
    else:
        sample = touched[:: max(1, len(touched) // 4096)][:4096]
        arrays["vox_idx"] = sample.astype(np.int32)
        for n in vox.dtype.names:
            arrays["vox_" + n] = vox[n][sample]
    arrays["range_region"] = T.range_region(res.range_image, sc.w, sc.h)
    maps = {"raycast": res.raycast, "points": res.points, "normals": res.normals, "image": res.image}
    # rays that found nothing keep an unspecified xyz in the reference; only w is compared for them
    ray = res.raycast.copy()
    ray[ray[..., 3] <= 0, :3] = 0
    maps["raycast"] = ray
    for k, m in maps.items():
        meta[k + "_sha256"] = sha(m)
        arrays[k] = m if full else m[::4, ::4]
    return arrays, meta


def load_golden(name):
    path = os.path.join(T.GOLDEN_DIR, name)
    with open(path + ".json") as f:
        meta = json.load(f)
    return np.load(path + ".npz"), meta


def check_against_golden(res: T.RunResult, sc: Scenario, full: bool):
    """Bit-exact comparison of a run with the committed reference vectors."""
    gold, meta = load_golden(sc.name)
    arrays, m = summarise(res, sc, full)
    for got, want in zip(res.counters, meta["counters"]):
        for k, v in want.items():
            assert got[k] == v, f"{sc.name}: counter {k} {got[k]} != {v}"
    for key in ("voxels_touched", "sdf_sum_touched", "voxel_field_sha256", "raycast_sha256", "points_sha256",
                "normals_sha256", "image_sha256", "visible_type_sha256", "hash_offset_nonzero"):
        if key in meta:
            assert m[key] == meta[key], f"{sc.name}: {key} differs: {m[key]} vs {meta[key]}"
    for k in gold.files:
        assert k in arrays, f"{sc.name}: missing {k}"
        assert arrays[k].shape == gold[k].shape, f"{sc.name}: {k} shape {arrays[k].shape} vs {gold[k].shape}"
        assert np.array_equal(arrays[k], gold[k]), f"{sc.name}: {k} differs from the reference golden"
    # generator drift check on the inputs
    for k, want in enumerate(meta["depth_sha256"]):
        assert sha(sc.depth(k)) == want, f"{sc.name}: synthetic depth frame {k} drifted"
