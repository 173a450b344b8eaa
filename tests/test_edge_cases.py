"""Edge cases of the path: empty / invalid depth, ragged image sizes, camera inside geometry, depth outside
the view frustum, repeated identical frames (weight saturation), very small and very large voxels."""
import numpy as np
import pytest

import itm_testlib as T
from infinitam_amd import capi, synth
from itm_testlib import Scenario


class DepthOverride(Scenario):
    """Scenario whose depth frames come from a callable."""
    def __init__(self, fn, **kw):
        super().__init__(**kw)
        object.__setattr__(self, "_fn", fn)

    def depth(self, k):
        return np.ascontiguousarray(self._fn(self, k).astype(np.float32))


def _invalid(sc, k):
    return np.full((sc.h, sc.w), -1.0, np.float32)


def _zeros(sc, k):
    return np.zeros((sc.h, sc.w), np.float32)


def _out_of_frustum(sc, k):
    d = synth.depth_frame(sc.w, sc.h, sc.position(k), sc.intr())
    d[:, : sc.w // 2] = 5.0        # beyond viewFrustum_max
    d[: sc.h // 3, :] = 0.2        # closer than viewFrustum_min
    return d


def _holes_and_first_empty(sc, k):
    if k == 0:
        return _invalid(sc, k)
    d = synth.depth_frame(sc.w, sc.h, sc.position(k), sc.intr())
    d[::7, ::5] = -1.0
    d[10:30, 40:90] = 0.0
    return d


CASES = [
    DepthOverride(_invalid, name="all_invalid", w=160, h=120, voxelSize=0.01, frames=2),
    DepthOverride(_zeros, name="all_zero", w=160, h=120, voxelSize=0.01, frames=2),
    DepthOverride(_out_of_frustum, name="out_of_frustum", w=160, h=120, voxelSize=0.01, frames=2),
    DepthOverride(_holes_and_first_empty, name="holes_first_frame_empty", w=160, h=120, voxelSize=0.01, frames=3),
    Scenario(name="ragged_161x123", w=161, h=123, voxelSize=0.01, frames=2),
    Scenario(name="ragged_37x29", w=37, h=29, voxelSize=0.02, frames=2),
    Scenario(name="tiny_17x9", w=17, h=9, voxelSize=0.02, frames=2),
    Scenario(name="weight_saturation", w=80, h=60, voxelSize=0.02, frames=6, maxW=3),
    Scenario(name="stop_at_max_hash", w=80, h=60, voxelSize=0.02, frames=6, maxW=3, stopIntegratingAtMaxW=True),
    Scenario(name="large_voxels", w=160, h=120, voxelSize=0.04, mu=0.08, frames=2),
    Scenario(name="small_voxels_many_steps", w=80, h=60, voxelSize=0.002, mu=0.04, frames=2),   # ~6 ray steps per pixel
    Scenario(name="yaw_noise_colour", w=160, h=120, voxelSize=0.01, frames=3, trajectory="yaw", noise_seed=99,
             voxelType=T.VOXEL_S_RGB, colour=True),
]


def _static(sc_frames_same):
    return sc_frames_same


@pytest.mark.parametrize("sc", CASES, ids=lambda s: s.name)
def test_oracle_matches_reference_on_edge_cases(oracle, reference, sc):
    T.compare_results(T.run_scenario(oracle, sc), T.run_scenario(reference, sc), sc)


@pytest.mark.gpu
@pytest.mark.parametrize("sc", CASES, ids=lambda s: s.name)
def test_hip_matches_oracle_on_edge_cases(hip, oracle, sc):
    T.compare_results(T.run_scenario(hip, sc), T.run_scenario(oracle, sc), sc)
    T.compare_results(T.run_scenario(hip, sc, fused=True), T.run_scenario(oracle, sc), sc, what=sc.name + "/fused")


def test_argument_validation(oracle):
    """Error behaviour of the ABI: bad configurations are refused with ITM_ERR_INVALID, never a crash."""
    with pytest.raises(capi.ItmError):
        oracle.create_scene(voxelType=9)
    with pytest.raises(capi.ItmError):
        oracle.create_scene(bucketNum=1000)            # not a power of two
    s = oracle.create_scene(capi.VOXEL_S, capi.INDEX_HASH, capi.default_params())
    rs = s.vis.CreateRenderState((64, 48))
    d = oracle.to_backend(np.ones((48, 32), np.float32))
    with pytest.raises(capi.ItmError):
        s.reco.AllocateSceneFromDepth(capi.View(d, 32, 48), rs)   # view / render state size mismatch


@pytest.mark.gpu
def test_argument_validation_hip(hip):
    with pytest.raises(capi.ItmError):
        hip.create_scene(voxelType=9)
    with pytest.raises(capi.ItmError):
        hip.create_scene(bucketNum=1000)
    with pytest.raises(capi.ItmError):
        hip.create_scene(excessNum=12345)              # table size must be a multiple of 8
    s = hip.create_scene(capi.VOXEL_S_RGB, capi.INDEX_HASH, capi.default_params())
    s.reco.ResetScene()
    rs = s.vis.CreateRenderState((64, 48))
    d = hip.to_backend(np.ones((48, 64), np.float32))
    s.reco.AllocateSceneFromDepth(capi.View(d, 64, 48), rs)
    with pytest.raises(capi.ItmError):
        s.reco.IntegrateIntoScene(capi.View(d, 64, 48), rs)       # colour voxels without an rgb image
    other = hip.create_scene(capi.VOXEL_S, capi.INDEX_HASH, capi.default_params())
    with pytest.raises(capi.ItmError):
        other.reco.AllocateSceneFromDepth(capi.View(d, 64, 48), rs)  # render state of another scene
