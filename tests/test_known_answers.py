"""Known answers of the reference recorded in SURVEY.md section 8c / Appendix D (produced there with the reference's own
ITMMainEngine on the fork's defaults: 640x480, ITMVoxel_s, 5 mm, mu 0.02, parity trajectory frames 0-4): the oracle
(CPU, here) and the HIP path (GPU) must reproduce every number."""
import numpy as np
import pytest

import itm_testlib as T
from infinitam_amd.capi import BUF_RAYCAST_IMAGE, BUF_VOXEL_BLOCKS

SC = T.Scenario(name="survey_8c", voxelSize=0.005, mu=0.02, frames=5)
LAST_FREE = [62420, 61906, 61848, 61834, 61824]


def check(be, fused):
    ses = T.Session(be, SC)
    free = []
    for k in range(SC.frames):
        ses.frame(k, fused=fused)
        free.append(ses.scene.counters(ses.rs)["lastFreeBlockId"])
    assert free == LAST_FREE
    vox = ses.scene.download(BUF_VOXEL_BLOCKS)          # structured: sdf (int16), w_depth (uint8)
    fused_mask = vox["w_depth"] > 0
    assert int(fused_mask.sum()) == 1521925
    assert int(vox["sdf"][fused_mask].astype(np.int64).sum()) == 13851885396
    img = ses.scene.download(BUF_RAYCAST_IMAGE, ses.rs)
    assert int(img.reshape(-1, 4)[:, 0].astype(np.int64).sum()) == 70809109
    pts = ses.points.numpy().reshape(-1, 4)
    hit = pts[:, 3] > 0
    assert int(hit.sum()) == 298472
    assert abs(float(pts[hit, 2].astype(np.float64).mean()) - 1.8850) < 5e-5
    ses.close()


def test_oracle_reproduces_the_survey_run(oracle):
    check(oracle, fused=False)


@pytest.mark.gpu
@pytest.mark.parametrize("fused", [False, True])
def test_hip_reproduces_the_survey_run(hip, fused):
    check(hip, fused)
