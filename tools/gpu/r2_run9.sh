#!/bin/bash
# round 2, GPU call 9: parity suite (meshing, checkpoint validation, free-view scratch), bench, mesh timing
cd "$(dirname "$0")/../.."
R=$PWD; O=gpurun_out/r2i; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -30 $O/pytest.log
python bench.py > $O/bench_n1.json 2> $O/bench_n1.err
cut -c1-200 $O/bench_n1.json
python - <<'PY' > $O/mesh_timing.txt 2>&1
import sys, time
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import itm_testlib as T
from infinitam_amd.capi import Mesh
hip = T.hip_backend()
sc = T.Scenario(name="m", voxelSize=0.004, frames=20, trajectory="bench", localBlockNum=0x40000)
ses = T.Session(hip, sc)
for k in range(sc.frames): ses.frame(k, fused=True)
m = Mesh(ses.scene)
for i in range(3):
    hip.sync(); t0 = time.perf_counter(); m.MeshScene(); hip.sync(); print("MeshScene ms", round((time.perf_counter() - t0) * 1e3, 3), m.info())
PY
cat $O/mesh_timing.txt
