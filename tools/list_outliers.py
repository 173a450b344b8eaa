#!/usr/bin/env python3
"""Which frames of the bench trajectory make the visible-list launch long.  rocprofv3's kernel trace
(tools/gpu/r5_counters.sh -> trace_c2_summary.json) gives the duration of every launch; this prints, per frame of the same workload, what the
launch had to do: block requests seen by the sweep (noAllocRequests), blocks allocated (drop of lastFreeBlockId), visible blocks -- so that the
long launches can be attributed.  usage: python tools/list_outliers.py [trace_c2_summary.json]"""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import itm_testlib as T
sc = T.Scenario(name="outliers", voxelSize=0.004, frames=320, trajectory="bench", localBlockNum=0x40000)
ses = T.Session(T.hip_backend(), sc)
depth = {}
rows = []
last = 0x40000 - 1
for k in range(sc.frames):
    if k % 100 not in depth: depth[k % 100] = ses.be.to_backend(sc.depth(k))
    v = T.View(depth[k % 100], sc.w, sc.h, M_d=sc.pose(k), intr_d=sc.intr())
    ses.scene.process_frame(v, ses.rs, ses.points, ses.normals)
    c = ses.scene.counters(ses.rs)
    rows.append((k, c["noAllocRequests"], last - c["lastFreeBlockId"], c["noVisibleEntries"]))
    last = c["lastFreeBlockId"]
ses.close()
rows = np.array(rows)
us = None
if len(sys.argv) > 1:
    us = np.array(json.load(open(sys.argv[1]))["per_launch_us"][:320])
print("frame  requests  allocated  visible  list_us")
for k, req, alloc, nv in rows:
    if k < 20 or req > 0 or (us is not None and us[k] > 18):
        print("%5d  %8d  %9d  %7d  %s" % (k, req, alloc, nv, ("%.1f" % us[k]) if us is not None else "-"))
st = rows[120:]
print("steady state (frames 120-319): frames with requests %d of %d, max requests %d, blocks allocated %d" % ((st[:, 1] > 0).sum(), len(st), st[:, 1].max(), st[:, 2].sum()))
if us is not None:
    a = us[120:320]; r = st[:, 1]
    print("list_us with requests: %.1f (n=%d)   without: %.1f (n=%d)   corr(requests, us) %.2f   corr(visible, us) %.2f" % (
        a[r > 0].mean() if (r > 0).any() else 0, (r > 0).sum(), a[r == 0].mean() if (r == 0).any() else 0, (r == 0).sum(),
        np.corrcoef(r, a)[0, 1] if r.std() > 0 else 0, np.corrcoef(st[:, 3], a)[0, 1]))
