#!/usr/bin/env python3
"""Diagnostic for seeds of tools/dense_classify_sweep.py: where the strip kernel's volume differs from the exact path's, with the
projection of each differing voxel recomputed on the host in float32.  usage: python tools/dense_classify_diag.py seed [seed ...]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
seeds = [int(a) for a in sys.argv[1:]] or [397]
sys.argv = [sys.argv[0]]
import dense_classify_sweep as S  # noqa: E402

F = np.float32
for seed in seeds:
    c = S.case(seed)
    exact, _ = S.run(c, 1, 1)
    strips, _ = S.run(c, 0, 0)
    old_cls, _ = S.run(c, 0, 1)       # the old launch shape WITH per-group classification
    n = c["n"]
    print(f"seed {seed}: n={n} vs={c['vs']} mu={c['mu']} W,H={c['W']},{c['H']} maxW={c['maxW']} stop={c['stop']} off={c['off']} intr={[float(v) for v in c['intr']]}")
    for k, ((M, d), a, b, o) in enumerate(zip(c["frames"], exact, strips, old_cls)):
        a = a.view(np.uint32).ravel(); b = b.view(np.uint32).ravel(); o = o.view(np.uint32).ravel()
        idx = np.flatnonzero(a != b)
        print(f" frame {k}: {idx.size} words differ (strips), {int(np.count_nonzero(a != o))} (old shape classified); M={np.asarray(M, F).ravel().tolist()}")
        prev = exact[k - 1].view(np.uint32).ravel() if k else None
        Mm = np.asarray(M, F).reshape(4, 4).T if np.asarray(M).size == 16 else None       # column-major storage -> matrix
        for i in idx[:12]:
            z, r = divmod(int(i), n * n); y, x = divmod(r, n)
            pt = (np.array([x + c["off"][0], y + c["off"][1], z + c["off"][2]], F) * F(c["vs"])).astype(F)
            pc = Mm[:3, :3] @ pt + Mm[:3, 3]
            u = c["intr"][0] * pc[0] / pc[2] + c["intr"][2]; v = c["intr"][1] * pc[1] / pc[2] + c["intr"][3]
            ui, vi = int(u + 0.5), int(v + 0.5)
            dd = d[vi, ui] if 0 <= ui < c["W"] and 0 <= vi < c["H"] else None
            print(f"   ({x},{y},{z}) exact {a[i]:#010x} strips {b[i]:#010x} before {(prev[i] if prev is not None else 0x7fff):#010x}  pc={pc.tolist()} u,v={float(u):.3f},{float(v):.3f} depth={dd} eta={None if dd is None else float(dd - pc[2])}")
