#!/usr/bin/env python3
"""Offline view of gpurun_out/wave_stats.npy + wave_trace.npy written by tools/wave_stats.py for a while-while build."""
import numpy as np, sys
st = np.load('gpurun_out/wave_stats.npy').astype(float); tr = np.load('gpurun_out/wave_trace.npy')
tot, outer = st[:4800, 0], st[:4800, 1]
print("max total", tot.max(), "mean", tot.mean(), "max outer", outer.max())
for wv in [int(a) for a in sys.argv[1:]] or [1089]:
    k = wv >> 5; n = int(min(outer[wv], 64)); print("wave", wv, "total", int(tot[wv]), "outer", int(outer[wv]))
    for i in range(n):
        t0, tA, tB, info = [int(v) for v in tr[k, i]]
        print("  o %2d lanes %2d march %2d tri %2d inner %d start %7d A %6d B %6d" % (i, info & 255, (info >> 8) & 255, (info >> 16) & 255, info >> 24, t0, tA - t0, tB - tA))
