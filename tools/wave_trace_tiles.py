import numpy as np, sys
st = np.load('gpurun_out/wave_stats.npy').astype(float); tr = np.load('gpurun_out/wave_trace.npy')
for slot in [int(a) for a in sys.argv[1:]]:
    print("tile", 1066 + slot // 4, "wave", slot % 4)
    for i in range(64):
        t0, tA, tB, info = [int(v) for v in tr[slot, i]]
        if info == 0 and i > 0: break
        print("  o %2d lanes %2d march %2d tri %2d inner %d start %7d near %6d tri %6d" % (i, info & 255, (info >> 8) & 255, (info >> 16) & 255, info >> 24, t0, tA - t0, tB - tA))
