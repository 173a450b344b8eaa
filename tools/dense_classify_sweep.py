#!/usr/bin/env python3
"""Seeded sweep of the dense integration's classification (integrate.hip: classify_group, strip kernel): random volumes, voxel sizes,
band widths, intrinsics and 6-DoF poses over smooth depth images (sphere + wall, random holes, an occasional NaN / zero region).
Per case and frame (a) the check mode -- classify every group, run the exact path anyway, count disagreements -- must report none,
and (b) the strip kernel's volume must equal the volume of the unclassified exact path, bit for bit.  GPU against GPU: no oracle
needed, so hundreds of cases run in a minute.  usage: python tools/dense_classify_sweep.py [first=0] [count=200]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import infinitam_amd as itm  # noqa: E402
from infinitam_amd import capi, synth  # noqa: E402
from test_dense_cull import pose, rotation  # noqa: E402

F = np.float32
first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 200
be = itm.load()


def case(seed):
    rng = np.random.default_rng(seed)
    W, H = [(160, 120), (320, 240), (96, 72), (200, 152)][seed % 4]
    n = int(rng.choice([64, 96, 128]))
    vs = float(rng.choice([0.004, 0.008, 0.012, 0.02]))
    mu = float(vs * rng.choice([2.0, 4.0, 5.0, 8.0]))
    f = float(rng.uniform(0.7, 1.4) * W * 0.9)
    intr = (F(f), F(f * rng.uniform(0.9, 1.1)), F(W / 2 + rng.uniform(-8, 8)), F(H / 2 + rng.uniform(-8, 8)))
    # the volume somewhere in front of the scene's sphere (centre (0, 0, 1.5)); its offset in voxels
    ext = n * vs
    centre = np.array([rng.uniform(-0.3, 0.3), rng.uniform(-0.3, 0.3), rng.uniform(0.9, 1.9)])
    off = tuple(int(round((centre[k] - ext / 2) / vs)) for k in range(3))
    frames = []
    for k in range(4):
        ang = rng.uniform(-0.5, 0.5, 3) * rng.choice([1.0, 0.3, 0.0])
        pos = rng.uniform(-0.25, 0.25, 3) * np.array([1, 1, 0.6])
        t = (F(pos[0]), F(pos[1]), F(pos[2]))
        d = synth.depth_frame(W, H, t, intr).astype(F)          # rendered for the translated camera; the rotation only enters the pose
        kind = rng.integers(0, 6)
        if kind == 0:
            d[rng.random(d.shape) < 0.01] = F(-1.0)
        elif kind == 1:
            y0, x0 = int(rng.integers(0, H - 20)), int(rng.integers(0, W - 20)); d[y0:y0 + 20, x0:x0 + 20] = F(0.0)
        elif kind == 2:
            d[int(rng.integers(0, H)), int(rng.integers(0, W))] = np.nan
        elif kind == 3:
            d += rng.normal(0, 0.003, d.shape).astype(F)
        frames.append((pose(rotation(*ang), pos), np.ascontiguousarray(d)))
    return dict(W=W, H=H, n=n, vs=vs, mu=mu, intr=intr, off=off, frames=frames, maxW=int(rng.choice([2, 3, 100])), stop=bool(rng.integers(0, 2)))


def run(c, key16, key17):
    be.check(be.fn["debug_set"](16, key16), "debug_set"); be.check(be.fn["debug_set"](17, key17), "debug_set")
    prm = capi.default_params(c["vs"], c["mu"], c["maxW"], 0.35, 3.0, c["stop"])
    s = be.create_scene(capi.VOXEL_S, capi.INDEX_DENSE, prm, denseSize=(c["n"],) * 3, denseOffset=c["off"])
    s.reco.ResetScene()
    rs = s.vis.CreateRenderState((c["W"], c["H"]))
    vols, checks = [], []
    cc = (C.c_int32 * 4)()
    for M, d in c["frames"]:
        if key16 == 3:
            be.check(be.fn["debug_dense_classify_check"](cc, 1), "check")
        s.reco.IntegrateIntoScene(capi.View(be.to_backend(d), c["W"], c["H"], M_d=M, intr_d=c["intr"]), rs)
        if key16 == 3:
            be.check(be.fn["debug_dense_classify_check"](cc, 1), "check"); checks.append(list(cc))
        vols.append(s.download(capi.BUF_VOXEL_BLOCKS).view(np.uint32))      # (a fresh array per call; a .copy() of the structured dtype would leave the pad byte undefined)
    rs.close(); s.close()
    return vols, checks


def main():
    bad, free, shadow = [], 0, 0
    try:
        for seed in range(first, first + count):
            c = case(seed)
            exact, _ = run(c, 1, 1)                 # round-2 launch shape, no classification: the exact path for every group
            strips, _ = run(c, 0, 0)                # the shipped kernel
            _, checks = run(c, 3, 1)                # check mode
            ok = all(np.array_equal(a.view(np.uint32), b.view(np.uint32)) for a, b in zip(exact, strips)) and all(ch[3] == 0 for ch in checks)
            free += sum(ch[0] for ch in checks); shadow += sum(ch[1] for ch in checks)
            if not ok:
                bad.append(seed)
                print("seed", seed, "FAIL", [ch for ch in checks], [int(np.count_nonzero(a.view(np.uint32) != b.view(np.uint32))) for a, b in zip(exact, strips)], flush=True)
    finally:
        be.check(be.fn["debug_set"](16, 0), "debug_set"); be.check(be.fn["debug_set"](17, 0), "debug_set")
    print(f"{count} seeds from {first}: {len(bad)} failures {bad}; groups classified free {free}, shadow {shadow}")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
