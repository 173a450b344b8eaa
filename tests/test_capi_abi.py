"""CPU-side checks of the boundary: the product library builds, loads and exports every symbol the
header declares; struct layouts agree between C and the ctypes binding; the product never touches
oracle/; known-answer values of the hash function."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

import itm_testlib as T
from golden_scenarios import load_golden
from infinitam_amd import capi


@pytest.fixture(scope="module")
def libpath():
    import infinitam_amd
    if not os.path.exists(infinitam_amd.lib_path()):
        infinitam_amd.build()
    return infinitam_amd.lib_path()


def test_library_exports_every_declared_symbol(libpath):
    out = subprocess.run(["nm", "-D", "--defined-only", libpath], capture_output=True, text=True, check=True).stdout
    exported = {line.split()[-1] for line in out.splitlines() if line.strip()}
    declared = capi.declared_functions()
    assert len(declared) >= 35
    missing = [n for n in declared if "itm_" + n not in exported]
    assert not missing, f"libitmhip.so does not export: {missing}"
    be = capi.Backend(libpath, "itm_")          # binds every symbol with its signature; no compute calls
    assert be.on_device and "gfx950" in be.version()
    assert set(declared) == set(capi._SIGS) | set(capi._HOST_IO_SIGS), "ctypes binding and header disagree on the function set"
    for t, nbytes in ((capi.VOXEL_S, 4), (capi.VOXEL_F, 8), (capi.VOXEL_S_RGB, 8), (capi.VOXEL_F_RGB, 12)):
        assert be.fn["voxel_size_bytes"](t) == nbytes == capi.VOXEL_DTYPES[t].itemsize


def test_struct_layouts_match_the_header(tmp_path):
    src = tmp_path / "sizes.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "itm_hip.h"\nint main(void){printf("%zu %zu %zu %zu %zu %zu %zu %zu\\n",'
                   'sizeof(itm_scene_params), sizeof(itm_scene_config), sizeof(itm_view), sizeof(itm_counters), sizeof(itm_profile), sizeof(itm_rgbd_calib),'
                   'offsetof(itm_view, M_d), offsetof(itm_view, rgb_to_depth_inv)); return 0;}\n')
    exe = tmp_path / "sizes"
    subprocess.run(["gcc", "-std=c99", "-I", os.path.join(T.ROOT, "include"), str(src), "-o", str(exe)], check=True)
    got = [int(x) for x in subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split()]
    want = [C.sizeof(capi.SceneParams), C.sizeof(capi.SceneConfig), C.sizeof(capi.ViewStruct), C.sizeof(capi.Counters),
            C.sizeof(capi.Profile), C.sizeof(capi.RGBDCalib), capi.ViewStruct.M_d.offset, capi.ViewStruct.rgb_to_depth_inv.offset]
    assert got == want


def test_product_does_not_reference_the_oracle(libpath):
    pkg = os.path.join(T.ROOT, "infinitam_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".hpp", ".cpp")) or f == "Makefile":
                text = open(os.path.join(dirpath, f), errors="replace").read()
                assert not re.search(r"oracle/|libitm_oracle|libitm_ref|itmo_|itmr_", text), f"{f} references the oracle"
    strings = subprocess.run(["strings", libpath], capture_output=True, text=True).stdout
    assert "itmo_" not in strings and "libitm_oracle" not in strings


def test_load_fails_loudly_without_the_library(tmp_path, monkeypatch):
    import infinitam_amd
    monkeypatch.setattr(infinitam_amd, "_backend", None)
    monkeypatch.setattr(infinitam_amd, "lib_path", lambda: str(tmp_path / "libitmhip.so"))
    with pytest.raises(capi.ItmError):
        infinitam_amd.load()


def hash_index(pos, mask=0xFFFFF):
    p = np.asarray(pos, np.int64)
    u = p.astype(np.int64) & 0xFFFFFFFF
    h = ((u[..., 0] * 73856093) & 0xFFFFFFFF) ^ ((u[..., 1] * 19349669) & 0xFFFFFFFF) ^ ((u[..., 2] * 83492791) & 0xFFFFFFFF)
    return (h & mask).astype(np.int64)


def test_hash_known_answers():
    """hashIndex on short coordinates (reference DeviceAgnostic/ITMRepresentationAccess.h:8-10), values of Appendix D."""
    cases = {(0, 0, 0): 0, (1, 0, 0): 455773, (0, 1, 0): 475301, (0, 0, 1): 655287, (-1, 0, 0): 592803,
             (-1, -1, -1): 505009, (12, -7, 40): 113593, (-120, 33, 5): 253886, (32767, -32768, 1): 259092}
    for pos, want in cases.items():
        assert int(hash_index(np.array(pos))) == want


@pytest.mark.parametrize("name", ["g_micro_hash_s", "g_vga_hash_s_5mm", "g_vga_hash_s_4mm_bench"])
def test_reference_goldens_obey_hash_layout(name):
    """Properties of the reference's table (in the goldens): ordered entries sit in the bucket their
    position hashes to, block pointers are unique, the visible list is strictly ascending."""
    g, meta = load_golden(name)
    slots, pos, ptr = g["hash_slots"], g["hash_pos"], g["hash_ptr"]
    ordered = slots < 0x100000
    assert np.array_equal(hash_index(pos[ordered]), slots[ordered])
    assert len(np.unique(ptr)) == len(ptr)
    assert np.all(np.diff(g["visible_ids"]) > 0)
    assert meta["counters"][-1]["lastFreeBlockId"] == 0x10000 - 1 - len(ptr)
