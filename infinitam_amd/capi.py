"""ctypes binding of the C-ABI declared in include/itm_hip.h.

`Backend(lib_path, prefix)` binds any shared library that exports that ABI under `prefix`
(the product library uses ``itm_``).  On top of the raw functions it offers thin Python mirrors of
the reference's engine interfaces -- `SceneReconstructionEngine` (ResetScene /
AllocateSceneFromDepth / IntegrateIntoScene, reference Engine/ITMSceneReconstructionEngine.h:28-52)
and `VisualisationEngine` (FindVisibleBlocks / CreateExpectedDepths / RenderImage / FindSurface /
CreatePointCloud / CreateICPMaps / ForwardRender / CreateRenderState, reference
Engine/ITMVisualisationEngine.h:18-81) -- so that tests read like calls into the reference.

This module contains no numerics: every method forwards to the library.
"""
from __future__ import annotations

import ctypes as C
import os
import re
from dataclasses import dataclass, field
from typing import Optional

import numpy as np

# ---- enums (include/itm_hip.h) ---------------------------------------------------------------
VOXEL_S, VOXEL_F, VOXEL_S_RGB, VOXEL_F_RGB = 0, 1, 2, 3
INDEX_HASH, INDEX_DENSE = 0, 1
RENDER_SHADED_GREYSCALE, RENDER_COLOUR_FROM_VOLUME, RENDER_COLOUR_FROM_NORMAL = 0, 1, 2
(BUF_HASH_ENTRIES, BUF_EXCESS_LIST, BUF_VOXEL_BLOCKS, BUF_ALLOCATION_LIST, BUF_VISIBLE_IDS,
 BUF_VISIBLE_TYPE, BUF_RANGE_IMAGE, BUF_RAYCAST_RESULT, BUF_RAYCAST_IMAGE, BUF_FORWARD_PROJECTION,
 BUF_MISSING_POINTS, BUF_SWAP_STATES) = range(12)
ERR_INVALID, ERR_DEVICE, ERR_UNSUPPORTED = -1, -2, -3

VOXEL_NAMES = {VOXEL_S: "ITMVoxel_s", VOXEL_F: "ITMVoxel_f", VOXEL_S_RGB: "ITMVoxel_s_rgb",
               VOXEL_F_RGB: "ITMVoxel_f_rgb"}

# numpy views of the reference's POD layouts (Utils/ITMLibDefines.h:71-82, :100-199)
HASH_ENTRY_DTYPE = np.dtype({"names": ["pos", "offset", "ptr"],
                             "formats": [("<i2", 3), "<i4", "<i4"],
                             "offsets": [0, 8, 12], "itemsize": 16})
VOXEL_DTYPES = {
    VOXEL_S: np.dtype({"names": ["sdf", "w_depth"], "formats": ["<i2", "u1"],
                       "offsets": [0, 2], "itemsize": 4}),
    VOXEL_F: np.dtype({"names": ["sdf", "w_depth"], "formats": ["<f4", "u1"],
                       "offsets": [0, 4], "itemsize": 8}),
    VOXEL_S_RGB: np.dtype({"names": ["sdf", "w_depth", "clr", "w_color"],
                           "formats": ["<i2", "u1", ("u1", 3), "u1"],
                           "offsets": [0, 2, 3, 6], "itemsize": 8}),
    VOXEL_F_RGB: np.dtype({"names": ["sdf", "w_depth", "clr", "w_color"],
                           "formats": ["<f4", "u1", ("u1", 3), "u1"],
                           "offsets": [0, 4, 5, 8], "itemsize": 12}),
}


class SceneParams(C.Structure):
    """itm_scene_params == ITMSceneParams (Objects/ITMSceneParams.h:14-70)."""
    _fields_ = [("voxelSize", C.c_float), ("mu", C.c_float), ("maxW", C.c_int32),
                ("viewFrustum_min", C.c_float), ("viewFrustum_max", C.c_float),
                ("stopIntegratingAtMaxW", C.c_int32)]


class SceneConfig(C.Structure):
    _fields_ = [("voxelType", C.c_int32), ("indexType", C.c_int32), ("bucketNum", C.c_int32),
                ("excessNum", C.c_int32), ("localBlockNum", C.c_int32),
                ("denseSize", C.c_int32 * 3), ("denseOffset", C.c_int32 * 3),
                ("denseOffsetSet", C.c_int32), ("maxRenderingBlocks", C.c_int32), ("useSwapping", C.c_int32), ("transferBlockNum", C.c_int32)]


class ViewStruct(C.Structure):
    _fields_ = [("depth", C.c_void_p), ("rgb", C.c_void_p), ("w", C.c_int32), ("h", C.c_int32),
                ("w_rgb", C.c_int32), ("h_rgb", C.c_int32), ("M_d", C.c_float * 16),
                ("intr_d", C.c_float * 4), ("intr_rgb", C.c_float * 4),
                ("rgb_to_depth", C.c_float * 16), ("rgb_to_depth_inv", C.c_float * 16)]


class Profile(C.Structure):
    _fields_ = [("calls", C.c_int32 * 8), ("total_ms", C.c_double * 8)]


TIMED_KERNELS = ["request", "alloc_sweep", "visible_list", "integrate", "range", "raycast", "icp_maps", "empty"]


class TrackerConfig(C.Structure):
    """itm_tracker_config; defaults of ITMLibSettings (Utils/ITMLibSettings.cpp:12-13,62-71)."""
    _fields_ = [("noHierarchyLevels", C.c_int32), ("trackingRegime", C.c_int32 * 8), ("noICPRunTillLevel", C.c_int32),
                ("distThresh", C.c_float), ("terminationThreshold", C.c_float)]

    @staticmethod
    def default():
        t = TrackerConfig()
        t.noHierarchyLevels = 5
        t.trackingRegime[:5] = [3, 3, 1, 1, 1]
        t.noICPRunTillLevel = 0
        t.distThresh = 0.1 * 0.1
        t.terminationThreshold = 1e-3
        return t


class RGBDCalib(C.Structure):
    """itm_rgbd_calib: calibration text of the reference (ITMLib/Utils/ITMCalibIO.cpp)."""
    _fields_ = [("size_rgb", C.c_float * 2), ("intr_rgb", C.c_float * 4), ("size_d", C.c_float * 2), ("intr_d", C.c_float * 4),
                ("rgb_to_depth", C.c_float * 16), ("rgb_to_depth_inv", C.c_float * 16), ("disparityType", C.c_int32),
                ("disparityParams", C.c_float * 2)]


class TrackerGH(C.Structure):
    _fields_ = [("f", C.c_float), ("nabla", C.c_float * 6), ("hessian", C.c_float * 36), ("noValidPoints", C.c_int32)]


class Counters(C.Structure):
    _fields_ = [("lastFreeBlockId", C.c_int32), ("lastFreeExcessListId", C.c_int32),
                ("noVisibleEntries", C.c_int32), ("noFwdProjMissingPoints", C.c_int32),
                ("noTotalPoints", C.c_int32), ("noRenderingBlocks", C.c_int32),
                ("noAllocRequests", C.c_int32), ("statusFlags", C.c_int32)]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


def default_params(voxelSize=0.005, mu=0.02, maxW=100, vf_min=0.35, vf_max=3.0,
                   stopIntegratingAtMaxW=False) -> SceneParams:
    """Defaults of ITMLibSettings (Utils/ITMLibSettings.cpp:10)."""
    return SceneParams(voxelSize, mu, maxW, vf_min, vf_max, int(stopIntegratingAtMaxW))


IDENTITY16 = (1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1)


def header_path() -> str:
    return os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "itm_hip.h")


def declared_functions() -> list:
    """Names declared through ITM_FN(...) in include/itm_hip.h (the boundary) and include/itm_debug.h (the test hooks)."""
    text = ""
    for path in (header_path(), os.path.join(os.path.dirname(header_path()), "itm_debug.h")):
        with open(path) as f:
            text += f.read()
    names = re.findall(r"ITM_FN\((\w+)\)\s*\(", text)
    return sorted(set(n for n in names if n != "name"))


class AccelInfo(C.Structure):
    """itm_accel_info (include/itm_hip.h)."""
    _fields_ = [("directory_bytes", C.c_int64), ("slot_directory_bytes", C.c_int64), ("mirror_bytes", C.c_int64),
                ("origin_directory", C.c_int32 * 3), ("origin_mirror", C.c_int32 * 3), ("placed", C.c_int32), ("moves", C.c_int64),
                ("mirror_pages", C.c_int32), ("mirror_pages_mapped", C.c_int32)]


class ItmError(RuntimeError):
    pass


_P = C.c_void_p
_SIGS = {
    "version": (C.c_char_p, []),
    "last_error": (C.c_char_p, []),
    "uses_device_memory": (C.c_int, []),
    "voxel_size_bytes": (C.c_size_t, [C.c_int]),
    "dev_malloc": (C.c_int, [C.POINTER(_P), C.c_size_t]),
    "dev_free": (C.c_int, [_P]),
    "memcpy_h2d": (C.c_int, [_P, _P, C.c_size_t, _P]),
    "memcpy_d2h": (C.c_int, [_P, _P, C.c_size_t, _P]),
    "stream_synchronize": (C.c_int, [_P]),
    "set_device": (C.c_int, [C.c_int]),
    "scene_create": (C.c_int, [C.POINTER(SceneConfig), C.POINTER(SceneParams), C.POINTER(_P)]),
    "scene_destroy": (C.c_int, [_P]),
    "scene_get_config": (C.c_int, [_P, C.POINTER(SceneConfig), C.POINTER(SceneParams)]),
    "reset_scene": (C.c_int, [_P, _P]),
    "render_state_create": (C.c_int, [_P, C.c_int, C.c_int, C.POINTER(_P)]),
    "render_state_destroy": (C.c_int, [_P]),
    "allocate_scene_from_depth": (C.c_int, [_P, C.POINTER(ViewStruct), _P, C.c_int, _P]),
    "integrate_into_scene": (C.c_int, [_P, C.POINTER(ViewStruct), _P, _P]),
    "find_visible_blocks": (C.c_int, [_P, C.POINTER(C.c_float), C.POINTER(C.c_float), _P, _P]),
    "create_expected_depths": (C.c_int, [_P, C.POINTER(C.c_float), C.POINTER(C.c_float), _P, _P]),
    "render_image": (C.c_int, [_P, C.POINTER(C.c_float), C.POINTER(C.c_float), _P, _P, C.c_int, _P]),
    "find_surface": (C.c_int, [_P, C.POINTER(C.c_float), C.POINTER(C.c_float), _P, _P]),
    "create_point_cloud": (C.c_int, [_P, C.POINTER(ViewStruct), _P, C.c_int, _P, _P, _P]),
    "create_icp_maps": (C.c_int, [_P, C.POINTER(ViewStruct), _P, _P, _P, _P]),
    "forward_render": (C.c_int, [_P, C.POINTER(ViewStruct), _P, _P]),
    "process_frame": (C.c_int, [_P, C.POINTER(ViewStruct), _P, _P, _P, _P]),
    "process_frame_ahead": (C.c_int, [_P, C.POINTER(ViewStruct), C.POINTER(ViewStruct), _P, _P, _P, _P]),
    "flush": (C.c_int, [_P, _P, _P]),
    "cancel_ahead": (C.c_int, [_P, _P, _P]),
    "convert_depth_affine": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_float, C.c_float, _P]),
    "convert_disparity": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, _P]),
    "filter_depth": (C.c_int, [_P, _P, C.c_int, C.c_int, _P]),
    "compute_normal_and_weights": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, _P, _P]),
    "update_view": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, _P, C.c_int, C.c_int, _P, _P, _P, _P, _P]),
    "filter_subsample_with_holes": (C.c_int, [_P, C.c_int, C.c_int, _P, _P]),
    "tracker_compute_g_and_h": (C.c_int, [_P, C.c_int, C.c_int, C.POINTER(C.c_float), _P, _P, C.c_int, C.c_int, C.POINTER(C.c_float),
                                          C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_float, C.c_int, C.POINTER(TrackerGH), _P]),
    "track_camera": (C.c_int, [C.POINTER(TrackerConfig), C.POINTER(ViewStruct), _P, _P, C.POINTER(C.c_float), C.POINTER(C.c_float), _P]),
    "get_counters": (C.c_int, [_P, _P, C.POINTER(Counters), _P]),
    "set_counters": (C.c_int, [_P, _P, C.POINTER(Counters), _P]),
    "buffer_bytes": (C.c_size_t, [_P, _P, C.c_int]),
    "download": (C.c_int, [_P, _P, C.c_int, _P, C.c_size_t, _P]),
    "upload": (C.c_int, [_P, _P, C.c_int, _P, C.c_size_t, _P]),
    "buffer_ptr": (_P, [_P, _P, C.c_int]),
    "debug_set": (C.c_int, [C.c_int, C.c_int]),
    "debug_div32767": (C.c_int, [_P, _P, C.c_int, _P]),
    "debug_divide": (C.c_int, [C.c_int, _P, _P, _P, _P, C.c_int, _P]),
    "profile_enable": (C.c_int, [_P, C.c_uint32]),
    "profile_sample": (C.c_int, [_P, C.c_int]),
    "profile_calibrate": (C.c_int, [_P, C.c_int, _P]),
    "profile_read": (C.c_int, [_P, C.POINTER(Profile), C.c_int]),
    "export_visible_record": (C.c_int, [_P, C.POINTER(C.c_float), C.c_int, _P, _P]),
    "mesh_create": (C.c_int, [_P, C.c_uint32, C.POINTER(_P)]),
    "mesh_destroy": (C.c_int, [_P]),
    "mesh_scene": (C.c_int, [_P, _P, _P]),
    "mesh_info": (C.c_int, [_P, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(_P), _P]),
    "mesh_download": (C.c_int, [_P, _P, C.c_uint32, C.POINTER(C.c_uint32), _P]),
    "mesh_write_obj": (C.c_int, [_P, C.c_char_p, _P]),
    "mesh_write_stl": (C.c_int, [_P, C.c_char_p, _P]),
}


# host-side file formats: exported by the product and by the reference shim; the CPU oracle has no need for them
_HOST_IO_SIGS = {
    "read_depth_image": (C.c_int, [C.c_char_p, _P, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "read_rgb_image": (C.c_int, [C.c_char_p, _P, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "write_depth_image": (C.c_int, [C.c_char_p, _P, C.c_int, C.c_int]),
    "write_rgb_image": (C.c_int, [C.c_char_p, _P, C.c_int, C.c_int]),
    "write_float_depth_image": (C.c_int, [C.c_char_p, _P, C.c_int, C.c_int]),
    "read_rgbd_calib": (C.c_int, [C.c_char_p, _P]),
    "scene_save": (C.c_int, [_P, _P, C.c_char_p, _P]),
    "scene_load": (C.c_int, [_P, _P, C.c_char_p, _P]),
    # tracker handles (the oracle / reference shims have no use for them: their trackers are stateless)
    "debug_icp_track": (C.c_int, [C.POINTER(TrackerConfig), C.POINTER(C.c_float), _P, _P, C.POINTER(C.c_float)]),
    "tracker_create": (C.c_int, [C.POINTER(_P)]),
    "tracker_destroy": (C.c_int, [_P]),
    "tracker_g_and_h": (C.c_int, [_P, _P, C.c_int, C.c_int, C.POINTER(C.c_float), _P, _P, C.c_int, C.c_int, C.POINTER(C.c_float),
                                  C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_float, C.c_int, C.POINTER(TrackerGH), _P]),
    "tracker_track_camera": (C.c_int, [_P, C.POINTER(TrackerConfig), C.POINTER(ViewStruct), _P, _P, C.POINTER(C.c_float), C.POINTER(C.c_float), _P]),
    "debug_column_cull_rows": (C.c_int, [C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_int, C.c_int, C.c_float, C.POINTER(C.c_int), C.POINTER(C.c_int),
                                         C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "debug_dense_classify_check": (C.c_int, [C.POINTER(C.c_int32), C.c_int]),
    "scene_accel_info": (C.c_int, [_P, C.POINTER(AccelInfo)]),
    "scene_set_deferred_fusion": (C.c_int, [_P, C.c_int]),
    "swap_integrate_global_into_local": (C.c_int, [_P, _P, _P]),
    "swap_save_to_global_memory": (C.c_int, [_P, _P, _P]),
    "global_cache_get": (C.c_int, [_P, C.c_int, _P, C.POINTER(C.c_int)]),
    "global_cache_flags": (C.c_int, [_P, _P, C.c_size_t]),
    "host_malloc": (C.c_int, [C.POINTER(_P), C.c_size_t]),
    "host_free": (C.c_int, [_P]),
    "host_register": (C.c_int, [_P, C.c_size_t]),
    "host_unregister": (C.c_int, [_P]),
    "depth_stager_create": (C.c_int, [C.c_int, C.c_int, C.c_int, C.POINTER(_P)]),
    "depth_stager_destroy": (C.c_int, [_P]),
    "depth_stager_upload": (C.c_int, [_P, _P]),
    "depth_stager_acquire": (C.c_int, [_P, _P, C.POINTER(_P)]),
    "depth_stager_release": (C.c_int, [_P, _P]),
    "depth_stager_set_conversion": (C.c_int, [_P, C.c_int, C.c_float, C.c_float, C.c_float]),
    "depth_stager_acquire_depth": (C.c_int, [_P, _P, C.POINTER(_P), C.POINTER(_P)]),
    "depth_stager_pending": (C.c_int, [_P, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "stream_create": (C.c_int, [C.POINTER(_P)]),
    "stream_destroy": (C.c_int, [_P]),
    # multi-stream exchange issued from the library (RCCL); the CPU shims exchange through torch.distributed (streams.py)
    "exchange_unique_id": (C.c_int, [C.POINTER(C.c_ubyte)]),
    "exchange_create": (C.c_int, [C.c_int, C.c_int, C.POINTER(C.c_ubyte), C.c_int, C.c_int, C.POINTER(_P)]),
    "exchange_destroy": (C.c_int, [_P]),
    "exchange_step": (C.c_int, [_P, _P, C.POINTER(C.c_float), _P]),
    "exchange_info": (C.c_int, [_P, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(_P)]),
    "exchange_table": (C.c_int, [_P, _P, C.c_size_t]),
    "exchange_self_check": (C.c_int, [_P, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "exchange_acquire": (C.c_int, [_P, _P, C.POINTER(_P), C.POINTER(C.c_longlong)]),
    "exchange_release": (C.c_int, [_P, _P]),
    "debug_checksum": (C.c_int, [_P, C.c_size_t, C.c_int, _P, _P]),
}


class Backend:
    """One loaded implementation of the C-ABI."""

    def __init__(self, lib_path: str, prefix: str = "itm_"):
        if not os.path.exists(lib_path):
            raise ItmError(f"shared library not found: {lib_path}")
        self.path, self.prefix = lib_path, prefix
        self.lib = C.CDLL(lib_path)
        self.fn = {}
        for name, (res, args) in _SIGS.items():
            f = getattr(self.lib, prefix + name)   # AttributeError => missing symbol, fail loudly
            f.restype, f.argtypes = res, args
            self.fn[name] = f
        for name, (res, args) in _HOST_IO_SIGS.items():
            f = getattr(self.lib, prefix + name, None)
            if f is None:
                if prefix == "itm_":
                    raise ItmError(f"product library lacks {prefix}{name}")
                continue
            f.restype, f.argtypes = res, args
            self.fn[name] = f
        self.on_device = bool(self.fn["uses_device_memory"]())

    # ---- host-side file formats (numpy in / out) -----------------------------------------------
    def read_depth_image(self, path: str) -> np.ndarray:
        cap = 4096 * 4096
        buf = np.empty(cap, np.int16); w, h = C.c_int(), C.c_int()
        self.check(self.fn["read_depth_image"](path.encode(), buf.ctypes.data_as(_P), cap, C.byref(w), C.byref(h)), "read_depth_image")
        return buf[: w.value * h.value].reshape(h.value, w.value).copy()

    def read_rgb_image(self, path: str) -> np.ndarray:
        cap = 4096 * 4096
        buf = np.empty(cap * 4, np.uint8); w, h = C.c_int(), C.c_int()
        self.check(self.fn["read_rgb_image"](path.encode(), buf.ctypes.data_as(_P), cap, C.byref(w), C.byref(h)), "read_rgb_image")
        return buf[: w.value * h.value * 4].reshape(h.value, w.value, 4).copy()

    def write_image(self, path: str, img: np.ndarray):
        img = np.ascontiguousarray(img)
        h, w = img.shape[:2]
        kind = {np.dtype(np.int16): "write_depth_image", np.dtype(np.uint8): "write_rgb_image", np.dtype(np.float32): "write_float_depth_image"}[img.dtype]
        self.check(self.fn[kind](path.encode(), img.ctypes.data_as(_P), w, h), kind)

    def read_rgbd_calib(self, path: str) -> "RGBDCalib":
        out = RGBDCalib()
        self.check(self.fn["read_rgbd_calib"](path.encode(), C.byref(out)), "read_rgbd_calib")
        return out

    def check(self, rc: int, what: str = ""):
        if rc != 0:
            msg = self.fn["last_error"]() or b""
            raise ItmError(f"{self.prefix}{what} failed ({rc}): {msg.decode(errors='replace')}")

    def version(self) -> str:
        return self.fn["version"]().decode()

    # ---- memory in the backend's address space ---------------------------------------------
    def malloc(self, nbytes: int) -> int:
        p = _P()
        self.check(self.fn["dev_malloc"](C.byref(p), nbytes), "dev_malloc")
        return p.value

    def free(self, ptr: int):
        self.check(self.fn["dev_free"](_P(ptr)), "dev_free")

    def to_backend(self, arr: np.ndarray, stream=None) -> "DevBuffer":
        arr = np.ascontiguousarray(arr)
        buf = DevBuffer(self, arr.nbytes, arr.dtype, arr.shape)
        self.check(self.fn["memcpy_h2d"](_P(buf.ptr), arr.ctypes.data_as(_P), arr.nbytes, _P(stream)), "memcpy_h2d")
        self.sync(stream)
        return buf

    def sync(self, stream=None):
        self.check(self.fn["stream_synchronize"](_P(stream)), "stream_synchronize")

    def create_scene(self, voxelType=VOXEL_S, indexType=INDEX_HASH, params: Optional[SceneParams] = None,
                     bucketNum=0, excessNum=0, localBlockNum=0, denseSize=(0, 0, 0), denseOffset=None,
                     maxRenderingBlocks=0, useSwapping=False, transferBlockNum=0) -> "Scene":
        cfg = SceneConfig()
        cfg.useSwapping, cfg.transferBlockNum = (1 if useSwapping else 0), transferBlockNum
        cfg.voxelType, cfg.indexType = voxelType, indexType
        cfg.bucketNum, cfg.excessNum, cfg.localBlockNum = bucketNum, excessNum, localBlockNum
        cfg.denseSize[:] = denseSize
        cfg.maxRenderingBlocks = maxRenderingBlocks
        if denseOffset is not None:
            cfg.denseOffset[:] = denseOffset
            cfg.denseOffsetSet = 1
        prm = params if params is not None else default_params()
        h = _P()
        self.check(self.fn["scene_create"](C.byref(cfg), C.byref(prm), C.byref(h)), "scene_create")
        return Scene(self, h.value)


class DevBuffer:
    """Owning handle to memory in the backend's address space (HBM for the product)."""

    def __init__(self, be: Backend, nbytes: int, dtype=np.uint8, shape=None):
        self.be, self.nbytes, self.dtype = be, int(nbytes), np.dtype(dtype)
        self.shape = shape if shape is not None else (self.nbytes // self.dtype.itemsize,)
        self.ptr = be.malloc(max(self.nbytes, 1))

    def numpy(self, stream=None) -> np.ndarray:
        out = np.empty(self.shape, dtype=self.dtype)
        self.be.check(self.be.fn["memcpy_d2h"](out.ctypes.data_as(_P), _P(self.ptr), self.nbytes, _P(stream)), "memcpy_d2h")
        self.be.sync(stream)
        return out

    def close(self):
        if self.ptr:
            self.be.free(self.ptr)
            self.ptr = 0

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


@dataclass
class View:
    """Python mirror of what the engines read from ITMView + ITMTrackingState::pose_d."""
    depth: DevBuffer
    w: int
    h: int
    M_d: np.ndarray = field(default_factory=lambda: np.array(IDENTITY16, np.float32))
    intr_d: tuple = (580.0, 580.0, 320.0, 240.0)
    rgb: Optional[DevBuffer] = None
    w_rgb: int = 0
    h_rgb: int = 0
    intr_rgb: Optional[tuple] = None
    rgb_to_depth: np.ndarray = field(default_factory=lambda: np.array(IDENTITY16, np.float32))
    rgb_to_depth_inv: np.ndarray = field(default_factory=lambda: np.array(IDENTITY16, np.float32))

    def struct(self) -> ViewStruct:
        v = ViewStruct()
        v.depth = self.depth.ptr if isinstance(self.depth, DevBuffer) else int(self.depth)
        v.rgb = (self.rgb.ptr if isinstance(self.rgb, DevBuffer) else (int(self.rgb) if self.rgb else None))
        v.w, v.h = self.w, self.h
        v.w_rgb, v.h_rgb = (self.w_rgb or self.w), (self.h_rgb or self.h)
        v.M_d[:] = [float(x) for x in np.asarray(self.M_d, np.float32).reshape(16)]
        v.intr_d[:] = [float(x) for x in self.intr_d]
        v.intr_rgb[:] = [float(x) for x in (self.intr_rgb or self.intr_d)]
        v.rgb_to_depth[:] = [float(x) for x in np.asarray(self.rgb_to_depth, np.float32).reshape(16)]
        v.rgb_to_depth_inv[:] = [float(x) for x in np.asarray(self.rgb_to_depth_inv, np.float32).reshape(16)]
        return v


def _fptr(a):
    arr = np.ascontiguousarray(np.asarray(a, np.float32).reshape(-1))
    return arr, arr.ctypes.data_as(C.POINTER(C.c_float))


class Scene:
    """ITMScene<TVoxel,TIndex> + the two engines bound to it."""

    def __init__(self, be: Backend, handle: int):
        self.be, self.h = be, handle
        cfg, prm = SceneConfig(), SceneParams()
        be.check(be.fn["scene_get_config"](_P(handle), C.byref(cfg), C.byref(prm)), "scene_get_config")
        self.cfg, self.params = cfg, prm
        self.reco = SceneReconstructionEngine(self)
        self.vis = VisualisationEngine(self)

    @property
    def voxel_dtype(self):
        return VOXEL_DTYPES[self.cfg.voxelType]

    @property
    def is_hash(self):
        return self.cfg.indexType == INDEX_HASH

    def counters(self, rs: Optional["RenderState"] = None, stream=None) -> dict:
        c = Counters()
        self.be.check(self.be.fn["get_counters"](_P(self.h), _P(rs.h if rs else None), C.byref(c), _P(stream)), "get_counters")
        return c.as_dict()

    def save(self, directory: str, rs: Optional["RenderState"] = None, stream=None):
        """Scene checkpoint (itm_scene_save): MemoryBlockPersister-style files in an existing directory."""
        self.be.check(self.be.fn["scene_save"](_P(self.h), _P(rs.h if rs else None), directory.encode(), _P(stream)), "scene_save")

    def load(self, directory: str, rs: Optional["RenderState"] = None, stream=None):
        self.be.check(self.be.fn["scene_load"](_P(self.h), _P(rs.h if rs else None), directory.encode(), _P(stream)), "scene_load")

    def set_counters(self, rs, lastFreeBlockId, lastFreeExcessListId, noVisibleEntries, stream=None):
        c = Counters()
        c.lastFreeBlockId, c.lastFreeExcessListId, c.noVisibleEntries = lastFreeBlockId, lastFreeExcessListId, noVisibleEntries
        self.be.check(self.be.fn["set_counters"](_P(self.h), _P(rs.h if rs else None), C.byref(c), _P(stream)), "set_counters")

    def _dtype_of(self, which):
        return {BUF_HASH_ENTRIES: HASH_ENTRY_DTYPE, BUF_EXCESS_LIST: np.dtype("<i4"),
                BUF_VOXEL_BLOCKS: self.voxel_dtype, BUF_ALLOCATION_LIST: np.dtype("<i4"),
                BUF_VISIBLE_IDS: np.dtype("<i4"), BUF_VISIBLE_TYPE: np.dtype("u1"),
                BUF_RANGE_IMAGE: np.dtype("<f4"), BUF_RAYCAST_RESULT: np.dtype("<f4"),
                BUF_RAYCAST_IMAGE: np.dtype("u1"), BUF_FORWARD_PROJECTION: np.dtype("<f4"),
                BUF_MISSING_POINTS: np.dtype("<i4"), BUF_SWAP_STATES: np.dtype("u1")}[which]

    def profile_enable(self, mask: int):
        self.be.check(self.be.fn["profile_enable"](_P(self.h), mask), "profile_enable")

    def profile_sample(self, every: int):
        self.be.check(self.be.fn["profile_sample"](_P(self.h), every), "profile_sample")

    def profile_calibrate(self, n: int, stream=None):
        self.be.check(self.be.fn["profile_calibrate"](_P(self.h), int(n), _P(stream)), "profile_calibrate")

    def profile_read(self, reset=True) -> dict:
        p = Profile()
        self.be.check(self.be.fn["profile_read"](_P(self.h), C.byref(p), int(reset)), "profile_read")
        return {n: {"calls": int(p.calls[i]), "total_ms": float(p.total_ms[i])} for i, n in enumerate(TIMED_KERNELS)}

    def download(self, which: int, rs: Optional["RenderState"] = None, stream=None) -> np.ndarray:
        rsh = _P(rs.h if rs else None)
        n = self.be.fn["buffer_bytes"](_P(self.h), rsh, which)
        dt = self._dtype_of(which)
        out = np.empty(n // dt.itemsize, dtype=dt)
        self.be.check(self.be.fn["download"](_P(self.h), rsh, which, out.ctypes.data_as(_P), n, _P(stream)), "download")
        if rs is not None and which in (BUF_RANGE_IMAGE, BUF_RAYCAST_RESULT, BUF_FORWARD_PROJECTION, BUF_RAYCAST_IMAGE):
            comps = {BUF_RANGE_IMAGE: 2}.get(which, 4)
            out = out.reshape(rs.height, rs.width, comps)
        return out

    def upload(self, which: int, arr: np.ndarray, rs: Optional["RenderState"] = None, stream=None):
        arr = np.ascontiguousarray(arr)
        self.be.check(self.be.fn["upload"](_P(self.h), _P(rs.h if rs else None), which, arr.ctypes.data_as(_P), arr.nbytes, _P(stream)), "upload")

    def buffer_ptr(self, which, rs=None) -> int:
        return self.be.fn["buffer_ptr"](_P(self.h), _P(rs.h if rs else None), which) or 0

    # ITMSwappingEngine (scenes created with useSwapping)
    def swap_integrate_global_into_local(self, rs, stream=None):
        self.be.check(self.be.fn["swap_integrate_global_into_local"](_P(self.h), _P(rs.h), _P(stream)), "swap_integrate_global_into_local")

    def swap_save_to_global_memory(self, rs, stream=None):
        self.be.check(self.be.fn["swap_save_to_global_memory"](_P(self.h), _P(rs.h), _P(stream)), "swap_save_to_global_memory")

    def global_cache_flags(self) -> np.ndarray:
        n = self.be.fn["buffer_bytes"](_P(self.h), None, BUF_SWAP_STATES)
        out = np.zeros(n, np.uint8)
        self.be.check(self.be.fn["global_cache_flags"](_P(self.h), out.ctypes.data_as(_P), n), "global_cache_flags")
        return out

    def global_cache_block(self, entry: int):
        out = np.zeros(512, self.voxel_dtype)
        has = C.c_int()
        self.be.check(self.be.fn["global_cache_get"](_P(self.h), int(entry), out.ctypes.data_as(_P), C.byref(has)), "global_cache_get")
        return out if has.value else None

    def accel_info(self) -> dict:
        """Sizes, placement and move count of the directory / mirror cubes (product library only)."""
        a = AccelInfo()
        self.be.check(self.be.fn["scene_accel_info"](_P(self.h), C.byref(a)), "scene_accel_info")
        return {"directory_bytes": a.directory_bytes, "slot_directory_bytes": a.slot_directory_bytes, "mirror_bytes": a.mirror_bytes,
                "mirror_pages": a.mirror_pages, "mirror_pages_mapped": a.mirror_pages_mapped, "origin_directory": list(a.origin_directory), "origin_mirror": list(a.origin_mirror), "placed": bool(a.placed), "moves": a.moves}

    def process_frame(self, view: View, rs: "RenderState", points: DevBuffer, normals: DevBuffer, stream=None):
        """ITMDenseMapper::ProcessFrame + ITMTrackingController::Prepare (Engine/ITMMainEngine.cpp:123-126)."""
        vs = view if isinstance(view, ViewStruct) else view.struct()      # callers on a hot loop keep one ViewStruct and update M_d in place
        self.be.check(self.be.fn["process_frame"](_P(self.h), C.byref(vs), _P(rs.h), _P(points.ptr), _P(normals.ptr), _P(stream)), "process_frame")

    def process_frame_ahead(self, view, next_view, rs: "RenderState", points: DevBuffer, normals: DevBuffer, stream=None):
        """itm_process_frame with the next frame's block requests issued beside this frame's ICP maps (next_view may be None)."""
        vs = view if isinstance(view, ViewStruct) else view.struct()
        ns = None if next_view is None else (next_view if isinstance(next_view, ViewStruct) else next_view.struct())
        self.be.check(self.be.fn["process_frame_ahead"](_P(self.h), C.byref(vs), (C.byref(ns) if ns is not None else None), _P(rs.h), _P(points.ptr), _P(normals.ptr),
                                                        _P(stream)), "process_frame_ahead")

    def set_deferred_fusion(self, on=True):
        """itm_scene_set_deferred_fusion: the four per-frame engine calls may be recorded and launched as one fused frame (the host
        accepts the contract in include/itm_hip.h).  A launch-level matter of the product: other implementations of the ABI ignore it."""
        if "scene_set_deferred_fusion" in self.be.fn:
            self.be.check(self.be.fn["scene_set_deferred_fusion"](_P(self.h), int(bool(on))), "scene_set_deferred_fusion")

    def flush(self, rs: "RenderState" = None, stream=None):
        """Launches the engine calls the library has recorded for this scene / render state (itm_flush)."""
        self.be.check(self.be.fn["flush"](_P(self.h), _P(rs.h if rs is not None else None), _P(stream)), "flush")

    def cancel_ahead(self, rs: "RenderState", stream=None):
        """Abandons the block requests itm_process_frame_ahead issued for the next view (itm_cancel_ahead)."""
        self.be.check(self.be.fn["cancel_ahead"](_P(self.h), _P(rs.h), _P(stream)), "cancel_ahead")

    def close(self):
        if self.h:
            self.be.fn["scene_destroy"](_P(self.h))
            self.h = 0

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Mesh:
    """ITMMesh + ITMMeshingEngine::MeshScene (Objects/ITMMesh.h, Engine/ITMMeshingEngine.h)."""

    def __init__(self, scene: "Scene", max_triangles: int = 0):
        self.scene = scene
        p = _P()
        scene.be.check(scene.be.fn["mesh_create"](_P(scene.h), max_triangles, C.byref(p)), "mesh_create")
        self.h = p.value

    def MeshScene(self, stream=None):
        self.scene.be.check(self.scene.be.fn["mesh_scene"](_P(self.scene.h), _P(self.h), _P(stream)), "mesh_scene")

    def info(self, stream=None):
        n, cap = C.c_uint32(), C.c_uint32()
        self.scene.be.check(self.scene.be.fn["mesh_info"](_P(self.h), C.byref(n), C.byref(cap), None, _P(stream)), "mesh_info")
        return n.value, cap.value

    def triangles(self, stream=None) -> np.ndarray:
        """(noTotalTriangles, 3, 3) float32: p0, p1, p2 per triangle."""
        n, _ = self.info(stream)
        out = np.zeros((n, 3, 3), np.float32)
        got = C.c_uint32()
        self.scene.be.check(self.scene.be.fn["mesh_download"](_P(self.h), out.ctypes.data_as(_P), n, C.byref(got), _P(stream)), "mesh_download")
        return out

    def WriteOBJ(self, path: str, stream=None):
        self.scene.be.check(self.scene.be.fn["mesh_write_obj"](_P(self.h), path.encode(), _P(stream)), "mesh_write_obj")

    def WriteSTL(self, path: str, stream=None):
        self.scene.be.check(self.scene.be.fn["mesh_write_stl"](_P(self.h), path.encode(), _P(stream)), "mesh_write_stl")

    def close(self):
        if self.h:
            self.scene.be.fn["mesh_destroy"](_P(self.h))
            self.h = 0

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class RenderState:
    """ITMRenderState / ITMRenderState_VH (Objects/ITMRenderState.h, ITMRenderState_VH.h)."""

    def __init__(self, scene: Scene, w: int, h: int):
        self.scene, self.width, self.height = scene, w, h
        p = _P()
        scene.be.check(scene.be.fn["render_state_create"](_P(scene.h), w, h, C.byref(p)), "render_state_create")
        self.h = p.value

    def close(self):
        if self.h:
            self.scene.be.fn["render_state_destroy"](_P(self.h))
            self.h = 0

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class SceneReconstructionEngine:
    """Mirror of ITMSceneReconstructionEngine<TVoxel,TIndex> (Engine/ITMSceneReconstructionEngine.h:28-52)."""

    def __init__(self, scene: Scene):
        self.s = scene

    def ResetScene(self, stream=None):
        self.s.be.check(self.s.be.fn["reset_scene"](_P(self.s.h), _P(stream)), "reset_scene")

    def AllocateSceneFromDepth(self, view: View, renderState: RenderState, onlyUpdateVisibleList=False, stream=None):
        vs = view.struct()
        self.s.be.check(self.s.be.fn["allocate_scene_from_depth"](_P(self.s.h), C.byref(vs), _P(renderState.h), int(onlyUpdateVisibleList), _P(stream)), "allocate_scene_from_depth")

    def IntegrateIntoScene(self, view: View, renderState: RenderState, stream=None):
        vs = view.struct()
        self.s.be.check(self.s.be.fn["integrate_into_scene"](_P(self.s.h), C.byref(vs), _P(renderState.h), _P(stream)), "integrate_into_scene")


class VisualisationEngine:
    """Mirror of IITMVisualisationEngine (Engine/ITMVisualisationEngine.h:18-81)."""

    def __init__(self, scene: Scene):
        self.s = scene

    def CreateRenderState(self, imgSize) -> RenderState:
        return RenderState(self.s, int(imgSize[0]), int(imgSize[1]))

    def _pose_call(self, name, M, intr, rs, stream, *extra):
        Ma, Mp = _fptr(M)
        Ia, Ip = _fptr(intr)
        self.s.be.check(self.s.be.fn[name](_P(self.s.h), Mp, Ip, _P(rs.h), *extra, _P(stream)), name)

    def FindVisibleBlocks(self, pose_M, intrinsics, renderState, stream=None):
        self._pose_call("find_visible_blocks", pose_M, intrinsics, renderState, stream)

    def CreateExpectedDepths(self, pose_M, intrinsics, renderState, stream=None):
        self._pose_call("create_expected_depths", pose_M, intrinsics, renderState, stream)

    def RenderImage(self, pose_M, intrinsics, renderState, outputImage: Optional[DevBuffer] = None,
                    type=RENDER_SHADED_GREYSCALE, stream=None):
        self._pose_call("render_image", pose_M, intrinsics, renderState, stream,
                        _P(outputImage.ptr if outputImage else None), int(type))

    def FindSurface(self, pose_M, intrinsics, renderState, stream=None):
        self._pose_call("find_surface", pose_M, intrinsics, renderState, stream)

    def CreatePointCloud(self, view: View, renderState, locations: DevBuffer, colours: DevBuffer, skipPoints=False, stream=None):
        vs = view.struct()
        self.s.be.check(self.s.be.fn["create_point_cloud"](_P(self.s.h), C.byref(vs), _P(renderState.h), int(skipPoints), _P(locations.ptr), _P(colours.ptr), _P(stream)), "create_point_cloud")

    def CreateICPMaps(self, view: View, renderState, points: DevBuffer, normals: DevBuffer, stream=None):
        vs = view.struct()
        self.s.be.check(self.s.be.fn["create_icp_maps"](_P(self.s.h), C.byref(vs), _P(renderState.h), _P(points.ptr), _P(normals.ptr), _P(stream)), "create_icp_maps")

    def ForwardRender(self, view: View, renderState, stream=None):
        vs = view.struct()
        self.s.be.check(self.s.be.fn["forward_render"](_P(self.s.h), C.byref(vs), _P(renderState.h), _P(stream)), "forward_render")
