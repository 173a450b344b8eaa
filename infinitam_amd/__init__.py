"""infinitam_amd -- MI355X-native TSDF allocate / integrate / raycast path behind the InfiniTAM
engine interfaces.  The product is the HIP shared library `libitmhip.so`
(C-ABI: include/itm_hip.h); this package only binds it.

There is no CPU fallback: `load()` raises if the library has not been built.
"""
from __future__ import annotations

import os
import subprocess

from . import capi, synth  # noqa: F401
from .capi import (Backend, DevBuffer, ItmError, RenderState, Scene, View,  # noqa: F401
                   default_params)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_NAME = "libitmhip.so"
_backend = None


def lib_path() -> str:
    # ITM_LIB_OVERRIDE: a measurement build of the SAME library (tools/build_variant.sh) for A/B runs of bench.py and the tools
    return os.environ.get("ITM_LIB_OVERRIDE") or os.path.join(_HERE, LIB_NAME)


def build(verbose: bool = False) -> str:
    """Compile every HIP source for gfx950 into infinitam_amd/libitmhip.so (in-tree)."""
    res = subprocess.run(["make", "-C", os.path.join(_HERE, "csrc"), "-j8"], capture_output=True, text=True)
    if verbose or res.returncode != 0:
        print(res.stdout)
        print(res.stderr)
    if res.returncode != 0:
        raise ItmError("building libitmhip.so failed")
    return lib_path()


def load() -> Backend:
    """Bind libitmhip.so.  Fails loudly when the HIP library is missing -- no fallback."""
    global _backend
    if _backend is None:
        path = lib_path()
        if not os.path.exists(path):
            raise ItmError(f"{path} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(hipcc --offload-arch=gfx950); there is no CPU fallback")
        _backend = Backend(path, "itm_")
    return _backend
