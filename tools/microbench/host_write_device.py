#!/usr/bin/env python3
"""Can the host write fine-grained device memory directly (large BAR)?  Probe for the tracker's command path.  Runs the write in a
child process so that a fault is just a non-zero exit code."""
import ctypes as C, subprocess, sys
if len(sys.argv) > 1:
    hip = C.CDLL("libamdhip64.so")
    p = C.c_void_p()
    rc = hip.hipExtMallocWithFlags(C.byref(p), 4096, 0x1)      # hipDeviceMallocFinegrained
    print("hipExtMallocWithFlags rc", rc, hex(p.value or 0), flush=True)
    attr = (C.c_byte * 256)()
    src = (C.c_uint32 * 4)(1, 2, 3, 4)
    C.memmove(p, src, 16)                                       # host store into device memory
    print("host write ok", flush=True)
    back = (C.c_uint32 * 4)()
    rc = hip.hipMemcpy(back, p, 16, 2)
    print("read back through hipMemcpy:", rc, list(back), flush=True)
    dst = (C.c_uint32 * 4)()
    C.memmove(dst, p, 16)                                       # host load from device memory
    print("host read ok:", list(dst), flush=True)
else:
    r = subprocess.run([sys.executable, __file__, "child"], capture_output=True, text=True)
    print(r.stdout, r.stderr[-300:], "exit code", r.returncode)
