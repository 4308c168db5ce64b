#!/usr/bin/env python3
"""Probe: does an unprivileged process get a stable power state from the amdgpu driver (DRM_IOCTL_AMDGPU_CTX,
AMDGPU_CTX_OP_SET_STABLE_PSTATE = 6, flags 4 = PEAK, 1 = STANDARD), and what does it do to the latency-bound calls?
    python tools/ubench/stable_pstate_probe.py [none|standard|peak] [n ...]
Runs tools/trace_verify_resident.py's measurement in this process after the request.  Measurement only: the product
never issues this ioctl."""
import ctypes as C
import fcntl
import glob
import os
import struct
import sys

IOCTL_AMDGPU_CTX = 0xC0106442   # DRM_IOWR(DRM_COMMAND_BASE + DRM_AMDGPU_CTX, union drm_amdgpu_ctx /* 16 bytes */)
OP_ALLOC, OP_FREE, OP_GET_PSTATE, OP_SET_PSTATE = 1, 2, 5, 6
FLAGS = {"none": 0, "standard": 1, "min_sclk": 2, "min_mclk": 3, "peak": 4}


def ctx_ioctl(fd, op, flags=0, ctx_id=0):
    buf = bytearray(struct.pack("IIIi", op, flags, ctx_id, 0))
    fcntl.ioctl(fd, IOCTL_AMDGPU_CTX, buf, True)
    return struct.unpack("II", bytes(buf[:8]))


def main():
    want = sys.argv[1] if len(sys.argv) > 1 else "peak"
    keep = []
    if want != "none":
        for node in sorted(glob.glob("/dev/dri/renderD*")):
            try:
                fd = os.open(node, os.O_RDWR)
                ctx_id, _ = ctx_ioctl(fd, OP_ALLOC)
                before, _ = ctx_ioctl(fd, OP_GET_PSTATE, 0, ctx_id)
                ctx_ioctl(fd, OP_SET_PSTATE, FLAGS[want], ctx_id)
                after, _ = ctx_ioctl(fd, OP_GET_PSTATE, 0, ctx_id)
                print("%s: ctx %d stable pstate %d -> %d" % (node, ctx_id, before, after), flush=True)
                keep.append((fd, ctx_id))
            except OSError as e:
                print("%s: %s" % (node, e), flush=True)
    sys.argv = [sys.argv[0]] + (sys.argv[2:] or ["512", "4096"])
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.dirname(here))
    import trace_verify_resident
    trace_verify_resident.main()


if __name__ == "__main__":
    main()
