"""Build the native library in-tree with hipcc for gfx950 (no JIT cache: the .so must travel
with the repository snapshot to the GPU box)."""
from __future__ import annotations

import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(CSRC, "libsps_hip.so")
SOURCES = [os.path.join(CSRC, "sps_hip.hip")]          # one translation unit; the *.inc.h files are its sections
HEADERS = [os.path.join(os.path.dirname(HERE), "include", "sps_hip.h")] + sorted(
    os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".inc.h"))


def hipcc_path() -> str:
    p = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(p):
        raise RuntimeError("hipcc not found: the sps_amd native library cannot be built")
    return p


def have_hipcc() -> bool:
    return bool(shutil.which("hipcc")) or os.path.exists("/opt/rocm/bin/hipcc")


def is_stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(s) > t for s in SOURCES + HEADERS)


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile sps_amd/csrc/*.hip -> libsps_hip.so (gfx950 only).  Returns the library path."""
    if not force and not is_stale():
        return LIB
    import fcntl
    # several ranks of one job may arrive here together (torchrun): one builds, the others wait and find it fresh
    with open(LIB + ".lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and not is_stale():
                return LIB
            tmp = f"{LIB}.{os.getpid()}.tmp"
            cmd = [hipcc_path(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-o", tmp] + SOURCES
            if verbose:
                print(" ".join(cmd))
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                if os.path.exists(tmp):
                    os.remove(tmp)
                raise RuntimeError("hipcc failed:\n" + r.stdout + r.stderr)
            os.replace(tmp, LIB)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
