"""Build libopnet_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
from __future__ import annotations

import os
import shutil
import subprocess

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
LIBDIR = os.path.join(PKG, "lib")
LIB = os.path.join(LIBDIR, "libopnet_hip.so")
SOURCES = ["opnet_abi.hip", "opdet_abi.hip", "encode_host.cpp"]


def _deps():
    """every source the two translation units can include: all of csrc/ plus the public header (a hand-kept list went stale
    when opnet_xcd4_kernels.hip was added - an edited kernel file must always make the library stale)"""
    out = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith((".hip", ".h", ".hpp", ".cpp"))]
    out.append(os.path.join(PKG, "..", "include", "opnet_hip.h"))
    return out


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or install ROCm under /opt/rocm)")


def is_stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(d) > t for d in _deps() if os.path.exists(d))


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not is_stale():
        return LIB
    os.makedirs(LIBDIR, exist_ok=True)
    # -amdgpu-kernarg-preload-count: leading scalar kernel arguments arrive in user SGPRs at dispatch instead of through
    # a scalar load (used by opnet_step_pl; harmless for the struct-argument kernels, which have nothing to preload)
    cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC",
           "-Wno-pass-failed", "-mllvm", "-amdgpu-kernarg-preload-count=10", "-o", LIB] + SOURCES
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, cwd=CSRC, check=True)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
