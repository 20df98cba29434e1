"""Build libopnet_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
from __future__ import annotations

import os
import shutil
import subprocess

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
LIBDIR = os.path.join(PKG, "lib")
LIB = os.path.join(LIBDIR, "libopnet_hip.so")
ENCODE_LIB = os.path.join(LIBDIR, "libopnet_encode.so")      # encode_host.cpp alone (host code only): what dataset workers load
SOURCES = ["opnet_abi.hip", "opdet_abi.hip", "encode_host.cpp", "clipfile_host.cpp"]
ENCODE_SOURCES = ["encode_host.cpp", "clipfile_host.cpp"]      # host code only: the input encoder + the clip-file reader


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
    if not os.path.exists(LIB) or not os.path.exists(ENCODE_LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(d) > t for d in _deps() if os.path.exists(d))


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not is_stale():
        return LIB
    os.makedirs(LIBDIR, exist_ok=True)
    # -amdgpu-kernarg-preload-count: leading scalar kernel arguments arrive in user SGPRs at dispatch instead of through
    # a scalar load (used by opnet_step_pl; harmless for the struct-argument kernels, which have nothing to preload)
    tmp = f"{LIB}.{os.getpid()}.tmp"
    cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC",
           "-Wno-pass-failed", "-mllvm", "-amdgpu-kernarg-preload-count=10", "-o", tmp] + SOURCES
    if verbose:
        print(" ".join(cmd))
    _run_then_publish(cmd, tmp, LIB, cwd=CSRC)
    build_encoder(force=True, verbose=verbose)
    return LIB


def _run_then_publish(cmd, tmp: str, final: str, cwd=None) -> None:
    """compile into a process-private file, then os.replace it over the library: another process (a DataLoader worker, a
    second rank) either sees the old complete file or the new complete file, never a half-written one"""
    try:
        subprocess.run(cmd, cwd=cwd, check=True)
        os.replace(tmp, final)
    finally:
        if os.path.exists(tmp):
            os.remove(tmp)


def build_encoder(force: bool = False, verbose: bool = False) -> str:
    """the input encoder and the clip-file reader as their own small host library (g++ or hipcc's clang, no device code): a DataLoader worker that only
    encodes clips should not load the 5 MB GPU library and start the HIP runtime (measured: 1.1 s before a worker's first sample)"""
    srcs = [os.path.join(CSRC, f) for f in ENCODE_SOURCES]
    if not force and os.path.exists(ENCODE_LIB) and os.path.getmtime(ENCODE_LIB) >= max(os.path.getmtime(f) for f in srcs):
        return ENCODE_LIB
    os.makedirs(LIBDIR, exist_ok=True)
    cxx = shutil.which("g++") or shutil.which("c++")
    tmp = f"{ENCODE_LIB}.{os.getpid()}.tmp"
    cmd = ([cxx, "-O3", "-std=c++17", "-shared", "-fPIC", "-ffp-contract=off", "-o", tmp] + srcs if cxx else
           [_hipcc(), "-x", "c++", "-O3", "-std=c++17", "-shared", "-fPIC", "-ffp-contract=off", "-o", tmp] + srcs)
    if verbose:
        print(" ".join(cmd))
    _run_then_publish(cmd, tmp, ENCODE_LIB)      # several workers may get here at once: each publishes a complete file
    return ENCODE_LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
