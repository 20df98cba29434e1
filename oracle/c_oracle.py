"""ctypes wrapper of oracle/opnet_oracle.c (TEST INFRASTRUCTURE ONLY - see the C file's header).

The shared object is built with gcc into oracle/_build/ (git-ignored).  Because it is compiled with
-march=native it is keyed by a hash of this host's CPU flags, and rebuilt on a host whose flags
differ (the GPU box), so a binary built here is never executed on a CPU that lacks an instruction.
"""
from __future__ import annotations

import ctypes
import hashlib
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "opnet_oracle.c")
BUILD_DIR = os.path.join(HERE, "_build")
_LIB = None


def _cpu_key() -> str:
    flags = ""
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("flags"):
                    flags = line
                    break
    except OSError:
        pass
    return hashlib.sha1(flags.encode()).hexdigest()[:10]


def lib_path() -> str:
    return os.path.join(BUILD_DIR, f"libopnet_oracle_{_cpu_key()}.so")


def build(force: bool = False) -> str:
    path = lib_path()
    if not force and os.path.exists(path) and os.path.getmtime(path) >= os.path.getmtime(SRC):
        return path
    os.makedirs(BUILD_DIR, exist_ok=True)
    cmd = ["gcc", "-O3", "-march=native", "-fopenmp", "-fno-math-errno", "-shared", "-fPIC", "-o", path, SRC, "-lm"]
    subprocess.run(cmd, check=True)
    return path


def _load():
    global _LIB
    if _LIB is None:
        lib = ctypes.CDLL(build())
        fp = ctypes.POINTER(ctypes.c_float)
        lib.opnet_oracle_forward_f32.restype = ctypes.c_int
        lib.opnet_oracle_forward_f32.argtypes = [fp] * 9 + [ctypes.c_int] * 5
        lib.opnet_oracle_max_threads.restype = ctypes.c_int
        _LIB = lib
    return _LIB


def max_threads() -> int:
    return int(_load().opnet_oracle_max_threads())


def usable_cores() -> int:
    """Cores this process may actually use: the affinity mask capped by the cgroup CPU quota
    (a container can see 256 logical CPUs and be throttled to 16 - oversubscribing a spinning
    OpenMP barrier under a quota is catastrophically slow)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as f:
                txt = f.read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                if q > 0:
                    with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f2:
                        n = min(n, max(1, q // int(f2.read())))
            break
        except (OSError, ValueError, IndexError):
            continue
    return max(1, n)


def opnet_forward(boxes: np.ndarray, p, n_threads: int = 0):
    """boxes [B,T,15,6] f32, p: OPNet state_dict arrays -> (y [B,T,4], logits [B,15,T]) fp32.
    n_threads <= 0 means usable_cores()."""
    lib = _load()
    if n_threads <= 0:
        n_threads = usable_cores()
    boxes = np.ascontiguousarray(boxes, dtype=np.float32)
    B, T = boxes.shape[:2]
    names = ("object_to_track_LSTM.weight_ih_l0", "object_to_track_LSTM.weight_hh_l0",
             "object_to_track_prediction.weight", "video_LSTM.weight_ih_l0", "video_LSTM.weight_hh_l0",
             "prediction_layer.weight")
    ws = [np.ascontiguousarray(p[n], dtype=np.float32) for n in names]
    h1, h2 = ws[1].shape[1], ws[4].shape[1]
    y = np.empty((B, T, 4), dtype=np.float32)
    lg = np.empty((B, 15, T), dtype=np.float32)
    ptr = lambda a: a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))
    rc = lib.opnet_oracle_forward_f32(ptr(boxes), *[ptr(w) for w in ws], ptr(y), ptr(lg), B, T, h1, h2, n_threads)
    if rc != 0:
        raise RuntimeError("opnet_oracle_forward_f32 failed")
    return y, lg
