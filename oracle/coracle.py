"""ctypes front-end of oracle/oracle.c (TEST INFRASTRUCTURE ONLY; see the header of oracle.c)."""
from __future__ import annotations

import ctypes
import json
import os
import subprocess
from typing import List, Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_DATA = os.path.join(os.path.dirname(_HERE), "splintr_amd", "data")
_LIB = os.path.join(_HERE, "_build", "liboracle.so")

# name -> (vocab container, pattern id, byte_level, special-token table key)
_REG = {
    "cl100k_base": ("cl100k_base.splv", 0, 0, "cl100k_base"),
    "o200k_base": ("o200k_base.splv", 1, 0, "o200k_base"),
    "llama3": ("llama3.splv", 1, 0, "llama3"),
    "deepseek_v3": ("deepseek_v3.splv", 1, 1, "deepseek_v3"),
    "mistral_v3": ("mistral_v3.splv", 2, 1, "mistral_v3"),
}


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "oracle.c")
    if force or not os.path.exists(_LIB) or os.path.getmtime(_LIB) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "-B" if force else "-s"])
    return _LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        L = ctypes.CDLL(build())
        L.orc_create.restype = ctypes.c_void_p
        L.orc_create.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_int, ctypes.c_int]
        L.orc_destroy.argtypes = [ctypes.c_void_p]
        L.orc_set_memo.argtypes = [ctypes.c_void_p, ctypes.c_int]
        L.orc_add_special.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_uint32, ctypes.c_uint32]
        L.orc_encode.restype = ctypes.c_size_t
        L.orc_encode.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_int,
                                 ctypes.POINTER(ctypes.POINTER(ctypes.c_uint32))]
        L.orc_split.restype = ctypes.c_size_t
        L.orc_split.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_size_t,
                                ctypes.POINTER(ctypes.POINTER(ctypes.c_uint32))]
        L.orc_free.argtypes = [ctypes.c_void_p]
        L.orc_encode_batch.restype = ctypes.c_uint64
        L.orc_encode_batch.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64,
                                       ctypes.c_int, ctypes.c_int,
                                       ctypes.POINTER(ctypes.POINTER(ctypes.c_uint32)), ctypes.c_void_p]
        L.orc_pool_pin.argtypes = [ctypes.c_int]
        L.orc_pool_pin.restype = None
        L.orc_pool_set_cpus.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_int)]
        L.orc_pool_set_cpus.restype = None
        L.orc_class_of.restype = ctypes.c_int
        L.orc_class_of.argtypes = [ctypes.c_void_p, ctypes.c_uint32]
        _lib = L
    return _lib


def pool_pin(on: bool = True) -> None:
    """Pin the batch pool's threads (and the calling thread) to one CPU each: bench.py's cpu_baseline legs."""
    lib().orc_pool_pin(1 if on else 0)


def pool_set_cpus(cpus: Sequence[int]) -> None:
    """The CPUs the pinned pool uses, in order (empty: whatever the process may run on).  Call before pool_pin(True)."""
    arr = (ctypes.c_int * max(1, len(cpus)))(*cpus)
    lib().orc_pool_set_cpus(len(cpus), arr)


def idle_cpus(sample_s: float = 0.25, busy_max: float = 0.05) -> List[int]:
    """CPUs of this process's affinity mask that are idle right now (/proc/stat over `sample_s` seconds), minus the ones on which a thread
    of this process that was RUNNING during the sample last ran -- a GPU runtime's spinning helper thread, say (/proc/self/task/*/stat:
    utime + stime before and after, field 39 = the CPU).  Threads that merely exist -- a BLAS pool of 256 sleepers -- exclude nothing."""
    import os, time

    def snap():
        out = {}
        with open("/proc/stat") as f:
            for line in f:
                if line.startswith("cpu") and line[3].isdigit():
                    p = line.split()
                    v = [int(x) for x in p[1:9]]
                    out[int(p[0][3:])] = (sum(v), v[3] + v[4])          # total, idle + iowait
        return out

    def threads():
        out = {}
        me = os.getpid()
        try:
            for tid in os.listdir("/proc/self/task"):
                if int(tid) == me:
                    continue
                with open(f"/proc/self/task/{tid}/stat") as f:
                    fields = f.read().rsplit(")", 1)[1].split()
                out[int(tid)] = (int(fields[11]) + int(fields[12]), int(fields[36]))     # utime + stime, last CPU
        except (OSError, ValueError, IndexError):
            pass
        return out
    allowed = sorted(os.sched_getaffinity(0))
    try:
        a, ta = snap(), threads()
        time.sleep(sample_s)
        b, tb = snap(), threads()
    except OSError:
        return allowed
    mine = {cpu for tid, (ticks, cpu) in tb.items() if tid in ta and ticks > ta[tid][0]}
    good = []
    for c in allowed:
        if c not in a or c not in b or c in mine:
            continue
        dt, di = b[c][0] - a[c][0], b[c][1] - a[c][1]
        if dt <= 0 or 1.0 - di / dt <= busy_max:
            good.append(c)
    return good if len(good) >= 8 else allowed


class COracle:
    def __init__(self, name: str, memo: bool = False):
        base = {"llama3.1": "llama3", "llama3.2": "llama3", "llama3.3": "llama3",
                "deepseek-v3": "deepseek_v3"}.get(name, name)
        if base not in _REG:
            raise ValueError(
                f"Unknown pretrained model: {name}. See from_pretrained docstring for supported models.")
        fn, pid, bl, skey = _REG[base]
        self.name = base
        self._h = lib().orc_create(os.path.join(_DATA, fn).encode(),
                                   os.path.join(_DATA, "unicode_classes.bin").encode(), pid, bl)
        if not self._h:
            raise IOError("orc_create failed")
        with open(os.path.join(_DATA, "special_tokens.json"), encoding="utf-8") as f:
            for lit, tid in json.load(f)[skey].items():
                b = lit.encode("utf-8")
                lib().orc_add_special(self._h, b, len(b), tid)
        if memo:
            lib().orc_set_memo(self._h, 1)

    def __del__(self):
        if getattr(self, "_h", None):
            try:
                lib().orc_destroy(self._h)
            except TypeError:          # interpreter shutdown: module globals are gone already
                pass
            self._h = None

    def encode_bytes(self, data: bytes, with_special: bool = False) -> List[int]:
        p = ctypes.POINTER(ctypes.c_uint32)()
        n = lib().orc_encode(self._h, data, len(data), int(with_special), ctypes.byref(p))
        out = p[:n]
        lib().orc_free(p)
        return out

    def encode(self, text: str) -> List[int]:
        return self.encode_bytes(text.encode("utf-8"))

    def encode_with_special(self, text: str) -> List[int]:
        return self.encode_bytes(text.encode("utf-8"), True)

    def split_bytes(self, data: bytes) -> List[int]:
        p = ctypes.POINTER(ctypes.c_uint32)()
        n = lib().orc_split(self._h, data, len(data), ctypes.byref(p))
        out = p[:n]
        lib().orc_free(p)
        return out

    def encode_packed(self, text: np.ndarray, off: np.ndarray, with_special: bool = False,
                      threads: int = 1):
        """CSR in, CSR out: (ids uint32[T], out_off uint64[N+1])."""
        text = np.ascontiguousarray(text, dtype=np.uint8)
        off = np.ascontiguousarray(off, dtype=np.uint64)
        nd = len(off) - 1
        out_off = np.zeros(nd + 1, dtype=np.uint64)
        p = ctypes.POINTER(ctypes.c_uint32)()
        total = lib().orc_encode_batch(self._h, text.ctypes.data, off.ctypes.data, nd, int(with_special),
                                       threads, ctypes.byref(p), out_off.ctypes.data)
        ids = np.ctypeslib.as_array(p, shape=(max(total, 1),))[:total].copy()
        lib().orc_free(p)
        return ids, out_off

    def encode_batch(self, texts: Sequence[str], with_special: bool = False, threads: int = 1):
        bs = [t.encode("utf-8") for t in texts]
        off = np.zeros(len(bs) + 1, dtype=np.uint64)
        np.cumsum([len(b) for b in bs], out=off[1:])
        text = np.frombuffer(b"".join(bs), dtype=np.uint8)
        ids, oo = self.encode_packed(text, off, with_special, threads)
        return [ids[int(oo[i]):int(oo[i + 1])].tolist() for i in range(len(bs))]

    def class_of(self, cp: int) -> int:
        return lib().orc_class_of(self._h, cp)
