"""ctypes wrapper of tests/hostsim/hostsim.cpp (TEST TOOL; builds with g++, no GPU needed)."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(os.path.dirname(_HERE))
_DATA = os.path.join(_ROOT, "splintr_amd", "data")
_CSRC = os.path.join(_ROOT, "splintr_amd", "csrc")
_LIB = os.path.join(_HERE, "libhostsim.so")
_REG = {"cl100k_base": ("cl100k_base.splv", 0), "o200k_base": ("o200k_base.splv", 1),
        "llama3": ("llama3.splv", 1), "deepseek_v3": ("deepseek_v3.splv", 1), "mistral_v3": ("mistral_v3.splv", 2)}


def build():
    srcs = [os.path.join(_HERE, "hostsim.cpp"), os.path.join(_CSRC, "spl_tables.cpp"), os.path.join(_CSRC, "spl_regex.cpp")]
    deps = srcs + [os.path.join(_CSRC, h) for h in ("spl_regex.h", "spl_common.h", "spl_scan.h", "spl_scan_masks.h", "spl_scan_starts.h", "spl_scan_words.h", "spl_lookup.h", "spl_tables.h")]
    if not os.path.exists(_LIB) or any(os.path.getmtime(d) > os.path.getmtime(_LIB) for d in deps):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-o", _LIB] + srcs)
    return _LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        L = ctypes.CDLL(build())
        L.hs_create.restype = ctypes.c_void_p
        L.hs_create.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_int, ctypes.c_char_p, ctypes.c_int]
        L.hs_destroy.argtypes = [ctypes.c_void_p]
        L.hs_info.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        L.hs_split.restype = ctypes.c_int
        L.hs_split.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int]
        L.hs_split_sync.restype = ctypes.c_int
        L.hs_split_sync.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
        L.hs_split_masks.restype = ctypes.c_int
        L.hs_split_masks.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
        L.hs_split_starts.restype = ctypes.c_int
        L.hs_split_starts.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p,
                                      ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
        L.hs_bucket_stats.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        L.hs_row_head_check.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        L.hs_classify_check.restype = ctypes.c_int
        L.hs_classify_check.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int]
        L.hs_encode.restype = ctypes.c_int
        L.hs_encode.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
        L.hs_regex_compile.restype = ctypes.c_void_p
        L.hs_regex_compile.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int, ctypes.c_char_p, ctypes.c_int]
        L.hs_regex_free.argtypes = [ctypes.c_void_p]
        L.hs_regex_split.restype = ctypes.c_int
        L.hs_regex_split.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int]
        L.hs_regex_split_bits.restype = ctypes.c_int
        L.hs_regex_split_bits.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_void_p]
        _lib = L
    return _lib


class HostSim:
    def __init__(self, name, ucls="unicode_classes.bin"):
        fn, pid = _REG[name]
        err = ctypes.create_string_buffer(256)
        self._h = lib().hs_create(os.path.join(_DATA, fn).encode(), os.path.join(_DATA, ucls).encode(),
                                  pid, err, 256)
        if not self._h:
            raise ValueError(err.value.decode())

    @classmethod
    def from_file(cls, path, pattern_id=0, ucls="unicode_classes.bin"):
        """a vocabulary file in the reference's tiktoken text format (or an .splv container), built-in pattern `pattern_id`"""
        self = cls.__new__(cls)
        err = ctypes.create_string_buffer(256)
        self._h = lib().hs_create(str(path).encode(), os.path.join(_DATA, ucls).encode(), pattern_id, err, 256)
        if not self._h:
            raise ValueError(err.value.decode())
        return self

    def classify_check(self, data: bytes) -> bool:
        return bool(lib().hs_classify_check(self._h, data, len(data)))

    def info(self):
        a = np.zeros(8, dtype=np.uint32)
        lib().hs_info(self._h, a.ctypes.data)
        return dict(zip(["n_keys", "n_pairs", "max_key_len", "max_id", "short_cap", "long_cap", "pair_cap", "cjk_fast"],
                        a.tolist()))

    def bucket_stats(self):
        """(slots, keys) of the tiny and t8 tables, (buckets, overflowed buckets) of the short table, and the short-table key groups that found no salt."""
        a = np.zeros(7, dtype=np.uint32)
        lib().hs_bucket_stats(self._h, a.ctypes.data)
        return {"tiny": (int(a[0]), int(a[1])), "t8": (int(a[2]), int(a[3])), "short": (int(a[4]), int(a[5])), "unsalted_groups": int(a[6])}

    def row_head_check(self):
        """Prefix entries and the four-byte-prefix filter against the key tables: keys checked, violations, filter fill."""
        a = np.zeros(4, dtype=np.uint32)
        lib().hs_row_head_check(self._h, a.ctypes.data)
        return {"keys": int(a[0]), "violations": int(a[1]), "filter_slots": int(a[2]), "filter_nonzero": int(a[3])}

    def split(self, data: bytes, window: int = 0):
        out = np.zeros(len(data) + 1, dtype=np.uint32)
        k = lib().hs_split(self._h, data, len(data), out.ctypes.data, window)
        if k < 0:
            raise RuntimeError(f"hs_split failed: {k}")
        return out[:k].tolist()

    def split_sync(self, data: bytes):
        out = np.zeros(len(data) + 1, dtype=np.uint32)
        st = np.zeros(2, dtype=np.uint32)
        k = lib().hs_split_sync(self._h, data, len(data), out.ctypes.data, st.ctypes.data)
        if k < 0:
            raise RuntimeError(f"hs_split_sync failed: {k}")
        return out[:k].tolist(), int(st[0]), int(st[1])

    def split_starts(self, docs, tb: int = 768, rh: int = 224, max_iter: int = 64):
        """Per-document start lists from the bit-vector start computation (cl100k), and (tiles, fast tiles, no end sync
        point, disqualifying bytes, loops not converged)."""
        data = b"".join(docs)
        off = np.zeros(len(docs) + 1, dtype=np.int32)
        off[1:] = np.cumsum([len(d) for d in docs])
        out = np.zeros(len(data) + 1, dtype=np.uint32)
        st = np.zeros(7, dtype=np.uint32)
        k = lib().hs_split_starts(self._h, data, len(data), off.ctypes.data, len(docs), out.ctypes.data, tb, rh, max_iter,
                                  st.ctypes.data)
        if k < 0:
            raise RuntimeError(f"hs_split_starts failed: {k}")
        starts = out[:k].astype(np.int64)
        res = []
        for d in range(len(docs)):
            a = starts[(starts >= off[d]) & (starts < off[d + 1])] - off[d]
            res.append(a.tolist())
        return res, tuple(int(x) for x in st)

    def split_masks(self, data: bytes, tb: int = 64, rh: int = 32):
        out = np.zeros(len(data) + 1, dtype=np.uint32)
        k = lib().hs_split_masks(self._h, data, len(data), out.ctypes.data, tb, rh)
        if k < 0:
            raise RuntimeError(f"hs_split_masks failed: {k}")
        return out[:k].tolist()

    def encode(self, data: bytes):
        out = np.zeros(len(data) + 1, dtype=np.uint32)
        hits = ctypes.c_uint32(0)
        k = lib().hs_encode(self._h, data, len(data), out.ctypes.data, ctypes.byref(hits))
        if k < 0:
            raise RuntimeError(f"hs_encode failed: {k}")
        return out[:k].tolist()


class HostRegex:
    """The product's host splitter for custom split patterns (splintr_amd/csrc/spl_regex.cpp), driven directly."""

    def __init__(self, pattern: str, sim: "HostSim" = None):
        self._sim = sim or HostSim("cl100k_base")         # (only its code-point class table matters here)
        err = ctypes.create_string_buffer(512)
        pb = pattern.encode("utf-8")
        self._r = lib().hs_regex_compile(self._sim._h, pb, len(pb), err, 512)
        if not self._r:
            raise ValueError(err.value.decode())

    def __del__(self):
        if getattr(self, "_r", None):
            lib().hs_regex_free(self._r)
            self._r = None

    def split(self, data: bytes):
        out = np.zeros(2 * (len(data) + 1), dtype=np.uint32)
        k = lib().hs_regex_split(self._r, data, len(data), out.ctypes.data, len(data) + 1)
        if k < 0:
            raise RuntimeError("step budget exhausted")
        return [(int(out[2 * i]), int(out[2 * i + 1])) for i in range(k)]

    def split_bits(self, data: bytes, base: int = 0):
        nw = (base + len(data)) // 32 + 2
        st, gp = np.zeros(nw, dtype=np.uint32), np.zeros(nw, dtype=np.uint32)
        assert lib().hs_regex_split_bits(self._r, data, len(data), base, st.ctypes.data, gp.ctypes.data) == 0
        return st, gp
