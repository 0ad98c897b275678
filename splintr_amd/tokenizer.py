"""`Tokenizer`: the reference's Python surface over the HIP backend.

Mirrors the PyO3 class `Tokenizer` of the reference (src/python/bindings.rs:57-446) method for
method for the encode path: same names, argument meaning, return shapes and error behaviour.
Every encode call goes through the C ABI (include/splintr_hip.h) into the gfx950 kernels; there
is no CPU implementation in this package.
"""
from __future__ import annotations

import ctypes
import json
import os
from typing import Dict, List, Optional, Sequence

import numpy as np

from . import _ffi

_HERE = os.path.dirname(os.path.abspath(__file__))
_DATA = os.path.join(_HERE, "data")

# Verbatim pattern strings exported by the reference module (src/lib.rs:42-44,
# src/core/tokenizer.rs:39, :42, :45).  The HIP backend implements exactly these two.
CL100K_BASE_PATTERN = (
    r"(?i:'s|'t|'re|'ve|'m|'ll|'d)|[^\r\n\p{L}\p{N}]?\p{L}+|\p{N}{1,3}| ?[^\s\p{L}\p{N}]+[\r\n]*|\s*[\r\n]+|\s+(?!\S)|\s+"
)
O200K_BASE_PATTERN = (
    r"[^\r\n\p{L}\p{N}]?[\p{Lu}\p{Lt}\p{Lm}\p{Lo}\p{M}]*[\p{Ll}\p{Lm}\p{Lo}\p{M}]+(?i:'s|'t|'re|'ve|'m|'ll|'d)?"
    r"|[^\r\n\p{L}\p{N}]?[\p{Lu}\p{Lt}\p{Lm}\p{Lo}\p{M}]+[\p{Ll}\p{Lm}\p{Lo}\p{M}]*(?i:'s|'t|'re|'ve|'m|'ll|'d)?"
    r"|\p{N}{1,3}| ?[^\s\p{L}\p{N}]+[\r\n]*|\s*[\r\n]+|\s+(?!\S)|\s+"
)
LLAMA3_PATTERN = O200K_BASE_PATTERN
MISTRAL_V3_PATTERN = (
    r"[^\r\n\p{L}\p{N}]?[\p{Lu}\p{Lt}\p{Lm}\p{Lo}\p{M}]*[\p{Ll}\p{Lm}\p{Lo}\p{M}]+"
    r"|[^\r\n\p{L}\p{N}]?[\p{Lu}\p{Lt}\p{Lm}\p{Lo}\p{M}]+[\p{Ll}\p{Lm}\p{Lo}\p{M}]*"
    r"|\p{N}| ?[^\s\p{L}\p{N}]+[\r\n/]*|\s*[\r\n]+|\s+(?!\S)|\s+"
)
# The GPU scanner implements exactly these patterns; any other pattern is SPL_PATTERN_CUSTOM: its split runs on the
# host cores (csrc/spl_regex.cpp) and feeds the same probe / merge kernels.
_PATTERN_ID = {CL100K_BASE_PATTERN: 0, O200K_BASE_PATTERN: 1, MISTRAL_V3_PATTERN: 2}

# name -> (vocab container, pattern, special-token table key)      src/python/bindings.rs:101-160
_PRETRAINED = {
    "cl100k_base": ("cl100k_base.splv", CL100K_BASE_PATTERN, "cl100k_base"),
    "o200k_base": ("o200k_base.splv", O200K_BASE_PATTERN, "o200k_base"),
    "llama3": ("llama3.splv", LLAMA3_PATTERN, "llama3"),
    "llama3.1": ("llama3.splv", LLAMA3_PATTERN, "llama3"),
    "llama3.2": ("llama3.splv", LLAMA3_PATTERN, "llama3"),
    "llama3.3": ("llama3.splv", LLAMA3_PATTERN, "llama3"),
    "deepseek_v3": ("deepseek_v3.splv", LLAMA3_PATTERN, "deepseek_v3"),
    "deepseek-v3": ("deepseek_v3.splv", LLAMA3_PATTERN, "deepseek_v3"),
    "mistral_v3": ("mistral_v3.splv", MISTRAL_V3_PATTERN, "mistral_v3"),
}


def _byte_level_alphabet() -> List[str]:
    """byte -> char of the GPT-2 ByteLevel alphabet (src/core/byte_level.rs:46-74)."""
    direct = set(range(33, 127)) | set(range(161, 173)) | set(range(174, 256))
    out, nxt = [], 256
    for b in range(256):
        if b in direct:
            out.append(chr(b))
        else:
            out.append(chr(nxt))
            nxt += 1
    return out


_BYTE_TO_CHAR = _byte_level_alphabet()


def _unicode_table_path(which: str) -> str:
    if which == "pcre2":
        return os.path.join(_DATA, "unicode_classes.bin")
    if which == "regex":
        return os.path.join(_DATA, "unicode_classes_regex.bin")
    return which


def _read(path: str) -> bytes:
    with open(path, "rb") as f:
        return f.read()


def _pack(texts: Sequence[str]):
    """list[str] -> (uint8 buffer, uint64 offsets), plain Python (DeviceBatch, tests).  TypeError for
    anything that is not a sequence of str (PyO3 refuses a bare str for Vec<String>;
    src/python/bindings.rs:337)."""
    if isinstance(texts, (str, bytes)):
        raise TypeError("Can't extract `str` to `Vec`")
    parts = []
    for t in texts:
        if not isinstance(t, str):
            raise TypeError(f"argument 'texts': '{type(t).__name__}' object cannot be converted to 'PyString'")
        parts.append(t.encode("utf-8"))          # lone surrogates -> UnicodeEncodeError, as in PyO3
    off = np.zeros(len(parts) + 1, dtype=np.uint64)
    if parts:
        np.cumsum([len(p) for p in parts], out=off[1:])
    buf = b"".join(parts)
    return buf, off


class Tokenizer:
    r"""The reference's `splintr.Tokenizer` surface on the MI355X backend: same constructor
    arguments (a tiktoken-format vocabulary file, a pattern, a special-token map), same methods.

    Differences from the reference, all refused loudly rather than approximated:
    * `pattern`: CL100K_BASE_PATTERN, O200K_BASE_PATTERN (= LLAMA3_PATTERN) and MISTRAL_V3_PATTERN are split on the
      GPU by closed-form scanners; any other pattern is compiled by the library's own regex matcher and run at every text position
      on the GPU (csrc/spl_rx_split.h; on the host cores for the rare DOCUMENT the device matcher gives up
      on -- a match longer than ~1 KB -- while the rest of its batch keeps the device split) (literals, classes, \s \d \w,
      every general category and every script as \p{..} / \P{..}, groups, (?i:), (?>), alternation, greedy / lazy / possessive
      quantifiers, look-ahead, ^ $ \A \Z \z \b \B: upstream tiktoken's cl100k_base / o200k_base strings, Qwen2's, GPT-2's and
      \p{Han}-style patterns are accepted as they are) and merged on the GPU; a pattern with anything else in it (binary
      properties, script extensions, look-behind, back-references, \p{Lu} under (?i)) or one that can match the empty string
      raises the reference's "Regex error" ValueError naming the construct;
    * the vocabulary must contain all 256 single bytes and ids below 2**21;
    * every key of up to 8 bytes gets a table slot of its own by hash-and-displace (csrc/spl_tables.cpp): a vocabulary in
      which so many such keys share one two-byte prefix (keys of 1..4 bytes: 16-bit salt) or one four-byte-prefix filter slot
      (keys of 5..8 bytes: 10-bit salt) that no salt separates them even after the table was doubled three times is refused
      ("could not give every key ... a slot of its own").  None of the pretrained or tested vocabularies comes near it
      (largest group: 256 keys); the reference's FxHashMap has no such limit;
    * special-token literals are at most 255 bytes (any set: literals that contain or chain into one
      another are matched as the reference's Aho-Corasick matcher does).
    Extensions: `device=` / `devices=` select the GPU(s), `byte_level=True` is the reference's
    `from_bytes_byte_level`; the vocabulary may also be this repo's packed SPLV container.
    `unicode_tables=`: which Unicode character data \p{L} \p{N} \s ... mean -- "pcre2" (default: probed from PCRE2 10.39,
    Unicode 14.0, the engine the reference's own tests declare equivalent to its default one), "regex" (probed from the
    Python `regex` module, a newer Unicode: 14 186 code points are classed differently, U+180E among them), or the path
    of a table made by tools/gen_unicode_tables.py.  The reference's default engine (regexr, Cargo.toml:41) ships tables
    of a version that cannot be determined here; nothing in the reference pins the difference.
    """

    def __init__(self, vocab_path: str, pattern: str, special_tokens: Optional[Dict[str, int]] = None, *,
                 device: int = 0, byte_level: bool = False, unicode_tables: str = "pcre2"):
        # src/python/bindings.rs:70-83: every failure of from_file surfaces as IOError
        try:
            blob = _read(vocab_path)
            self._init_from_blob(blob, pattern, special_tokens or {}, device, byte_level, unicode_tables)
        except (OSError, ValueError) as e:
            raise IOError(str(e)) from None

    # ------------------------------------------------------------------ construction
    def _init_from_blob(self, blob: bytes, pattern: str, special_tokens: Dict[str, int], device: int,
                        byte_level: bool = False, unicode_tables: str = "pcre2"):
        if not isinstance(pattern, str):
            raise TypeError("argument 'pattern': object cannot be converted to 'PyString'")
        L = _ffi.lib()
        ucls = _read(_unicode_table_path(unicode_tables))
        flags = _ffi.SPL_OPT_BYTE_LEVEL if byte_level else 0
        if pattern in _PATTERN_ID:
            opts = _ffi.SplOpts(_PATTERN_ID[pattern], device, flags)
        else:                                   # src/core/tokenizer.rs:426: any pattern is compiled
            self._pattern_bytes = pattern.encode("utf-8")
            opts = _ffi.SplOpts(_ffi.SPL_PATTERN_CUSTOM, device, flags, self._pattern_bytes)
        self._h = L.spl_create(blob, len(blob), ucls, len(ucls), ctypes.byref(opts))
        if not self._h:
            raise ValueError(_ffi.last_error())
        self._device = device
        self._pattern = pattern
        self._special = dict(special_tokens)
        for lit, tid in self._special.items():
            b = lit.encode("utf-8")
            if L.spl_add_special(self._h, b, len(b), int(tid)) != 0:
                raise ValueError(_ffi.last_error())

    @classmethod
    def _from_blob(cls, blob: bytes, pattern: str, special_tokens: Dict[str, int], device: int = 0,
                   byte_level: bool = False, unicode_tables: str = "pcre2") -> "Tokenizer":
        self = cls.__new__(cls)
        self._init_from_blob(blob, pattern, special_tokens, device, byte_level, unicode_tables)
        return self

    @staticmethod
    def from_pretrained(name: str, device: int = 0, unicode_tables: str = "pcre2") -> "Tokenizer":
        """src/python/bindings.rs:101-166 (in-scope names: cl100k_base, o200k_base, llama3*,
        deepseek_v3 / deepseek-v3)."""
        if not isinstance(name, str):
            raise TypeError("argument 'name': object cannot be converted to 'PyString'")
        ent = _PRETRAINED.get(name)
        if ent is None:
            raise ValueError(
                f"Unknown pretrained model: {name}. See from_pretrained docstring for supported models.")
        fn, pattern, skey = ent
        with open(os.path.join(_DATA, "special_tokens.json"), encoding="utf-8") as f:
            special = json.load(f)[skey]
        return Tokenizer._from_blob(_read(os.path.join(_DATA, fn)), pattern, special, device, False, unicode_tables)

    @staticmethod
    def from_bytes(vocab_data: bytes, pattern: str, special_tokens: Optional[Dict[str, int]] = None,
                   device: int = 0, unicode_tables: str = "pcre2") -> "Tokenizer":
        """src/python/bindings.rs:174-187: `vocab_data` is tiktoken text (`base64 rank` lines), as in
        the reference; this repo's SPLV container is accepted as well."""
        return Tokenizer._from_blob(bytes(vocab_data), pattern, special_tokens or {}, device, False, unicode_tables)

    @staticmethod
    def from_bytes_byte_level(vocab_data: bytes, pattern: str, special_tokens: Optional[Dict[str, int]] = None,
                              device: int = 0, unicode_tables: str = "pcre2") -> "Tokenizer":
        """src/core/tokenizer.rs:562-569 (not exposed by the reference's Python class; used by its
        from_pretrained for deepseek_v3 and mistral_v3)."""
        return Tokenizer._from_blob(bytes(vocab_data), pattern, special_tokens or {}, device, True, unicode_tables)

    def set_devices(self, devices: Sequence[int]) -> "Tokenizer":
        """Extension: spread `encode_batch` over several GPUs from this one process
        (spl_set_devices: documents sharded by bytes, one pinned CSR result)."""
        arr = (ctypes.c_int32 * len(devices))(*[int(d) for d in devices])
        if _ffi.lib().spl_set_devices(self._h, arr, len(devices)) != 0:
            raise ValueError(_ffi.last_error())
        return self

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            try:
                _ffi.lib().spl_destroy(h)
            except Exception:
                pass
            self._h = None

    # ------------------------------------------------------------------ encode path
    def _encode_packed(self, buf, off, nd: int, flags: int):
        """C-ABI call: packed UTF-8 + offsets in (objects or addresses), CSR (ids uint32, offsets
        uint64) out as numpy arrays."""
        L = _ffi.lib()
        res = ctypes.c_void_p()
        rc = L.spl_encode_batch(self._h, buf, off, nd, flags, ctypes.byref(res))
        if rc != 0:
            raise RuntimeError(f"spl_encode_batch failed ({rc}): {_ffi.last_error()}")
        try:
            nt = L.spl_result_n_tokens(res)
            ids = (np.ctypeslib.as_array(L.spl_result_tokens(res), shape=(nt,)).copy() if nt
                   else np.zeros(0, dtype=np.uint32))
            oo = np.ctypeslib.as_array(L.spl_result_offsets(res), shape=(nd + 1,)).copy()
        finally:
            L.spl_result_free(res)
        return ids, oo

    def encode_packed(self, utf8: bytes, offsets: np.ndarray, with_special: bool = False):
        """Extension: the C-ABI call on an already packed corpus (bytes + uint64 offsets[N+1])."""
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        return self._encode_packed(utf8, offsets.ctypes.data, len(offsets) - 1,
                                   _ffi.SPL_WITH_SPECIAL if with_special else 0)

    def encode_batch_csr(self, texts: Sequence[str], with_special: bool = False):
        """Extension: the batch result as CSR numpy arrays (ids uint32[T], offsets uint64[N+1])
        without materialising Python lists."""
        # ONE C call with the GIL held (staging, spl_encode_batch, copy out): safe from several threads
        ids, off = _ffi.shim().encode_batch_csr(self._h, texts, _ffi.SPL_WITH_SPECIAL if with_special else 0)
        return np.frombuffer(ids, dtype=np.uint32), np.frombuffer(off, dtype=np.uint64)

    def _encode_one(self, text: str, flags: int) -> List[int]:
        return _ffi.shim().encode(self._h, text, flags)

    def encode(self, text: str) -> List[int]:
        """src/python/bindings.rs:254-256.  Texts of up to 4 KB take the library's latency path (csrc/spl_api.hip encode_small: no copy
        engine, no stream synchronisation -- the tile kernel reads the text from pinned host memory, the host spins on a completion word)."""
        return _ffi.shim().encode(self._h, text, 0)

    def encode_rayon(self, text: str) -> List[int]:
        """src/python/bindings.rs:273-275: same ids as encode (the GPU path is always
        chunk-parallel inside a text)."""
        return self._encode_one(text, _ffi.SPL_INTRA_DOC)

    def encode_with_special(self, text: str) -> List[int]:
        """src/python/bindings.rs:286-288."""
        return self._encode_one(text, _ffi.SPL_WITH_SPECIAL)

    def _batch(self, texts: Sequence[str], flags: int) -> List[List[int]]:
        # ONE C call: UTF-8 straight into pinned staging, spl_encode_batch, lists from the pinned CSR
        return _ffi.shim().encode_batch(self._h, texts, flags)

    def encode_batch(self, texts: Sequence[str]) -> List[List[int]]:
        """src/python/bindings.rs:337-339."""
        return self._batch(texts, 0)

    def encode_batch_with_special(self, texts: Sequence[str]) -> List[List[int]]:
        """src/python/bindings.rs:348-350."""
        return self._batch(texts, _ffi.SPL_WITH_SPECIAL)

    # ------------------------------------------------------------------ decode (test helper / "next" row)
    def decode_bytes(self, tokens: Sequence[int]) -> bytes:
        """src/python/bindings.rs:313-315."""
        return self._decode_batch_bytes([tokens])[0]

    def _decode_batch_bytes(self, token_lists: Sequence[Sequence[int]]) -> List[bytes]:
        L = _ffi.lib()
        off = np.zeros(len(token_lists) + 1, dtype=np.uint64)
        if len(token_lists):
            np.cumsum([len(t) for t in token_lists], out=off[1:])
        ids = np.fromiter((x for t in token_lists for x in t), dtype=np.uint32, count=int(off[-1]))
        ob = ctypes.POINTER(ctypes.c_uint8)()
        oo = ctypes.POINTER(ctypes.c_uint64)()
        rc = L.spl_decode_batch(self._h, ids.ctypes.data, off.ctypes.data, len(token_lists), ctypes.byref(ob),
                                ctypes.byref(oo))
        if rc != 0:
            raise RuntimeError(f"spl_decode_batch failed ({rc}): {_ffi.last_error()}")
        try:
            o = [oo[i] for i in range(len(token_lists) + 1)]
            raw = ctypes.string_at(ob, o[-1]) if o[-1] else b""
        finally:
            L.spl_free(ob)
            L.spl_free(oo)
        return [raw[o[i]:o[i + 1]] for i in range(len(token_lists))]

    def decode(self, tokens: Sequence[int]) -> str:
        """src/python/bindings.rs:300-304 (ValueError on invalid UTF-8)."""
        try:
            return self.decode_bytes(tokens).decode("utf-8")
        except UnicodeDecodeError:
            raise ValueError("Decoding error: invalid UTF-8") from None

    def decode_lossy(self, tokens: Sequence[int]) -> str:
        return self.decode_bytes(tokens).decode("utf-8", "replace")

    def decode_batch(self, token_lists: Sequence[Sequence[int]]) -> List[str]:
        try:
            return [b.decode("utf-8") for b in self._decode_batch_bytes(token_lists)]
        except UnicodeDecodeError:
            raise ValueError("Decoding error: invalid UTF-8") from None

    def decode_batch_lossy(self, token_lists: Sequence[Sequence[int]]) -> List[str]:
        return [b.decode("utf-8", "replace") for b in self._decode_batch_bytes(token_lists)]

    # ------------------------------------------------------------------ streaming decoders (host side)
    def _token_bytes(self, token_id: int, byte_level_decoded: bool) -> Optional[bytes]:
        """What the reference's decoder / special_tokens_decoder maps give for one id.
        byte_level_decoded False: the vocabulary KEY (ByteLevel text for ByteLevel vocabularies)."""
        if not 0 <= token_id < (1 << 32):
            return None
        L = _ffi.lib()
        p, n = ctypes.c_void_p(), ctypes.c_uint32()
        kind = L.spl_token_bytes(self._h, token_id, ctypes.byref(p), ctypes.byref(n))
        if kind == 0:
            return None
        b = ctypes.string_at(p, n.value) if n.value else b""
        if kind == 1 and not byte_level_decoded and L.spl_is_byte_level(self._h):
            b = "".join(_BYTE_TO_CHAR[x] for x in b).encode("utf-8")      # back to the key (byte_level.rs:105-107)
        return b

    def streaming_decoder(self):
        """src/python/bindings.rs:386-405."""
        from .streaming import StreamingDecoder
        return StreamingDecoder(lambda t: self._token_bytes(t, False))

    def byte_level_streaming_decoder(self):
        """src/python/bindings.rs:407-427."""
        from .streaming import ByteLevelStreamingDecoder
        return ByteLevelStreamingDecoder(lambda t: self._token_bytes(t, True))

    # ------------------------------------------------------------------ cheap surface
    @property
    def vocab_size(self) -> int:
        """src/python/bindings.rs:382-385 -> src/core/tokenizer.rs:964-972."""
        return int(_ffi.lib().spl_vocab_size(self._h))

    @property
    def cache_len(self) -> int:
        """Tokenizer::cache_len (src/core/tokenizer.rs:1002-1005; bindings.rs:438-440).  The reference's LRU of encoded chunks is, on the GPU
        path, the chunk memo (csrc/spl_k_memo.h): result-transparent there and here.  The memo is filled BETWEEN launches, once the tiles have
        logged enough chunks it did not hold -- so unlike the reference's cache it may still be empty after one short text."""
        import ctypes
        out = (ctypes.c_uint64 * 4)()
        L = _ffi.lib()
        L.spl_memo_stats.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint64)]
        if L.spl_memo_stats(self.handle, out) != 0:
            raise RuntimeError(_ffi.last_error())
        return int(out[1] + out[2])

    def clear_cache(self) -> None:
        """Tokenizer::clear_cache (src/core/tokenizer.rs:995-1000): the memo starts empty again."""
        if _ffi.lib().spl_set_option(self.handle, b"memo_clear", 1) != 0:
            raise RuntimeError(_ffi.last_error())

    @property
    def has_custom_pattern(self) -> bool:
        """Extension: True when the split pattern is not one of the three the GPU scanner implements (its split then
        runs as a matcher program -- on the GPU, or on the host cores for what the device matcher gives up on --, and no context-free cut position is
        known for it: splintr_amd.distributed keeps such a tokenizer's documents whole)."""
        return self._pattern not in _PATTERN_ID

    def pcre2(self, use_pcre2: bool = True) -> "Tokenizer":
        """Backend switches (src/python/bindings.rs:207-242) select between regex engines that
        the reference's tests require to agree; here both map to the one scanner."""
        return self

    def jit(self, use_jit: bool = True) -> "Tokenizer":
        return self

    def __repr__(self) -> str:
        return f"Tokenizer(vocab_size={self.vocab_size})"

    # ------------------------------------------------------------------ backend hooks (bench / tests)
    @property
    def handle(self) -> int:
        return self._h
