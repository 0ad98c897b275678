"""Streaming decoders: UTF-8 safe token-by-token decoding on the host.

Mirrors the reference's PyStreamingDecoder / PyByteLevelStreamingDecoder
(src/python/bindings.rs:465-834; core: src/core/streaming.rs): token bytes are appended to a
buffer and only complete UTF-8 characters leave it.  This is sequential CPU work by nature (one
token at a time, as an LLM emits them); the id -> bytes lookups go to the handle's host tables
(spl_token_bytes), nothing runs on the GPU.
"""
from __future__ import annotations

from typing import Callable, Iterable, Optional


def _valid_utf8(b: bytes) -> bool:
    try:
        b.decode("utf-8")
        return True
    except UnicodeDecodeError:
        return False


def _could_be_incomplete(b: bytes) -> bool:
    """src/python/bindings.rs:623-640: a lead byte with fewer continuation bytes than it announces."""
    if not b:
        return False
    f = b[0]
    if 0xC0 <= f <= 0xDF:
        return len(b) < 2
    if 0xE0 <= f <= 0xEF:
        return len(b) < 3
    if 0xF0 <= f <= 0xF7:
        return len(b) < 4
    return False


def valid_prefix_len(buf: bytes) -> int:
    """find_valid_utf8_len (src/python/bindings.rs:591-621): how much of the buffer can be emitted."""
    n = len(buf)
    if n == 0:
        return 0
    if _valid_utf8(buf):
        return n
    for inc in range(1, min(3, n) + 1):
        chk = n - inc
        if chk == 0:
            continue
        if _valid_utf8(buf[:chk]) and _could_be_incomplete(buf[chk:]):
            return chk
    for i in range(n - 1, -1, -1):
        if _valid_utf8(buf[:i + 1]):
            return i + 1
    return 0


class StreamingDecoder:
    """`tokenizer.streaming_decoder()`: src/python/bindings.rs:469-562."""

    _name = "StreamingDecoder"

    def __init__(self, lookup: Callable[[int], Optional[bytes]]):
        self._lookup = lookup
        self._buf = bytearray()

    def _extract(self) -> Optional[str]:
        if not self._buf:
            return None
        n = valid_prefix_len(bytes(self._buf))
        if n == 0:
            return None
        out = bytes(self._buf[:n]).decode("utf-8")
        del self._buf[:n]
        return out

    def add_token(self, token_id: int) -> Optional[str]:
        b = self._lookup(int(token_id))
        if b is None:
            return None                     # unknown id: nothing is emitted, the buffer is left alone
        self._buf += b
        return self._extract()

    def add_tokens(self, token_ids: Iterable[int]) -> Optional[str]:
        for t in token_ids:
            b = self._lookup(int(t))
            if b is not None:
                self._buf += b
        return self._extract()

    def flush(self) -> str:
        out = bytes(self._buf).decode("utf-8", "replace")
        self._buf.clear()
        return out

    def reset(self) -> None:
        self._buf.clear()

    @property
    def has_pending(self) -> bool:
        return bool(self._buf)

    @property
    def pending_bytes(self) -> int:
        return len(self._buf)

    def __repr__(self) -> str:
        return f"{self._name}(pending_bytes={len(self._buf)})"


class ByteLevelStreamingDecoder(StreamingDecoder):
    """`tokenizer.byte_level_streaming_decoder()`: src/python/bindings.rs:653-834 -- the same buffer,
    fed with the ByteLevel-DECODED bytes of every token."""

    _name = "ByteLevelStreamingDecoder"
