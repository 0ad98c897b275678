"""Streaming decoders and agent-token classes: CPU checks of the buffer logic
against the known answers of the reference's own unit tests (src/core/streaming.rs:420-643), and
-m gpu checks through real tokenizers (the id -> bytes lookups go to the handle's host tables)."""
import pytest

from splintr_amd.streaming import ByteLevelStreamingDecoder, StreamingDecoder, valid_prefix_len


def _toy():
    # src/core/streaming.rs:400-418: one token per byte value, 256 = "Hello", 257 = "世界", special 300
    table = {i: bytes([i]) for i in range(256)}
    table[256] = b"Hello"
    table[257] = "世界".encode()
    table[300] = b"<|think|>"
    return table.get


def test_streaming_known_answers():
    d = StreamingDecoder(_toy())
    assert d.add_token(ord("H")) == "H" and d.add_token(ord("i")) == "i" and not d.has_pending      # :420-428
    assert d.add_token(257) == "世界" and not d.has_pending                                           # :431-438
    assert d.add_token(0xE4) is None and d.has_pending and d.pending_bytes == 1                       # :441-456
    assert d.add_token(0xB8) is None and d.pending_bytes == 2
    assert d.add_token(0x96) == "世" and not d.has_pending
    d.add_token(0xE4)
    d.add_token(0xB8)
    assert "�" in d.flush() and not d.has_pending                                                # :459-471
    d.add_token(0xE4)
    assert d.has_pending
    d.reset()
    assert not d.has_pending                                                                          # :474-483
    assert d.add_token(ord("H")) == "H" and d.add_token(0xE4) is None and d.has_pending               # :486-499
    d.reset()
    assert d.add_tokens([ord("H"), ord("i"), ord("!")]) == "Hi!"                                      # :502-509
    assert d.add_token(99999) is None and not d.has_pending                                           # unknown id: nothing
    assert d.add_tokens([ord("a"), 99999, 0xE4]) == "a" and d.pending_bytes == 1
    assert repr(d) == "StreamingDecoder(pending_bytes=1)"
    assert d.flush() == "�" and d.flush() == ""
    assert repr(ByteLevelStreamingDecoder(_toy())) == "ByteLevelStreamingDecoder(pending_bytes=0)"


def test_valid_prefix_len():
    # src/python/bindings.rs:591-640
    assert valid_prefix_len(b"") == 0 and valid_prefix_len(b"abc") == 3
    assert valid_prefix_len("世".encode()[:2]) == 0                 # nothing complete yet
    assert valid_prefix_len(b"ab" + "世".encode()[:1]) == 2
    assert valid_prefix_len(b"ab" + "\U0001F30D".encode()[:3]) == 2
    assert valid_prefix_len(b"ab\xff") == 2                          # invalid, not incomplete: longest valid prefix
    assert valid_prefix_len(b"\x80abc") == 0


def test_agent_token_classes():
    import splintr_amd as S
    assert S.CL100K_AGENT_TOKENS.SYSTEM == 100277 and S.CL100K_AGENT_TOKENS.THINK == 100282       # generated docs :15-24
    assert S.O200K_AGENT_TOKENS.SYSTEM == 200019 and S.MISTRAL_V3_AGENT_TOKENS.SYSTEM == 131072
    assert S.DEEPSEEK_V3_AGENT_TOKENS.THINK_END - S.DEEPSEEK_V3_AGENT_TOKENS.THINK == 1
    with pytest.raises(AttributeError):
        S.CL100K_AGENT_TOKENS.SYSTEM = 0
    with pytest.raises(TypeError):
        S.CL100K_AGENT_TOKENS()
    for n in ("CL100K", "O200K", "LLAMA3", "DEEPSEEK_V3", "MISTRAL_V1", "MISTRAL_V2", "MISTRAL_V3"):
        assert n + "_AGENT_TOKENS" in S.__all__


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["cl100k_base", "deepseek_v3", "mistral_v3"])
def test_streaming_through_a_real_tokenizer(name):
    import json
    import os
    from splintr_amd import Tokenizer
    import splintr_amd as S
    t = Tokenizer.from_pretrained(name)
    text = "Hello 🌍 World! 你好世界 — don't"
    ids = t.encode(text)
    d = t.byte_level_streaming_decoder() if name != "cl100k_base" else t.streaming_decoder()
    out = "".join(x for x in (d.add_token(i) for i in ids) if x) + d.flush()
    assert out == text
    # token by token never emits a broken character, and pending bytes drain to zero
    d.reset()
    pieces = [d.add_token(i) for i in ids]
    assert all(p is None or p.encode("utf-8").decode("utf-8") == p for p in pieces) and not d.has_pending
    # special ids decode to their literal; every agent-token constant is the id of a special literal
    with open(os.path.join(os.path.dirname(S.__file__), "data", "special_tokens.json"), encoding="utf-8") as f:
        special = json.load(f)[name]
    lit, tid = next(iter(special.items()))
    assert d.add_token(tid) == lit
    cls = {"cl100k_base": S.CL100K_AGENT_TOKENS, "deepseek_v3": S.DEEPSEEK_V3_AGENT_TOKENS, "mistral_v3": S.MISTRAL_V3_AGENT_TOKENS}[name]
    ids_of = set(special.values())
    assert all(v in ids_of for k, v in vars(cls).items() if k.isupper())
    if name != "cl100k_base":
        # the plain decoder of a ByteLevel vocabulary yields the ByteLevel TEXT (bindings.rs:386-405 clones `decoder`)
        plain = t.streaming_decoder()
        assert plain.add_tokens(t.encode(" world")) == "Ġworld"
