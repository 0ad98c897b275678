"""C-ABI surface: the library loads without a GPU, exports every symbol the header declares, and
reports errors through return codes (no compute calls here)."""
import ctypes
import os
import re

import pytest

from conftest import ROOT


@pytest.fixture(scope="module")
def built():
    import __graft_entry__ as entry
    entry.build()
    from splintr_amd import _ffi
    return _ffi


def test_header_symbols_are_exported(built):
    hdr = open(os.path.join(ROOT, "include", "splintr_hip.h")).read()
    declared = set(re.findall(r"\b(spl_[a-z_0-9]+)\s*\(", hdr)) - {"spl_opts"}
    assert declared == set(built.SYMBOLS), declared ^ set(built.SYMBOLS)
    L = built.lib()
    for s in declared:
        assert hasattr(L, s), s


def test_header_cites_reference_interfaces():
    hdr = open(os.path.join(ROOT, "include", "splintr_hip.h")).read()
    for cite in ("src/core/tokenizer.rs:932-942", "src/core/tokenizer.rs:842-874", "src/python/bindings.rs:57-446",
                 "src/core/tokenizer.rs:964-972"):
        assert cite in hdr


def test_errors_without_gpu_or_bad_input(built):
    L = built.lib()
    assert L.spl_device_count() >= 0
    opts = built.SplOpts(0, 0)
    assert not L.spl_create(b"junk", 4, b"junk", 4, ctypes.byref(opts))
    assert b"spl_create" in L.spl_last_error()
    # spl_opts is versioned by its first member: a caller that does not set it is refused, one built against a
    # LONGER struct than this library knows is accepted (the tail is ignored)
    raw = (ctypes.c_uint32 * 4)(0, 0, 0, 0)
    assert not L.spl_create(b"junk", 4, b"junk", 4, ctypes.cast(raw, ctypes.POINTER(built.SplOpts)))
    assert b"struct_size" in L.spl_last_error()
    big = (ctypes.c_uint32 * 16)(64, 0, 0, 0)
    assert not L.spl_create(b"junk", 4, b"junk", 4, ctypes.cast(big, ctypes.POINTER(built.SplOpts)))
    assert b"struct_size" not in L.spl_last_error()
    assert L.spl_kernel_name(0) and L.spl_kernel_name(99) is None
    assert L.spl_vocab_size(None) == 0


def test_product_has_no_cpu_fallback_and_never_touches_the_oracle():
    pkg = os.path.join(ROOT, "splintr_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".h", ".hip", ".cpp")):
                src = open(os.path.join(dirpath, fn), encoding="utf-8").read()
                assert "oracle" not in src.replace("the oracle's split engine", ""), (fn, "product code mentions the oracle")
                assert "/root/reference" not in src, fn
    import splintr_amd
    from splintr_amd import _ffi
    if _ffi.lib().spl_device_count() == 0:
        with pytest.raises(Exception):
            splintr_amd.Tokenizer.from_pretrained("cl100k_base")      # fails loudly without a GPU


def test_python_surface_argument_errors(built):
    from splintr_amd import Tokenizer
    from splintr_amd.tokenizer import _pack
    with pytest.raises(ValueError, match="Unknown pretrained model: gpt9. See from_pretrained docstring"):
        Tokenizer.from_pretrained("gpt9")
    with pytest.raises(TypeError):
        _pack("a bare string")
    with pytest.raises(TypeError):
        _pack([b"bytes"])
    with pytest.raises(UnicodeEncodeError):
        _pack(["\ud800"])
    buf, off = _pack(["ab", "", "é"])
    assert buf == b"ab\xc3\xa9" and off.tolist() == [0, 2, 2, 4]


def test_missing_librccl_is_an_error_code_not_a_crash(built):
    """with librccl hidden the communicator entry points return SPL_EDEVICE with a message (the error
    string used to be built from a second dlerror() call, which returns NULL: a segfault no guard catches), and a
    failed load does not poison a later one.  Own process: the loader state is a process-wide static."""
    import subprocess
    import sys
    code = r"""
import ctypes, os, sys
sys.path.insert(0, %r)
os.environ["SPL_RCCL_LIB"] = "/nonexistent/librccl-hidden.so"
from splintr_amd import _ffi
L = _ffi.lib()
buf = ctypes.create_string_buffer(128)
rc = L.spl_comm_unique_id(buf)
msg = L.spl_last_error().decode()
assert rc == -2 and "librccl not found" in msg, (rc, msg)
assert not L.spl_comm_create(buf, 0, 1, 0)
assert "librccl not found" in L.spl_last_error().decode()
# a second attempt starts from a clean slate: with the override gone the message must not be the stale one
os.environ["SPL_RCCL_LIB"] = ""
rc2 = L.spl_comm_unique_id(buf)
msg2 = L.spl_last_error().decode() if rc2 else ""
assert "librccl-hidden" not in msg2, msg2
print("ok", rc2)
""" % ROOT
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and "ok" in p.stdout, (p.returncode, p.stdout[-500:], p.stderr[-1500:])
