"""Dev aid (round 6): where the C3 Python surface (1600 -> 1001 MB/s) and the custom-pattern host path (5.50 -> 4.99 GB/s)
lost their time.  One process; the round-5 options toggled one at a time; the Python surface in its three parts (pack, the C call,
list building); every figure the median of 7 calls.
   python tools/dev/surface_bisect.py [c3|custom|both]"""
import ctypes, os, statistics, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
from splintr_amd import Tokenizer, corpus, _ffi
L = _ffi.lib(); shim = _ffi.shim()
what = sys.argv[1] if len(sys.argv) > 1 else "both"

def med(fn, n=7):
    fn()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); r = fn(); ts.append(time.perf_counter() - t0); del r
    return statistics.median(ts)

def opt(tok, k, v):
    assert L.spl_set_option(tok.handle, k.encode(), int(v)) == 0, _ffi.last_error()

if what in ("c3", "both"):
    texts = corpus.c3(10000)
    nb = sum(len(t.encode()) for t in texts)
    base = dict(twin_streams=1, copy_threads=4, pick_streams=1, chunk_bytes=5 << 20)
    variants = [("default", {}), ("twin_streams=0", dict(twin_streams=0)), ("copy_threads=1", dict(copy_threads=1)),
                ("pick_streams=0", dict(pick_streams=0)), ("chunk_bytes=8M", dict(chunk_bytes=8 << 20)),
                ("r04-like: twin 0, copy 1, pick 0, 8M", dict(twin_streams=0, copy_threads=1, pick_streams=0, chunk_bytes=8 << 20)), ("default again", {})]
    for name, o in variants:
        tok = Tokenizer.from_pretrained("o200k_base")             # (a fresh handle: pick_streams acts at the pipeline's first use)
        for k, v in {**base, **o}.items(): opt(tok, k, v)
        t_all = med(lambda: tok.encode_batch(texts))
        t_csr = med(lambda: shim.encode_batch_csr(tok.handle, texts, 0))
        ids_b, off_b = shim.encode_batch_csr(tok.handle, texts, 0)
        t_lists = med(lambda: shim.lists_from_csr(ids_b, off_b))
        t_pack = med(lambda: shim.pack_bytes(texts))
        blob, offs = shim.pack_bytes(texts)
        p = L.spl_host_alloc(nb + 64); ctypes.memmove(p, blob, nb)
        off_np = np.frombuffer(offs, dtype=np.uint64)
        def c_call():
            r = ctypes.c_void_p()
            assert L.spl_encode_batch(tok.handle, p, off_np.ctypes.data, len(texts), 0, ctypes.byref(r)) == 0
            L.spl_result_free(r)
        t_c = med(c_call)
        L.spl_host_free(p)
        print(f"[c3 {name:40s}] encode_batch {nb/t_all/1e6:7.0f} MB/s ({t_all*1e3:6.2f} ms) | csr {t_csr*1e3:6.2f} ms | lists_from_csr {t_lists*1e3:6.2f} ms | pack_bytes {t_pack*1e3:6.2f} ms | C call pinned {t_c*1e3:6.2f} ms ({nb/t_c/1e9:5.1f} GB/s)", flush=True)
        del tok

if what in ("custom", "both"):
    GPT2 = r"""'s|'t|'re|'ve|'m|'ll|'d| ?\p{L}+| ?\p{N}+| ?[^\s\p{L}\p{N}]+|\s+(?!\S)|\s+"""
    texts = corpus.c2(1000)
    bs = [t.encode() for t in texts]; nb = sum(map(len, bs))
    off = np.zeros(len(bs) + 1, dtype=np.uint64); np.cumsum([len(b) for b in bs], out=off[1:])
    blob = b"".join(bs)
    for name, o in [("default", {}), ("direct_read=0", dict(direct_read=0)), ("direct_write=0", dict(direct_write=0)), ("fuse=0", dict(fuse=0)), ("device_split=0", dict(device_split=0)), ("default again", {})]:
        tok = Tokenizer(os.path.join(ROOT, "splintr_amd", "data", "cl100k_base.splv"), GPT2)
        for k, v in o.items(): opt(tok, k, v)
        p = L.spl_host_alloc(nb + 64); ctypes.memmove(p, blob, nb)
        def c_call():
            r = ctypes.c_void_p()
            assert L.spl_encode_batch(tok.handle, p, off.ctypes.data, len(bs), 0, ctypes.byref(r)) == 0, _ffi.last_error()
            L.spl_result_free(r)
        t_c = med(c_call, n=41)
        t_py = med(lambda: tok.encode_batch(texts), n=21)
        L.spl_host_free(p)
        print(f"[custom {name:16s}] c_abi_host {nb/t_c/1e6:7.0f} MB/s ({t_c*1e6:6.1f} us) | python_surface {nb/t_py/1e6:7.0f} MB/s", flush=True)
        del tok

if what in ("c2", "both"):
    texts = corpus.c2(1000)
    bs = [t.encode() for t in texts]; nb = sum(map(len, bs))
    off = np.zeros(len(bs) + 1, dtype=np.uint64); np.cumsum([len(b) for b in bs], out=off[1:])
    blob = b"".join(bs)
    tok = Tokenizer.from_pretrained("cl100k_base")
    p = L.spl_host_alloc(nb + 64); ctypes.memmove(p, blob, nb)
    def c_call():
        r = ctypes.c_void_p()
        assert L.spl_encode_batch(tok.handle, p, off.ctypes.data, len(bs), 0, ctypes.byref(r)) == 0, _ffi.last_error()
        L.spl_result_free(r)
    for name, o in [("default", {}), ("fuse=0", dict(fuse=0)), ("direct_read=0", dict(direct_read=0)), ("fuse=0 direct_read=0", dict(fuse=0, direct_read=0)), ("default", {}), ("fuse=0", dict(fuse=0))]:
        for k, v in dict(fuse=1, direct_read=1).items(): opt(tok, k, v)
        for k, v in o.items(): opt(tok, k, v)
        t_c = med(c_call, n=201)
        t_py = med(lambda: tok.encode_batch(texts), n=21)
        print(f"[c2 {name:22s}] c_abi_host {nb/t_c/1e6:7.0f} MB/s ({t_c*1e6:6.1f} us) | python_surface {nb/t_py/1e6:7.0f} MB/s", flush=True)
    L.spl_host_free(p)
