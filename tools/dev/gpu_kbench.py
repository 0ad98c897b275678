"""Dev aid: kernel-only timing of the tile-owned path for one library build (SPL_LIB_PATH selects it): the bench
rotation of c2 and of c2_wide (8 x 1000 documents), an 8 MB batch of each, per-kernel microseconds, parity vs the
oracle.   SPL_LIB_PATH=_ab/lib_x.so python tools/dev/gpu_kbench.py [label]"""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from splintr_amd import Tokenizer, corpus, _ffi
from splintr_amd.device import DeviceBatch, encode_device, reserve, result_csr
from oracle.coracle import COracle
label = sys.argv[1] if len(sys.argv) > 1 else os.path.basename(os.environ.get("SPL_LIB_PATH", "default"))
L = _ffi.lib(); dev = torch.device("cuda", 0); orc = COracle("cl100k_base")
def packed(texts):
    bs = [t.encode() for t in texts]; off = np.zeros(len(bs) + 1, dtype=np.uint64); np.cumsum([len(b) for b in bs], out=off[1:])
    return np.frombuffer(b"".join(bs), dtype=np.uint8), off
def kernels(tok, batches, n=64):
    L.spl_profile_enable(tok.handle, 1); L.spl_profile_reset(tok.handle)
    for i in range(n): encode_device(tok, batches[i % len(batches)])
    torch.cuda.synchronize()
    ms = (ctypes.c_double * 16)(); cnt = (ctypes.c_uint64 * 16)(); L.spl_profile_read(tok.handle, ms, cnt); L.spl_profile_enable(tok.handle, 0)
    return {L.spl_kernel_name(i).decode().split("|")[-1]: round(ms[i] / cnt[i] * 1e3, 2) for i in range(16) if L.spl_kernel_name(i) and cnt[i]}
out = []
for gen, seed in (("c2", 1002), ("c2_wide", 2002)):
    sets = [getattr(corpus, gen)(1000, seed=seed + k) for k in range(8)]
    batches = [DeviceBatch(t, dev) for t in sets]
    tok = Tokenizer.from_pretrained("cl100k_base")
    reserve(tok, 9 << 20, 8000)
    ok = True
    for b, t in zip(batches[:3], sets[:3]):
        encode_device(tok, b); torch.cuda.synchronize()
        ids, off = result_csr(b); tn, _ = packed(t); o_ids, o_off = orc.encode_packed(tn, b.host_offsets, threads=32)
        ok = ok and np.array_equal(ids, o_ids) and np.array_equal(off, o_off)
    best = 1e9
    for rep in range(3):
        for i in range(40): encode_device(tok, batches[i % 8])
        torch.cuda.synchronize()
        t0 = time.perf_counter(); n = 400
        for i in range(n): encode_device(tok, batches[i % 8])
        torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / n)
    ks = kernels(tok, batches)
    big = DeviceBatch([t for s in sets for t in s], dev)
    encode_device(tok, big); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(20): encode_device(tok, big)
    torch.cuda.synchronize(); dt8 = (time.perf_counter() - t0) / 20
    out.append(f"{gen}: {best*1e6:6.2f} us/step {batches[0].n_bytes/best/1e9:6.2f} GB/s  k_pretok {ks.get('k_pretok')} k_tile_out {ks.get('k_tile_out')}  8MB {big.n_bytes/dt8/1e9:6.2f} GB/s  {'ok' if ok else 'MISMATCH'}")
print(f"[{label}] " + " | ".join(out), flush=True)
