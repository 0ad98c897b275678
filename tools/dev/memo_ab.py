"""Dev aid (round 6): the chunk memo (spl_k_memo.h) on / off in ONE process: parity vs the oracle cold and warm, the bench rotations
(us per step), larger launches, the memo's statistics.
   python tools/dev/memo_ab.py [label]       (SPL_LIB_PATH selects an A/B build)"""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from splintr_amd import Tokenizer, corpus, _ffi
from splintr_amd.device import DeviceBatch, encode_device, reserve, result_csr
from oracle.coracle import COracle
label = sys.argv[1] if len(sys.argv) > 1 else os.path.basename(os.environ.get("SPL_LIB_PATH", "default"))
quick = os.environ.get("MEMO_AB_QUICK") == "1"
L = _ffi.lib(); dev = torch.device("cuda", 0)
L.spl_memo_stats.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint64)]
def opt(tok, k, v):
    if L.spl_set_option(tok.handle, k.encode(), int(v)) != 0: raise RuntimeError(_ffi.last_error())
def stats(tok):
    o = (ctypes.c_uint64 * 4)(); L.spl_memo_stats(tok.handle, o); return list(o)
def packed(texts):
    bs = [t.encode() for t in texts]; off = np.zeros(len(bs) + 1, dtype=np.uint64); np.cumsum([len(b) for b in bs], out=off[1:])
    return np.frombuffer(b"".join(bs), dtype=np.uint8), off
def check(tok, orc, b, t):
    encode_device(tok, b); torch.cuda.synchronize()
    ids, off = result_csr(b); tn, _ = packed(t); o_ids, o_off = orc.encode_packed(tn, b.host_offsets, threads=32)
    return np.array_equal(ids, o_ids) and np.array_equal(off, o_off)
def rotation(tok, batches, n=400, reps=5):
    ts = []
    for rep in range(reps):
        for i in range(40): encode_device(tok, batches[i % len(batches)])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n): encode_device(tok, batches[i % len(batches)])
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / n)
    ts.sort()
    return ts[len(ts) // 2]
stream = torch.cuda.Stream(dev)
with torch.cuda.stream(stream):
    cfgs = [("cl100k_base", "c2", 1002, 1000, 8), ("cl100k_base", "c2_wide", 2002, 1000, 8), ("o200k_base", "c3", 3003, 250, 4)]
    if not quick: cfgs += [("cl100k_base", "c2", 5002, 8000, 1), ("cl100k_base", "c2_wide", 6002, 8000, 1), ("o200k_base", "c3", 7003, 2500, 1), ("llama3", "c4", 8004, 100000, 1), ("deepseek_v3", "c5", 9005, 8, 1)]
    for vocab, gen, seed, ndocs, nb_ in cfgs:
        orc = COracle(vocab)
        sets = [getattr(corpus, gen)(ndocs, seed=seed + k) for k in range(nb_)]
        batches = [DeviceBatch(t, dev) for t in sets]
        for memo in (1, 0, 1, 0)[:4 if nb_ > 1 else 2]:
            tok = Tokenizer.from_pretrained(vocab)
            reserve(tok, max(b.n_bytes for b in batches) + (1 << 20), 200000)
            opt(tok, "memo", memo)
            if memo and os.environ.get("MEMO_BITS"): opt(tok, "memo_bits", int(os.environ["MEMO_BITS"]))
            if memo and os.environ.get("MEMO_LOG_CAP"): opt(tok, "memo_log_cap", int(os.environ["MEMO_LOG_CAP"]))
            ok_cold = all(check(tok, orc, b, t) for b, t in zip(batches[:2], sets[:2]))
            # first passes: the memo fills
            t0 = time.perf_counter()
            for i in range(3 * len(batches) + 4): encode_device(tok, batches[i % len(batches)])
            torch.cuda.synchronize(); t_first = (time.perf_counter() - t0) / (3 * len(batches) + 4)
            ok_warm = all(check(tok, orc, b, t) for b, t in zip(batches[:3], sets[:3]))
            med = rotation(tok, batches, n=400 if ndocs <= 1000 else 30, reps=5 if ndocs <= 1000 else 3)
            ok_end = check(tok, orc, batches[-1], sets[-1])
            nb = sum(b.n_bytes for b in batches) / len(batches)
            st = stats(tok)
            if os.environ.get("MEMO_STATS") == "1":
                ph = (ctypes.c_uint64 * 16)()
                L.spl_debug_phases(tok.handle, 1, ph)
                base_ = list(ph)[4:8]
                for b_ in batches: encode_device(tok, b_)
                torch.cuda.synchronize()
                L.spl_debug_phases(tok.handle, 0, ph)
                d_ = [int(x) - int(y) for x, y in zip(list(ph)[4:8], base_)]
                print(f"[{label}] {gen} x{ndocs} memo={memo}: chunks the vocabulary missed, one pass over the rotation: not probed {d_[0]}, not held {d_[1]}, held {d_[2]}, known too long {d_[3]}", flush=True)
            print(f"[{label}] {vocab} {gen} x{ndocs} memo={memo}: {med*1e6:9.2f} us/step {nb/med/1e9:6.2f} GB/s (first passes {t_first*1e6:9.1f} us/step)  fills {st[0]} put in {st[1]} beyond {st[2]}  "
                  f"{'ok' if ok_cold and ok_warm and ok_end else 'MISMATCH cold=%s warm=%s end=%s' % (ok_cold, ok_warm, ok_end)}", flush=True)
            del tok
