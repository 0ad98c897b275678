"""Dev aid (round 6; -DSPL_DEBUG_STAMPS -DSPL_STAMP_ALL -DSPL_STAMP_FUSE build via SPL_LIB_PATH): every workgroup's wall clock at the
boundaries of the tile kernel's END -- merge done, count published, base known, end -- fused (ONE launch) against k_pretok + k_tile_out.
   SPL_LIB_PATH=_aby/lib_fstamp.so python tools/dev/fuse_walls.py [gen] [ndocs]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from splintr_amd import Tokenizer, corpus, _ffi
from splintr_amd.device import DeviceBatch, encode_device, reserve
gen = sys.argv[1] if len(sys.argv) > 1 else "c2"
ndocs = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
TB = 800
L = _ffi.lib()
tok = Tokenizer.from_pretrained(os.environ.get("SPL_WALLS_VOCAB", "cl100k_base"))
texts = getattr(corpus, gen)(ndocs)
batch = DeviceBatch(texts, torch.device("cuda", 0))
reserve(tok, batch.n_bytes, batch.n_docs)
st = (ctypes.c_uint64 * 16)()
nt = min((batch.n_bytes + TB - 1) // TB, 2048)
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    for fuse in (1, 0):
        L.spl_set_option(tok.handle, b"fuse", fuse)
        L.spl_debug_phases(tok.handle, 1, st)
        acc = None
        for rep in range(12):
            encode_device(tok, batch); torch.cuda.synchronize()
            L.spl_debug_phases(tok.handle, 1, st)
            rec = (ctypes.c_uint64 * (4 * 4096))()
            L.spl_debug_blocks(tok.handle, rec, 4096)
            A = np.ctypeslib.as_array(rec).reshape(2048, 8).astype(np.int64)
            R = A[:nt].copy()
            S = R[:, 3:5].copy()                        # fused: poll statistics, not clocks
            R = R - R[:, 0].min()
            if fuse and rep == 11:
                rounds = S[:, 0] & 0xFFFFFFFF; span = (S[:, 0] >> 32) / 100.0; lastr = S[:, 1] / 100.0
                stats = (rounds, span, lastr)
            if rep >= 2: acc = R if acc is None else acc + R
        R = acc / 10.0 / 100.0          # us
        cols = [(0, "start"), (6, "merge done")] + ([(4, "poll begins"), (1, "count published"), (2, "base known"), (5, "barrier passed")] if fuse else []) + [(7, "end")]
        print(f"{gen} x{ndocs} fuse={fuse}: {nt} tiles; wall clock in us since the first workgroup started (mean of 10 launches)")
        print(f"  {'boundary':18s} {'p10':>6s} {'p50':>6s} {'p90':>6s} {'max':>6s}    since the previous boundary p50 / p90 / max")
        prev = None
        for i, nm in cols:
            c = R[:, i]
            line = f"  {nm:18s} {np.percentile(c,10):6.1f} {np.percentile(c,50):6.1f} {np.percentile(c,90):6.1f} {c.max():6.1f}"
            if prev is not None:
                d = R[:, i] - R[:, prev]
                line += f"    {np.percentile(d,50):5.1f} / {np.percentile(d,90):5.1f} / {d.max():5.1f}"
            print(line); prev = i
        if fuse:
            rounds, span, lastr = stats
            if 0: print(f"  polls (last launch): rounds p50 {np.percentile(rounds,50):.0f} p90 {np.percentile(rounds,90):.0f} max {rounds.max()};  the last round p50 {np.percentile(lastr,50):.2f} p90 {np.percentile(lastr,90):.2f} max {lastr.max():.2f} us;  "
                  f"first -> last poll p50 {np.percentile(span,50):.1f} us;  mean round {np.mean(span[rounds>1]/(rounds[rounds>1]-1)):.2f} us")
            w = R[:, 2] - R[:, 1]
            order = np.argsort(-R[:, 1])[:5]
            print("  the five tiles that publish last: tile, merge done, published, base known, end:", " | ".join(f"{int(t)} {R[t,6]:.1f} {R[t,1]:.1f} {R[t,2]:.1f} {R[t,7]:.1f}" for t in order))
            print(f"  last count published at {R[:,1].max():.1f}; last base known {R[:,2].max():.1f}; last end {R[:,7].max():.1f}; waited (published -> base known) p50 {np.percentile(w,50):.1f} p90 {np.percentile(w,90):.1f}")
