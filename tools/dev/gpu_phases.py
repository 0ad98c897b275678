"""Dev aid: phase cycle stamps of one k_pretok workgroup + per-kernel event times on the bench batch."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from splintr_amd import Tokenizer, corpus, _ffi
from splintr_amd.device import DeviceBatch, encode_device, reserve
L = _ffi.lib()
name = sys.argv[1] if len(sys.argv) > 1 else "cl100k_base"
gen = sys.argv[2] if len(sys.argv) > 2 else "c2"
n = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
tok = Tokenizer.from_pretrained(name)
texts = getattr(corpus, gen)(n)
batch = DeviceBatch(texts, torch.device("cuda", 0))
reserve(tok, batch.n_bytes, batch.n_docs)
st = (ctypes.c_uint64 * 16)()
force = int(sys.argv[4]) if len(sys.argv) > 4 else 0      # 0 auto, 1 small tiles, 2 large tiles
L.spl_debug_phases(tok.handle, 1 | (force << 1), st)
for _ in range(5):
    encode_device(tok, batch)
torch.cuda.synchronize()
L.spl_debug_phases(tok.handle, force << 1, st)
names = ["stage text", "barrier", "classify", "sync flags", "chains", "enumerate", "probe", "merge", "flush"]
print("k_pretok phases (shader cycles) of the middle workgroup:")
for i in range(8):
    print(f"  {names[i + 1]:12s} {st[i + 1] - st[i]:8d}")
print("  total        %8d" % (st[8] - st[0]))
print("  merge: medium (wave 0) %d, short (wave 0) %d, wait for other waves %d" % (st[9] - st[6], st[10] - st[9], st[7] - st[10]))
print("  wall clock (10 ns ticks since workgroup 0 started): middle wg start %d end %d, last wg start %d, kernel end %d"
      % (st[11] - st[14], st[12] - st[14], st[13] - st[14], st[15] - st[14]))
import numpy as np
nb_ = 4096
rec = (ctypes.c_uint64 * (4 * nb_))()
L.spl_debug_blocks(tok.handle, rec, nb_)
R = np.ctypeslib.as_array(rec).reshape(nb_, 4).astype(np.int64)
nblk = (batch.n_bytes + 767) // 768 if (force == 1 or (force == 0 and batch.n_bytes <= 8 << 20)) else (batch.n_bytes + 4095) // 4096
R = R[:min(nblk, nb_)]
k0 = int(st[14])
rel = R - k0
pc = lambda col: tuple(np.percentile(rel[:, col], [50, 90, 99, 100]))
print("  per-workgroup wall clock, 10 ns ticks since the kernel's first workgroup started (p50 p90 p99 max):")
print("    start          %6d %6d %6d %6d" % pc(0))
print("    merge done     %6d %6d %6d %6d" % pc(1))
print("    counts done    %6d %6d %6d %6d" % pc(2))
print("    end            %6d %6d %6d %6d" % pc(3))
slow = int(np.argmax(rel[:, 1]))
print("    slowest merge: wg %d at %d; its counts done %d, end %d" % (slow, rel[slow, 1], rel[slow, 2], rel[slow, 3]))
print("    tile record (end - counts done): p50 %d max %d" % tuple(np.percentile(rel[:, 3] - rel[:, 2], [50, 100])))
qc = (ctypes.c_uint32 * 4)()
L.spl_last_queue_counts(tok.handle, qc)
print("queues: q64", qc[0], "long", qc[2], "deferred", qc[3], "bytes", batch.n_bytes)
L.spl_profile_enable(tok.handle, 1); L.spl_profile_reset(tok.handle)
for _ in range(50):
    encode_device(tok, batch)
torch.cuda.synchronize()
ms = (ctypes.c_double * 16)(); cnt = (ctypes.c_uint64 * 16)()
L.spl_profile_read(tok.handle, ms, cnt); L.spl_profile_enable(tok.handle, 0)
for i in range(16):
    nm = L.spl_kernel_name(i)
    if nm and cnt[i]: print(f"  {nm.decode():24s} {ms[i] / cnt[i] * 1e3:9.2f} us")
import time
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(200): encode_device(tok, batch)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 200
print(f"step {dt * 1e6:.1f} us  -> {batch.n_bytes / dt / 1e6:.1f} MB/s")
# per-wavefront merge-phase record of the middle workgroup (stamps builds)
W = np.ctypeslib.as_array(rec).reshape(nb_, 4)[nb_ - 8:nb_ - 4]
print("merge phase per wavefront of the middle workgroup (cycles): medium loop, short loop, #medium, #short pulls; misses m16 m64")
for w in range(4):
    print("   wave %d: %7d %7d   %d %d   (%d, %d)" % (w, W[w][0], W[w][1], W[w][2] & 0xFFFFFFFF, W[w][2] >> 32, W[w][3] & 0xFFFFFFFF, W[w][3] >> 32))
W2 = np.ctypeslib.as_array(rec).reshape(nb_, 4)[nb_ - 16:nb_ - 8].reshape(4, 8).astype(np.int64)
print("first short pull per wavefront (-DSPL_STAMP_MEDIUM: first medium pull, since the medium loop began), cycles: enter, bytes+len_mask loaded, batch 1 done, batch 2 done, merges done, far spans tabulated")
for w in range(4):
    print("   wave %d: %s" % (w, " ".join("%7d" % x for x in W2[w][:6])))
