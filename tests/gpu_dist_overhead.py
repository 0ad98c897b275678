"""Dev aid: where the per-step time of the N>1 bench path goes (1-rank RCCL group on one GPU)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
from splintr_amd import Tokenizer, corpus
from splintr_amd.device import DeviceBatch, GatherV, encode_device, reserve, result_csr
tok = Tokenizer.from_pretrained("cl100k_base")
batch = DeviceBatch(corpus.c2(1000), dev)
reserve(tok, batch.n_bytes, batch.n_docs)
encode_device(tok, batch); torch.cuda.synchronize()
ids, off = result_csr(batch)
gv = GatherV(tok, dev, max_docs=batch.n_docs, max_tokens=int(off[-1] * 1.02) + 64)
def run(name, fn, n=300):
    for _ in range(30): fn()
    gv.finish(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    t_host = (time.perf_counter() - t0) / n
    gv.finish(); torch.cuda.synchronize()
    t_all = (time.perf_counter() - t0) / n
    print(f"{name:44s} host issue {t_host * 1e6:7.1f} us/step   total {t_all * 1e6:7.1f} us/step")
run("encode only", lambda: encode_device(tok, batch))
run("encode + submit (pack, all_gather, unpack)", lambda: (encode_device(tok, batch), gv.submit(batch)))
import ctypes
from splintr_amd import _ffi
L = _ffi.lib()
def enc_pack():
    encode_device(tok, batch)
    L.spl_gatherv_pack(tok.handle, batch.ids.data_ptr(), batch.out_off.data_ptr(), batch.n_docs, gv.send[0].data_ptr(),
                       gv.cap_words, gv.max_docs, torch.cuda.current_stream(dev).cuda_stream)
run("encode + pack only", enc_pack)
for d in (8, 16):
    g2 = GatherV(tok, dev, max_docs=batch.n_docs, max_tokens=int(off[-1] * 1.02) + 64, depth=d)
    def f():
        encode_device(tok, batch); g2.submit(batch)
    for _ in range(40): f()
    g2.finish(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(320): f()
    g2.finish(); torch.cuda.synchronize()
    print(f"depth {d}: total {(time.perf_counter() - t0) / 320 * 1e6:.1f} us/step")
send, recv = gv.send[0], gv.recv[0]
run("all_gather_into_tensor only (async)", lambda: dist.all_gather_into_tensor(recv, send, async_op=True))
def ag_wait():
    w = dist.all_gather_into_tensor(recv, send, async_op=True); w.wait()
run("all_gather_into_tensor + wait()", ag_wait)
dist.destroy_process_group()
