"""Dev aid: per-step time of the N>1 bench path (1-rank RCCL group on one GPU) for several bucket depths."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
def run(name, fn, fin, n=320):
    for _ in range(40): fn()
    fin(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    t_host = (time.perf_counter() - t0) / n
    fin(); torch.cuda.synchronize()
    t_all = (time.perf_counter() - t0) / n
    print(f"{name:34s} host issue {t_host * 1e6:7.1f} us/step   total {t_all * 1e6:7.1f} us/step")
run("encode only", lambda: encode_device(tok, batch), lambda: None)
for d in (4, 8, 16, 32):
    gv = GatherV(tok, dev, max_docs=batch.n_docs, max_tokens=int(off[-1] * 1.02) + 64, depth=d)
    run(f"encode_and_submit, depth {d}", lambda: gv.encode_and_submit(batch), gv.finish)
dist.destroy_process_group()
