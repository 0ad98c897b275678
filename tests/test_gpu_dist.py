"""GPU test (-m gpu) of the pipelined ragged all-gather (splintr_amd.device.GatherV) on a real RCCL
process group of ONE rank: streams, events, both bucket sets, the per-bucket callback and the
slab written by the encoder's last kernel, every batch's global CSR against the oracle.  (More than
one rank needs more than one GPU: the bucket / set logic at world 2 and 3 runs on gloo in
tests/test_distributed_cpu.py, the slab kernels with two simulated ranks in test_gpu_parity.py.)"""
import os
import socket

import numpy as np
import pytest

from test_gpu_parity import oracle_csr

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_gatherv_pipeline_one_rank_rccl(coracle):
    import torch
    import torch.distributed as dist
    from splintr_amd import Tokenizer, corpus
    from splintr_amd.device import DeviceBatch, GatherV, reserve
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(_free_port())
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        tok = Tokenizer.from_pretrained("cl100k_base")
        sets = [corpus.c2(120, seed=50 + k) + ["", "x"] for k in range(5)]
        batches = [DeviceBatch(t, dev) for t in sets]
        reserve(tok, max(b.n_bytes for b in batches), max(b.n_docs for b in batches))
        want = [oracle_csr(coracle("cl100k_base"), t) for t in sets]
        max_tok = max(int(w[1][-1]) for w in want)
        gv = GatherV(tok, dev, max_docs=max(b.n_docs for b in batches), max_tokens=max_tok + 64, depth=2)
        got = []
        gv.on_bucket = lambda res: got.extend((i.clone(), o.clone()) for i, o in res)
        order = [0, 1, 2, 3, 4, 0, 2, 4, 1]                       # 9 batches: four full buckets + a partial one
        for k in order:
            gv.encode_and_submit(batches[k])
        last_ids, last_off = gv.finish()
        torch.cuda.synchronize()
        assert not gv.overflowed() and len(got) == len(order)
        for (g_ids, g_off), k in zip(got, order):
            w_ids, w_off = want[k]
            nd = batches[k].n_docs
            assert np.array_equal(g_off[:nd + 1].cpu().numpy().astype(np.uint64), w_off)
            assert np.array_equal(g_ids[:int(w_off[-1])].cpu().numpy().view(np.uint32), w_ids)
        assert torch.equal(last_ids, got[-1][0]) and torch.equal(last_off, got[-1][1])
        # the separate pack launch (submit) gives the same slabs as the fused one
        got.clear()
        from splintr_amd.device import encode_device
        for k in (3, 1):
            encode_device(tok, batches[k])
            gv.submit(batches[k])
        gv.finish()
        torch.cuda.synchronize()
        for (g_ids, g_off), k in zip(got, (3, 1)):
            w_ids, w_off = want[k]
            assert np.array_equal(g_ids[:int(w_off[-1])].cpu().numpy().view(np.uint32), w_ids)
        # a slab that is too small is reported, not silently truncated
        small = GatherV(tok, dev, max_docs=max(b.n_docs for b in batches), max_tokens=100, depth=1)
        small.encode_and_submit(batches[0])
        small.finish()
        torch.cuda.synchronize()
        assert small.overflowed()
    finally:
        dist.destroy_process_group()
