"""N>1 path on CPU: world_size-2 gloo processes run the sharding + ragged all-gather logic of
splintr_amd/distributed.py with the oracle standing in for the per-rank GPU encoder."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle.coracle import COracle
    from splintr_amd import corpus
    from splintr_amd.distributed import encode_batch_sharded
    orc = COracle("cl100k_base")
    texts = corpus.c2(37) + ["", "x"] + corpus.c4(50)

    def encode_csr(local):
        bs = [t.encode("utf-8") for t in local]
        off = np.zeros(len(bs) + 1, dtype=np.uint64)
        if bs:
            np.cumsum([len(b) for b in bs], out=off[1:])
        return orc.encode_packed(np.frombuffer(b"".join(bs), dtype=np.uint8), off)

    ids, off = encode_batch_sharded(encode_csr, texts, torch.device("cpu"))
    want = orc.encode_batch(texts)
    got = [ids[int(off[i]):int(off[i + 1])].tolist() for i in range(len(texts))]
    q.put((rank, got == want, len(texts), int(off[-1])))
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_encode_allgatherv_gloo(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = [q.get(timeout=120) for _ in ps]
    for p in ps:
        p.join(timeout=60)
    assert all(ok for _, ok, _, _ in res), res
    assert len({(n, t) for _, _, n, t in res}) == 1


def test_shard_bounds_balance():
    from splintr_amd.distributed import shard_bounds
    rng = np.random.default_rng(0)
    sizes = rng.integers(0, 5000, size=1000).tolist()
    for world in (1, 2, 4, 8):
        b = shard_bounds(sizes, world)
        assert b[0] == 0 and b[-1] == 1000 and all(x <= y for x, y in zip(b, b[1:])) and len(b) == world + 1
        per = [sum(sizes[b[r]:b[r + 1]]) for r in range(world)]
        assert max(per) - min(per) <= 2 * max(sizes)
    assert shard_bounds([], 4) == [0, 0, 0, 0, 0]
    assert shard_bounds([10], 4)[-1] == 1
