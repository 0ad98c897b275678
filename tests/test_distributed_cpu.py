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
    # the last documents are large: shard boundaries fall inside them and are moved to a newline + letter/digit
    texts = corpus.c2(37) + ["", "x"] + corpus.c4(50) + corpus.c5(2, seed=3, doc_bytes=60000)

    def encode_csr(local):
        bs = [t.encode("utf-8") for t in local]
        off = np.zeros(len(bs) + 1, dtype=np.uint64)
        if bs:
            np.cumsum([len(b) for b in bs], out=off[1:])
        return orc.encode_packed(np.frombuffer(b"".join(bs), dtype=np.uint8), off)

    # (a plain closure says nothing about its split pattern, so documents would stay whole: the cl100k oracle's cuts ARE context-free)
    ids, off = encode_batch_sharded(encode_csr, texts, torch.device("cpu"), context_free_cuts=True)
    want = orc.encode_batch(texts)
    got = [ids[int(off[i]):int(off[i + 1])].tolist() for i in range(len(texts))]
    # the same batch exchanged in WAVES (the host-tensor form of splintr_amd.device.WaveGather: wave k lands behind the waves before
    # it, offsets rebased per wave -- VERDICT r04 #2): the whole CSR in document order on every rank, for several wave counts
    from splintr_amd.distributed import encode_batch_waves, plan_waves
    ok_w = True
    for n_waves, taper in ((1, 1.0), (3, 1.0), (8, 1.0), (4, 0.6), (7, 0.7)):    # equal and TAPERED waves
        w_ids, w_off = encode_batch_waves(encode_csr, texts, torch.device("cpu"), n_waves=n_waves, taper=taper)
        got_w = [w_ids[int(w_off[i]):int(w_off[i + 1])].tolist() for i in range(len(texts))]
        ok_w = ok_w and got_w == want and len(w_off) == len(texts) + 1
        lens = [len(t.encode("utf-8")) for t in texts]
        pw = plan_waves(lens, world, n_waves, taper)
        flat = [d for k in range(n_waves) for r in range(world) for d in range(*pw[k][r])]
        ok_w = ok_w and flat == list(range(len(texts)))                     # the slices tile the documents, in order
    q.put((rank, got == want and ok_w, len(texts), int(off[-1])))
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


class _CustomEncoder:
    """Stands in for a Tokenizer with a custom split pattern: the Python oracle with a pattern whose matches span
    "newline + letter" -- the position plan_shards cuts at -- so that a cut inside a document changes the ids."""
    has_custom_pattern = True
    PATTERN = r"[^\n]+\n*\p{L}+|\s+|[^\s]+"

    def __init__(self):
        from oracle import pyoracle as O
        enc, _ = O.load_splv(os.path.join(ROOT, "splintr_amd", "data", "cl100k_base.splv"))
        self.orc = O.Oracle(enc, self.PATTERN, False, {}, "regex")

    def encode_csr(self, local):
        rows = [self.orc.encode(t) for t in local]
        off = np.zeros(len(rows) + 1, dtype=np.uint64)
        if rows:
            np.cumsum([len(r) for r in rows], out=off[1:])
        return np.asarray([i for r in rows for i in r], dtype=np.uint32), off


def _custom_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from splintr_amd.distributed import encode_batch_sharded
    e = _CustomEncoder()
    # two large documents made of "line\nWord" joints: every cut candidate sits inside a match of the pattern
    texts = ["ab cd;\nEf gh. " * 900, "", "x);\ny " * 1500, "tail"]
    want = [e.orc.encode(t) for t in texts]
    ids, off = encode_batch_sharded(e.encode_csr, texts, torch.device("cpu"))          # bound method: asks the encoder
    got = [ids[int(off[i]):int(off[i + 1])].tolist() for i in range(len(texts))]
    # and the same call told (wrongly) that the cuts are context-free does diverge: the guard is what keeps it equal
    ids2, off2 = encode_batch_sharded(e.encode_csr, texts, torch.device("cpu"), context_free_cuts=True)
    got2 = [ids2[int(off2[i]):int(off2[i + 1])].tolist() for i in range(len(texts))]
    # the encoder behind functools.partial / a wrapper still answers for itself; an opaque lambda keeps the documents whole
    import functools
    ids3, off3 = encode_batch_sharded(functools.partial(e.encode_csr), texts, torch.device("cpu"))
    ids4, off4 = encode_batch_sharded(lambda local: e.encode_csr(local), texts, torch.device("cpu"))
    got3 = [ids3[int(off3[i]):int(off3[i + 1])].tolist() for i in range(len(texts))]
    got4 = [ids4[int(off4[i]):int(off4[i + 1])].tolist() for i in range(len(texts))]
    q.put((rank, got == want and got3 == want and got4 == want, got2 != want))
    dist.destroy_process_group()


def test_custom_pattern_documents_are_never_cut_gloo():
    """a custom pattern has no known context-free cut; sharded == unsharded only with whole documents."""
    pytest.importorskip("regex")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_custom_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = [q.get(timeout=180) for _ in ps]
    for p in ps:
        p.join(timeout=60)
    assert all(ok for _, ok, _ in res), res
    assert all(div for _, _, div in res), "the test corpus no longer exercises the guard"


class _Ev:
    def record(self, stream=None):
        pass


class _St:
    cuda_stream = 0

    def wait_event(self, ev):
        pass


class _Ctx:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


def _gatherv_worker(rank, world, port, q):
    """GatherV's bucket / set / callback logic with a stub encoder on CPU tensors: pack, exchange
    (all_gather_into_tensor on gloo) and unpack follow the slab format of include/splintr_hip.h."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from splintr_amd.device import GatherV

    class Batch:
        def __init__(self, seed):
            rng = np.random.default_rng(seed)
            self.n_docs = int(rng.integers(0, 7))
            counts = rng.integers(0, 9, size=self.n_docs)
            self.off = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
            self.ids = rng.integers(0, 1000, size=int(self.off[-1])).astype(np.int32)

    class CpuGatherV(GatherV):
        def _new_stream(self):
            return _St()

        def _new_event(self):
            return _Ev()

        def _main_stream(self):
            return _St()

        def _stream_ctx(self, st):
            return _Ctx()

        def _encode_packed(self, batch, slab, with_special, stream_ptr):
            slab[0], slab[1] = int(batch.off[-1]), batch.n_docs
            slab[2:2 + batch.n_docs + 1] = torch.from_numpy(batch.off.astype(np.int32))
            at = 3 + self.max_docs
            slab[at:at + len(batch.ids)] = torch.from_numpy(batch.ids)

        def _unpack(self, s, n, stream_ptr):
            recv = self.recv[s].view(self.world, self.depth, self.cap_words)
            for j in range(n):
                ids, off = self._views(s, j)
                tb = db = 0
                for r in range(self.world):
                    sl = recv[r, j]
                    T, N = int(sl[0]), int(sl[1])
                    off[db:db + N] = sl[2:2 + N].to(torch.int64) + tb
                    at = 3 + self.max_docs
                    ids[tb:tb + T] = sl[at:at + T]
                    tb += T
                    db += N
                off[db] = tb

    depth, n_batches = 3, 8                     # 8 batches: two full buckets and a partial one
    gv = CpuGatherV(None, torch.device("cpu"), max_docs=8, max_tokens=80, depth=depth)
    got = []
    gv.on_bucket = lambda res: got.extend((i.clone(), o.clone()) for i, o in res)
    mine = [Batch(1000 * rank + k) for k in range(n_batches)]
    for b in mine:
        gv.encode_and_submit(b)
    last_ids, last_off = gv.finish()
    ok = len(got) == n_batches
    for k in range(n_batches):                  # every rank can rebuild every rank's batch k
        exp_ids, exp_off, tb = [], [0], 0
        for r in range(world):
            b = Batch(1000 * r + k)
            exp_ids.append(b.ids)
            exp_off.extend((b.off[1:] + tb).tolist())
            tb += int(b.off[-1])
        exp_ids = np.concatenate(exp_ids) if exp_ids else np.zeros(0, np.int32)
        g_ids, g_off = got[k]
        nd = len(exp_off) - 1
        ok = ok and np.array_equal(g_ids[:tb].numpy(), exp_ids) and g_off[:nd + 1].tolist() == exp_off
    ok = ok and torch.equal(last_ids, got[-1][0]) and torch.equal(last_off, got[-1][1])
    q.put((rank, bool(ok)))
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_gatherv_buckets_and_sets_gloo(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_gatherv_worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = [q.get(timeout=120) for _ in ps]
    for p in ps:
        p.join(timeout=60)
    assert all(ok for _, ok in res), res


def test_plan_shards_cuts_documents_at_context_free_boundaries(coracle):
    from splintr_amd import corpus
    from splintr_amd.distributed import plan_shards
    docs = [t.encode() for t in corpus.c5(3, seed=11, doc_bytes=150000)] + [b"", b"tail"]
    total = sum(map(len, docs))
    orc = coracle("deepseek_v3")
    for world in (1, 2, 4, 8):
        shards = plan_shards(docs, world)
        assert len(shards) == world
        flat = [p for s in shards for p in s]
        # the pieces tile every document in order
        rebuilt = {}
        for d, lo, hi in flat:
            assert lo == len(rebuilt.get(d, b"")) and hi >= lo
            rebuilt[d] = rebuilt.get(d, b"") + docs[d][lo:hi]
        assert all(rebuilt.get(d, b"") == docs[d] for d in range(len(docs)) if docs[d] or d in rebuilt)
        per = [sum(hi - lo for _, lo, hi in s) for s in shards]
        assert max(per) - min(per) < total // world // 4 + 4096          # byte-balanced although 3 docs dominate
        # ids of the pieces concatenate to the ids of the whole documents
        for d in range(3):
            pieces = [docs[d][lo:hi] for dd, lo, hi in flat if dd == d]
            if len(pieces) > 1:
                cat = [i for pc in pieces for i in orc.encode_bytes(pc)]
                assert cat == orc.encode_bytes(docs[d])
                assert all(pc[:1].isalnum() for pc in pieces[1:])
    assert plan_shards(docs, 4, split_docs=False) != plan_shards(docs, 4)
    assert plan_shards([], 3) == [[], [], []]


def test_wave_fractions_and_bounds():
    from splintr_amd.distributed import fraction_bounds, plan_waves, wave_fractions
    f = wave_fractions(7, 0.7)
    assert abs(sum(f) - 1.0) < 1e-12 and all(a > b for a, b in zip(f, f[1:])) and 0.03 < f[-1] < 0.05 and 0.30 < f[0] < 0.35
    assert wave_fractions(4) == [0.25] * 4
    lens = [100] * 1000
    b = fraction_bounds(lens, f)
    assert b[0] == 0 and b[-1] == 1000 and b == sorted(b)
    shares = [(b[k + 1] - b[k]) / 1000 for k in range(7)]
    assert all(abs(s_ - f_) < 0.002 for s_, f_ in zip(shares, f))
    pw = plan_waves(lens, 8, 7, 0.7)
    assert [d for k in range(7) for r in range(8) for d in range(*pw[k][r])] == list(range(1000))
    with pytest.raises(ValueError):
        plan_waves(lens, 8, 3, fractions=[0.5, 0.5])
    # degenerate inputs: fewer documents than waves, empty documents
    pw = plan_waves([5, 0, 7], 2, 5, 0.5)
    assert [d for k in range(5) for r in range(2) for d in range(*pw[k][r])] == [0, 1, 2]


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
