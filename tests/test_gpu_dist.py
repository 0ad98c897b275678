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
        # the same with the slab's ids packed three bytes each ("slab_pack24": a quarter less on the links), fused and separate pack
        gp = GatherV(tok, dev, max_docs=max(b.n_docs for b in batches), max_tokens=max_tok + 64, depth=2, pack24=True)
        assert gp.cap_words < gv.cap_words - max_tok // 5
        got.clear()
        gp.on_bucket = lambda res: got.extend((i.clone(), o.clone()) for i, o in res)
        for k in order:
            gp.encode_and_submit(batches[k])
        for k in (3, 1):
            encode_device(tok, batches[k])
            gp.submit(batches[k])
        gp.finish()
        torch.cuda.synchronize()
        assert not gp.overflowed() and len(got) == len(order) + 2
        for (g_ids, g_off), k in zip(got, order + [3, 1]):
            w_ids, w_off = want[k]
            assert np.array_equal(g_off[:batches[k].n_docs + 1].cpu().numpy().astype(np.uint64), w_off)
            assert np.array_equal(g_ids[:int(w_off[-1])].cpu().numpy().view(np.uint32), w_ids)
        # ... and the waves of ONE batch landing behind each other (WaveGather over the library's communicator, both slab formats)
        from splintr_amd.device import Comm, WaveGather
        comm = Comm(Comm.unique_id(), 0, 1, 0)
        tok_b = Tokenizer.from_pretrained("cl100k_base")          # a second handle: consecutive waves on two streams in alternation
        reserve(tok_b, max(b.n_bytes for b in batches), max(b.n_docs for b in batches))
        for p24, t2 in ((False, None), (True, None), (False, tok_b), (True, tok_b)):
            wg = WaveGather(tok, dev, comm, 5, max_docs=max(b.n_docs for b in batches), max_tokens=max_tok + 64,
                            total_tokens_cap=sum(int(w[1][-1]) for w in want) + 64, total_docs_cap=sum(b.n_docs for b in batches), pack24=p24,
                            tok2=t2)
            for _ in range(2):                                     # (twice: the running totals start over)
                wg.begin()
                for b in batches:
                    wg.encode_and_submit(b)
                a_ids, a_off, run = wg.finish()
            torch.cuda.synchronize()
            assert not wg.overflowed()
            w_all = np.concatenate([w[0] for w in want])
            assert run.cpu().tolist() == [len(w_all), sum(b.n_docs for b in batches)]
            assert np.array_equal(a_ids[:len(w_all)].cpu().numpy().view(np.uint32), w_all)
            o_all, tdone = [np.zeros(1, dtype=np.uint64)], 0
            for w in want:
                o_all.append(w[1][1:] + np.uint64(tdone))
                tdone += int(w[1][-1])
            assert np.array_equal(a_off.cpu().numpy().astype(np.uint64), np.concatenate(o_all))
        comm.close()
        _ffi_reset = __import__("splintr_amd")._ffi.lib().spl_set_option(tok.handle, b"slab_pack24", 0)
        assert _ffi_reset == 0
        # a slab that is too small is reported, not silently truncated
        small = GatherV(tok, dev, max_docs=max(b.n_docs for b in batches), max_tokens=100, depth=1)
        small.encode_and_submit(batches[0])
        small.finish()
        torch.cuda.synchronize()
        assert small.overflowed()
    finally:
        dist.destroy_process_group()


def test_pick_stream_returns_a_stream_that_runs_beside_the_given_ones():
    """spl_pick_stream (include/splintr_hip.h): a stream measured to run beside the given ones -- usable as a torch stream, work on it and
    on the given stream completes, and what conflict is left is small (on a GPU nobody else uses: none)."""
    import torch
    from splintr_amd.device import pick_stream
    dev = torch.device("cuda", 0)
    a = torch.cuda.Stream(dev)
    b = pick_stream(dev, [a])
    c = pick_stream(dev, [a, b])
    assert len({a.cuda_stream, b.cuda_stream, c.cuda_stream}) == 3
    assert b.conflict_us < 8.0 and c.conflict_us < 8.0, (b.conflict_us, c.conflict_us)
    x = torch.zeros(1 << 20, device=dev)
    torch.cuda.synchronize()
    for s in (a, b, c):
        with torch.cuda.stream(s):
            x.add_(1)                            # (racy on purpose? no: the three adds are ordered by the events below)
            ev = torch.cuda.Event(); ev.record(s)
        for o in (a, b, c):
            o.wait_event(ev)
    torch.cuda.synchronize()
    assert float(x[0].item()) == 3.0 and float(x.sum().item()) == 3.0 * (1 << 20)


def test_collective_behind_the_c_abi_one_rank(coracle):
    """spl_comm_* / spl_allgather_slabs / spl_allgatherv_csr (include/splintr_hip.h) on a one-rank communicator the
    LIBRARY creates (librccl bound by dlopen, no torch.distributed anywhere): the bucketed GatherV pipeline through
    spl_allgather_slabs and the exact all-gatherv (counts, then send/recv of exactly T ids and N offsets -- to
    itself here) against the oracle."""
    import torch
    from splintr_amd import Tokenizer, corpus, _ffi
    from splintr_amd.device import Comm, DeviceBatch, GatherV, encode_device, reserve
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    comm = Comm(Comm.unique_id(), 0, 1, 0)
    try:
        L = _ffi.lib()
        assert L.spl_comm_rank(comm.handle) == 0 and L.spl_comm_world(comm.handle) == 1
        tok = Tokenizer.from_pretrained("o200k_base")
        sets = [corpus.c3(40, seed=70 + k) + ["", "x"] for k in range(3)]
        batches = [DeviceBatch(t, dev) for t in sets]
        reserve(tok, max(b.n_bytes for b in batches), max(b.n_docs for b in batches))
        want = [oracle_csr(coracle("o200k_base"), t) for t in sets]
        gv = GatherV(tok, dev, max_docs=max(b.n_docs for b in batches), max_tokens=max(int(w[1][-1]) for w in want) + 64,
                     depth=2, comm=comm)
        got = []
        gv.on_bucket = lambda res: got.extend((i.clone(), o.clone()) for i, o in res)
        order = [0, 1, 2, 1, 0]
        for k in order:
            gv.encode_and_submit(batches[k])
        gv.finish()
        torch.cuda.synchronize()
        assert not gv.overflowed() and len(got) == len(order)
        for (g_ids, g_off), k in zip(got, order):
            w_ids, w_off = want[k]
            assert np.array_equal(g_off[:batches[k].n_docs + 1].cpu().numpy().astype(np.uint64), w_off)
            assert np.array_equal(g_ids[:int(w_off[-1])].cpu().numpy().view(np.uint32), w_ids)
        # the exact form
        for k in (2, 0):
            b = batches[k]
            w_ids, w_off = want[k]
            encode_device(tok, b)
            all_ids = torch.full((int(w_off[-1]) + 7,), -1, dtype=torch.int32, device=dev)
            all_off = torch.full((b.n_docs + 1,), -1, dtype=torch.int64, device=dev)
            nt, nd = comm.allgatherv_csr(b.ids, b.out_off, b.n_docs, all_ids, all_off)
            torch.cuda.synchronize()
            assert (nt, nd) == (int(w_off[-1]), b.n_docs)
            assert np.array_equal(all_ids[:nt].cpu().numpy().view(np.uint32), w_ids)
            assert np.array_equal(all_off.cpu().numpy().astype(np.uint64), w_off)
            assert int(all_ids[nt].item()) == -1                      # nothing written behind the last token
        # buffers that cannot hold the result: refused with SPL_ECAPACITY, nothing hangs
        with pytest.raises(RuntimeError, match="does not fit"):
            comm.allgatherv_csr(batches[0].ids, batches[0].out_off, batches[0].n_docs,
                                torch.empty(8, dtype=torch.int32, device=dev), torch.empty(4, dtype=torch.int64, device=dev))
        # an empty shard
        e = DeviceBatch([], dev)
        encode_device(tok, e)
        nt, nd = comm.allgatherv_csr(e.ids, e.out_off, 0, torch.empty(8, dtype=torch.int32, device=dev),
                                     torch.full((1,), -1, dtype=torch.int64, device=dev))
        assert (nt, nd) == (0, 0)
    finally:
        comm.close()


def test_bench_distributed_branch_rehearsal_world_1():
    """VERDICT r03 #3b: the exact code the driver launches at N = 8 -- bench.py's `use_dist` branches: the library's RCCL
    communicator, the bucketed slab exchange of the weak-scaling headline, spl_allgatherv_csr inside the C4 / C5
    strong-scaling steps, the per-rank encode / exchange breakdown -- runs here at world 1 (SPL_BENCH_FORCE_DIST=1)
    with small shards, in its own process, and its line must parse and carry the diagnostics a first curve needs."""
    import json
    import subprocess
    import sys
    from conftest import ROOT
    env = dict(os.environ, SPL_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()),
               RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--docs", "200", "--steps", "24", "--warmup", "8", "--regions", "3",
           "--c4-steps", "3", "--c5-steps", "3", "--c4-part-docs", "12000", "--c5-docs", "8", "--c5-doc-bytes", "1000000",
           "--no-cpu-baseline", "--no-throughputs", "--no-c2-wide"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert p.returncode == 0, (p.returncode, p.stdout[-1500:], p.stderr[-3000:])
    line = json.loads(p.stdout.strip().splitlines()[-1])
    assert line["rehearsal"] is True and line["n_gpus"] == 1 and line["value"] > 0
    assert "bit-exact" in line["parity"] and "RCCL all-gatherv" in line["config"]["workload"]
    assert line["timing"]["regions"] == 3 and len(line["timing"]["region_values"]) == 3
    d = line["dist"]
    assert d["rccl_ranks"] == 1 and d["torch_world"] == 1
    for key in ("step_ms", "encode_only_ms", "exchange_stream_ms_per_step"):
        assert len(d["per_rank"][key]) == 1 and d["per_rank"][key][0] > 0, (key, d)
    # a slab holds the batch's ids: four bytes each, three with the packed format
    assert d["slab_bytes_sent_per_batch"] >= d["ids_bytes_per_batch_4T"] * (3 if d["pack24"] else 4) // 4 > 0 and d["buckets_timed"] == 24 // d["bucket_depth"]
    # the start-up calibration: every depth x collective form was timed, the fastest one ran the timed region
    cal = d["calibration"]
    assert len(cal["ms_per_step"]) == 6 and all(v > 0 for v in cal["ms_per_step"].values())
    assert cal["chosen"] == min(cal["ms_per_step"], key=cal["ms_per_step"].get) and cal["chosen"].startswith(f"depth{d['bucket_depth']}_{d['collective']}")
    for key in ("c4_strong", "c5_strong"):
        c = line[key]
        assert c["scaling"] == "strong" and c["value"] > 0 and "bit-exact" in c["parity"], c
        cd = c["dist"]
        assert cd["rccl_ranks"] == 1 and cd["per_rank"]["exchange_stream_ms"][0] > 0 and cd["waves"] == 8, c
        assert isinstance(cd["stream_picks"], list) and len(cd["stream_picks"]) >= 1 and cd["least_bad_pick"] in (True, False)
        assert set(cd["calibration_ms_per_step"]) == {"allgather", "p2p", "allgather+pack24"} | {f"{f}+2s@{p}" for p in range(3) for f in ("allgather", "allgather+pack24")}
        assert cd["collective"] in ("allgather", "p2p") and cd["encode_streams"] in (1, 2)
        assert cd["bytes_received_per_rank"] >= cd["ids_bytes_4T"] * (3 if cd["pack24"] else 4) // 4 > 0
        # pipelined: what the exchange adds to a step is (at most) the last wave's exchange, not all of it (<= 5 % of the
        # step at full size; the rehearsal's waves are 2-3 MB, a step is under a millisecond -- eight waves of 80 us each -- and the exchange with itself shares the GPU with the encodes: 55 %; full size at world 1, 8 %: profiles/r06_rehearsal_world1.json)
        # (only when the timed steps ran at the speed the calibration saw a moment earlier: run behind the other tests of this file, whose RCCL
        #  communicator is still alive in the parent process, the first steps on the GPU have been seen to take 20 ms each)
        if max(cd["per_rank"]["step_ms"]) <= 2.0 * min(cd["calibration_ms_per_step"].values()):
            assert cd["exposed_exchange_ms"] <= 0.55 * max(cd["per_rank"]["step_ms"]), cd
