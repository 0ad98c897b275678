#!/usr/bin/env python3
"""encode_batch throughput on MI355X (BASELINE.json metric: MB/s of input bytes, MB = 1e6 B).

A "step" is ONE pass of the hot path (Tokenizer.encode_batch semantics through the C ABI,
spl_encode_batch_device) over ONE batch of synthetic text already resident in HBM:
config[1] of BASELINE.json = cl100k_base, 1000 x ~1 KB mixed English/code (splintr_amd.corpus.c2).
The timed loop rotates over EIGHT distinct batches of that shape (different seeds), so no step
re-encodes text that the previous step left in L2 / Infinity Cache.  With --gpus N > 1 every rank
holds its own eight 1000-document shards (weak scaling) and the step also all-gathers the ragged
ids over RCCL so that every rank ends up with the whole CSR result of every batch.

    python bench.py --gpus 1 --steps 500 --warmup 50
    python bench.py --gpus N ...          (no WORLD_SIZE in the environment: re-executes itself under
                                           torch.distributed.run, one rank per GPU, and relays rank 0's line)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0.  Beside the contract's keys it carries
  roofline       HBM roofline of the dominant kernel (algorithmic bytes / its duration in this run)
  roofline_valu  the binding one: VALU issue (counters from the committed rocprofv3 pass it names)
  throughputs    SURVEY 8d's three figures for C2 and C3: kernels only / C ABI host->host / Python surface
  c4_strong      BASELINE config 4 (llama3, 1 M short prompts) doc-sharded over the N ranks (strong scaling)
  c5_strong      BASELINE config 5 (deepseek_v3, 100 x 2 MiB documents) byte-sharded over the N ranks, documents
                 cut at context-free boundaries where a shard boundary falls inside one (encode_rayon's case)
  cpu_baseline   the oracle's C port of the reference's Rayon path on this box's host cores
The oracle (oracle/) is used only as the checker of the untimed verification passes and as the timed
CPU baseline ("port"); it is never on the measured path.
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
# HIP multiplexes a process's streams onto FOUR hardware queues by default; the distributed legs keep an encode stream or two, an exchange
# stream and RCCL's own busy at once.  Eight queues, unless the caller chose: fewer streams share one.  (What matters more -- which busy
# streams share a PIPE of the command processor -- cannot be set from here: the strong legs' calibration tries three pairs of encode streams,
# profiles/r05_wave_exchange.txt.)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E peak (MI355X_MICROARCH.md)
N_ROT = 8               # distinct batches in the timed rotation
C4_PARTS = 8            # the C4 batch is generated as 8 x 125 000 prompts (seed 1004 + part)
C4_PART_DOCS = 125_000


C5_DOCS = 100           # BASELINE config 5: 100 documents of 2 MiB (seed 1005 + document index)
C5_DOC_BYTES = 2 << 20


def _c4_part(k):
    from splintr_amd import corpus
    return corpus.c4(C4_PART_DOCS, seed=1004 + k)


def _c5_doc(k):
    from splintr_amd import corpus
    return corpus.c5(1, seed=1005 + k, doc_bytes=C5_DOC_BYTES)[0]


def _free_port():
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def self_launch(args, argv):
    """`python bench.py --gpus N` with no WORLD_SIZE in the environment: this process becomes the launcher --
    it re-executes bench.py under torch.distributed.run (one rank per GPU, rendezvous on 127.0.0.1) and relays
    the ranks' output; rank 0 prints the one JSON line."""
    import subprocess
    if not args.launch_selftest:
        have = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if have < args.gpus:
            raise SystemExit(f"bench.py --gpus {args.gpus} needs {args.gpus} GPUs, this box has {have}")
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + argv
    return subprocess.call(cmd, env=env)


def launch_selftest(rank, world):
    """CPU-side check of the launcher (tests/test_bench_launcher.py): the ranks rendezvous on gloo, agree on
    the world size and rank 0 prints one JSON line -- everything self_launch() sets up, without a GPU."""
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    t = torch.tensor([rank + 1], dtype=torch.int64)
    dist.all_reduce(t)
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"launcher": "ok", "n_gpus": world, "rank_sum": int(t.item())}), flush=True)


def _packed(texts):
    bs = [t.encode("utf-8") for t in texts]
    off = np.zeros(len(bs) + 1, dtype=np.uint64)
    if bs:
        np.cumsum([len(b) for b in bs], out=off[1:])
    return np.frombuffer(b"".join(bs), dtype=np.uint8), off


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--inflight", type=int, default=3,
                    help="N > 1: also report the steps with N batches in flight on their own handles and streams "
                         "(\"pipelined\", supplementary: `value` stays one batch at a time; skipped with --no-throughputs so that a profile sees one batch at a time)")
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--regions", type=int, default=15,
                    help="the timed region of --steps steps is run this many times back to back; `value` is the median region")
    ap.add_argument("--docs", type=int, default=1000, help="documents per batch (BASELINE config: 1000)")
    ap.add_argument("--corpus", choices=("c2", "c2_wide"), default="c2",
                    help="c2 = BASELINE config 2 (the headline); c2_wide = the same mix over a >= 20 000-word lexicon (profiles only: the line then says so)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-c2-wide", action="store_true", help="skip the c2_wide rotation leg")
    ap.add_argument("--no-throughputs", action="store_true", help="skip the C-ABI / Python-surface figures (C2, C3)")
    ap.add_argument("--no-c4", action="store_true", help="skip the doc-sharded 1 M-prompt run (BASELINE config 4)")
    ap.add_argument("--c4-steps", type=int, default=5)
    ap.add_argument("--no-c5", action="store_true", help="skip the 100 x 2 MiB run (BASELINE config 5)")
    ap.add_argument("--c5-steps", type=int, default=5)
    ap.add_argument("--launch-selftest", action="store_true", help=argparse.SUPPRESS)
    # rehearsal sizes (tests/test_gpu_dist.py runs the distributed branches at world 1 with small shards; the line then
    # carries "rehearsal": true -- the BASELINE configurations are the defaults)
    ap.add_argument("--c4-part-docs", type=int, default=C4_PART_DOCS, help=argparse.SUPPRESS)
    ap.add_argument("--c5-docs", type=int, default=C5_DOCS, help=argparse.SUPPRESS)
    ap.add_argument("--c5-doc-bytes", type=int, default=C5_DOC_BYTES, help=argparse.SUPPRESS)
    args = ap.parse_args()
    rehearsal = (args.c4_part_docs, args.c5_docs, args.c5_doc_bytes) != (C4_PART_DOCS, C5_DOCS, C5_DOC_BYTES)
    globals().update(C4_PART_DOCS=args.c4_part_docs, C5_DOCS=args.c5_docs, C5_DOC_BYTES=args.c5_doc_bytes)   # (forked generators read them)

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(self_launch(args, sys.argv[1:]))
    if args.launch_selftest:
        return launch_selftest(rank, world)
    if world != args.gpus:
        args.gpus = world
    # the C4 prompts are generated by forked workers BEFORE anything initialises the GPU or RCCL: a fork of a
    # process that already runs HIP / RCCL threads is not something to rely on
    c4_texts = c5_pieces = None
    if not args.no_c4:
        c4_texts = gen_c4_texts(rank, world)
    if not args.no_c5:
        c5_pieces = gen_c5_pieces(rank, world)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product has no CPU path)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or os.environ.get("SPL_BENCH_FORCE_DIST") == "1"   # the latter: 1-rank smoke of the RCCL path
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    # every leg launches on a stream of its own, never on the NULL stream: a launch there synchronises with the other streams' work (one encode
    # stream beside the exchange: 1.80 ms per WaveGather step on the null stream, 1.06 on any other; one batch at a time with nothing else
    # running: 30.9 against 30.7 us per step -- tools/dev/null_vs_side.py)
    torch.cuda.set_stream(torch.cuda.Stream(dev))

    import __graft_entry__ as entry
    if rank == 0:
        entry.build()
    if use_dist:
        dist.barrier()
    from splintr_amd import Tokenizer, corpus, _ffi
    from splintr_amd.device import Comm, DeviceBatch, GatherV, encode_device, reserve, result_csr
    from oracle.coracle import COracle

    ncpu = os.cpu_count() or 1
    tok = Tokenizer.from_pretrained("cl100k_base", device=local_rank)
    # eight rank-distinct batches of the same distribution
    gen_, seed0_ = (corpus.c2, 1002) if args.corpus == "c2" else (corpus.c2_wide, 2002)
    text_sets = [gen_(args.docs, seed=seed0_ + 100 * rank + k) for k in range(N_ROT)]
    batches = [DeviceBatch(t, dev) for t in text_sets]
    reserve(tok, max(b.n_bytes for b in batches), max(b.n_docs for b in batches))
    L = _ffi.lib()
    gv = None
    state = {"i": 0}

    def step():
        b = batches[state["i"] % N_ROT]
        state["i"] += 1
        if gv is None:
            encode_device(tok, b)
        else:
            # encode with the slab written by the encoder's last kernel; every 8th batch: one RCCL
            # all-gather of the bucket + one unpack launch, on a stream of their own
            gv.encode_and_submit(b)

    # ---- untimed verification pass: every batch of the rotation bit-exact vs the oracle ------------
    orc = COracle("cl100k_base")
    n_tokens, csr = [], []
    for b, texts in zip(batches, text_sets):
        encode_device(tok, b)
        torch.cuda.synchronize()
        ids, off = result_csr(b)
        text_np, _ = _packed(texts)
        o_ids, o_off = orc.encode_packed(text_np, b.host_offsets, threads=ncpu)
        if not (np.array_equal(ids, o_ids) and np.array_equal(off, o_off)):
            raise SystemExit(f"rank {rank}: HIP result differs from the oracle -- refusing to report a throughput")
        n_tokens.append(int(off[-1]))
        csr.append((ids, off))
    bytes_rot = sum(b.n_bytes for b in batches)

    if use_dist:
        # size the slabs from the largest shard (one-time, untimed), then check that the exchange
        # reproduces this rank's ids and offsets at its place in the global CSR of every batch
        mx = torch.tensor([max(n_tokens), max(b.n_docs for b in batches)], dtype=torch.int64, device=dev)
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        comm = Comm.from_torch_group(dev)       # the library's own RCCL communicator (spl_comm_*): torch only carries its 128-byte id
        # Untimed start-up calibration (VERDICT r04 #2b): bucket depth x collective form -- ncclAllGather of the bucket's slabs against
        # grouped send / recv of the same slabs -- on this very workload; the fastest (max over ranks) runs the timed region, all four
        # figures go into `dist.calibration`.  (xGMI is point to point: which form RCCL drives better depends on the message size.)
        cal = {}
        gv_max_docs, gv_max_tokens = int(mx[1].item()), int(int(mx[0].item()) * 1.02) + 64
        # calibration regions have the timed region's shape -- K steps, then the last bucket's exchange exposed -- because the best depth
        # depends on K: at the driver's K = 20 a bucket of 32 never fills and its one exchange is all exposed
        cal_regions = max(3, -(-64 // max(1, args.steps))) if not rehearsal else 2
        cal_steps = cal_regions * args.steps
        # (depths that leave a SMALL last bucket for this K -- its exchange is the one nothing hides -- beside the round ones)
        depths = sorted({2, 3, 4, 8, 32} | {d_ for d_ in (6, 9, 12) if 0 < args.steps % d_ <= d_ // 2}) if not rehearsal else (4, 8)
        for depth_c in depths:
            for form in ("allgather", "p2p", "allgather+pack24"):
                g_ = GatherV(tok, dev, max_docs=gv_max_docs, max_tokens=gv_max_tokens, comm=comm, depth=depth_c, collective=form.split("+")[0],
                             pack24=form.endswith("pack24"))
                for j in range(depth_c):
                    g_.encode_and_submit(batches[j % N_ROT])
                g_.finish()
                dist.barrier()
                torch.cuda.synchronize()
                c0 = time.perf_counter()
                for r_ in range(cal_regions):
                    for j in range(args.steps):
                        g_.encode_and_submit(batches[j % N_ROT])
                    g_.finish()
                    torch.cuda.synchronize()
                ct = torch.tensor([time.perf_counter() - c0], dtype=torch.float64, device=dev)
                dist.all_reduce(ct, op=dist.ReduceOp.MAX)
                cal[(depth_c, form)] = float(ct.item()) / cal_steps * 1e3
                del g_
                torch.cuda.empty_cache()
        (best_depth, best_form), _ = min(cal.items(), key=lambda kv: kv[1])
        gv = GatherV(tok, dev, max_docs=gv_max_docs, max_tokens=gv_max_tokens, comm=comm, depth=best_depth, collective=best_form.split("+")[0],
                     pack24=best_form.endswith("pack24"))
        got = []
        gv.on_bucket = lambda res: got.extend((i.clone(), o.clone()) for i, o in res)
        for _ in range(N_ROT):
            step()
        gv.finish()
        torch.cuda.synchronize()
        gv.on_bucket = None
        assert not gv.overflowed() and len(got) == N_ROT
        for k, (g_ids, g_off) in enumerate(got):
            cnt = torch.tensor([n_tokens[k], batches[k].n_docs], dtype=torch.int64, device=dev)
            allc = [torch.zeros_like(cnt) for _ in range(world)]
            dist.all_gather(allc, cnt)
            allc = torch.stack(allc).cpu().numpy()
            t_before, d_before = int(allc[:rank, 0].sum()), int(allc[:rank, 1].sum())
            assert int(g_off[int(allc[:, 1].sum())].item()) == int(allc[:, 0].sum())
            ids, off = csr[k]
            assert np.array_equal(g_ids[t_before:t_before + n_tokens[k]].cpu().numpy().view(np.uint32), ids)
            assert np.array_equal((g_off[d_before:d_before + batches[k].n_docs + 1] - t_before).cpu().numpy().astype(np.uint64), off)
        state["i"] = 0

    # ---- timed region ----------------------------------------------------------------------------
    # The contract's region -- W warm-up steps, then EXACTLY K steps between barrier + synchronize on both sides, the MAX
    # over ranks -- is run R = --regions times back to back and `value` is the MEDIAN region (one 20-step
    # region of a 30 us step is 0.6 ms of wall clock, and the first one behind an idle GPU came out 5 % low two rounds
    # running).  Every region is reported (`regions`), with p10 / p90 beside the median.
    # (behind the verification passes -- seconds of CPU work, the GPU idle -- the first ~150 steps of a run came out 7 % slower than the
    #  rest, run after run: 37.3 - 38.3 GB/s for the first eight 20-step regions, 40.1 - 40.6 for the others (profiles/r06_bench_regions.txt):
    #  the GPU's clocks come up under load.  The same steps are therefore run untimed for ~30 ms first; the contract's W warm-up steps follow.)
    ramp0 = time.perf_counter()
    while time.perf_counter() - ramp0 < 0.03:
        for _ in range(16):
            step()
        if gv is not None:
            gv.finish()
        torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    reg_t, reg_b = [], []
    i0 = state["i"]
    for _ in range(max(1, args.regions)):
        if gv is not None:
            gv.finish()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()
        i0 = state["i"]
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        if gv is not None:
            gv.finish()                   # the last batch's exchange completes inside the timed region
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        reg_t.append(time.perf_counter() - t0)
        reg_b.append(sum(batches[(i0 + j) % N_ROT].n_bytes for j in range(args.steps)))
    local_t = list(reg_t)
    if use_dist:
        et = torch.tensor(reg_t, dtype=torch.float64, device=dev)
        dist.all_reduce(et, op=dist.ReduceOp.MAX)
        reg_t = [float(x) for x in et.tolist()]
        nb = torch.tensor(reg_b, dtype=torch.int64, device=dev)
        dist.all_reduce(nb)
        reg_b = [int(x) for x in nb.tolist()]
    rates = sorted((b_ / t_ / 1e6, t_, k) for k, (t_, b_) in enumerate(zip(reg_t, reg_b)))
    value, elapsed, k_med = rates[len(rates) // 2]           # the median region (R odd: a region that was measured, not an average)
    elapsed_local = local_t[k_med]
    ms_per_step = elapsed / args.steps * 1e3
    pq = lambda f: rates[min(len(rates) - 1, int(f * len(rates)))][0]
    region_stats = {"regions": len(rates), "steps_per_region": args.steps,
                    "value_p10_p90": [round(pq(0.1), 2), round(pq(0.9), 2)],
                    "region_values": [round(b_ / t_ / 1e6, 1) for t_, b_ in zip(reg_t, reg_b)],
                    "note": "`value` / `ms_per_step` are the MEDIAN of `regions` timed regions of `steps` steps each (every region bracketed by "
                            "barrier + synchronize, MAX over ranks); region_values in run order; ~30 ms of the same steps run untimed before the W warm-up steps "
                            "(the GPU's clocks come up under load: without them the first eight regions of a run are 7 % slower than the rest)"}

    # ---- distributed runs: what a rank's step is made of (untimed repeats of the same steps) -------------
    # encode_only_ms: the same steps without the exchange; exchange_stream_ms: what the exchange stream spent in the
    # collective + unpack, per step; exposed = step - encode-only.  Per rank, so that a first N-GPU curve says whether a
    # rank is encode-bound, link-bound, or waiting for a slower peer.
    dist_info = None
    if use_dist:
        local_ms = elapsed_local / args.steps * 1e3
        torch.cuda.synchronize()
        dist.barrier()
        e0 = time.perf_counter()
        for j in range(args.steps):
            encode_device(tok, batches[(i0 + j) % N_ROT])
        torch.cuda.synchronize()
        enc_ms = (time.perf_counter() - e0) / args.steps * 1e3
        dist.barrier()
        gv.enable_timing()
        for _ in range(args.steps):
            step()
        gv.finish()
        ex_total, ex_buckets = gv.exchange_ms()
        gv.enable_timing(False)
        mine = torch.tensor([local_ms, enc_ms, ex_total / args.steps], dtype=torch.float64, device=dev)
        allm = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allm, mine)
        allm = torch.stack(allm).cpu().numpy()
        sent, recvd = gv.bytes_per_bucket()
        tok_b = sum(n_tokens) / N_ROT
        dist_info = {"rccl_ranks": comm.ranks(), "torch_world": world,
                     "per_rank": {"step_ms": [round(float(x), 5) for x in allm[:, 0]],
                                  "encode_only_ms": [round(float(x), 5) for x in allm[:, 1]],
                                  "exchange_stream_ms_per_step": [round(float(x), 5) for x in allm[:, 2]]},
                     "exposed_exchange_ms": round(float(allm[:, 0].max() - allm[:, 1].max()), 5),
                     "stream_picks": _stream_picks(), "least_bad_pick": any(p_["least_bad"] for p_ in _stream_picks()),
                     "bucket_depth": gv.depth, "collective": gv.collective, "pack24": gv.pack24, "buckets_timed": ex_buckets,
                     "calibration": {"ms_per_step": {f"depth{d_}_{f_}": round(v_, 5) for (d_, f_), v_ in sorted(cal.items())},
                                     "chosen": f"depth{gv.depth}_{gv.collective}{'+pack24' if gv.pack24 else ''}", "steps_each": cal_steps, "regions_each": cal_regions,
                                     "note": "untimed, before the timed region: regions of the same K steps (+ the last bucket's exchange) with every bucket depth x collective form (ncclAllGather of the "
                                             "slabs / grouped ncclSend+ncclRecv of the same slabs / ncclAllGather of slabs whose ids are packed three bytes each); "
                                             "max over ranks; the fastest runs the timed region"},
                     "slab_bytes_sent_per_batch": sent // gv.depth, "bytes_received_per_batch": recvd // gv.depth,
                     "ids_bytes_per_batch_4T": round(4 * tok_b), "slab_over_4T": round(sent / gv.depth / (4 * tok_b), 4),
                     "note": "per batch and rank: one slab of cap_words u32 (T, N, local offsets, ids; sized 1.02 x the largest shard) out, "
                             "world slabs in; ONE all-gather per bucket of `bucket_depth` batches on its own stream"}

    # ---- the chunk memo: the same rotation with it OFF (what a batch costs cold), and what it holds ------------------------------
    memo_info = None
    if rank == 0 and world == 1 and not use_dist:
        ms_ = (ctypes.c_uint64 * 4)()
        L.spl_memo_stats.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint64)]
        L.spl_memo_stats(tok.handle, ms_)
        L.spl_set_option(tok.handle, b"memo", 0)
        for j in range(args.warmup):
            encode_device(tok, batches[j % N_ROT])
        off_rates = []
        for _ in range(5):
            torch.cuda.synchronize()
            o0 = time.perf_counter()
            for j in range(args.steps):
                encode_device(tok, batches[j % N_ROT])
            torch.cuda.synchronize()
            off_rates.append(sum(batches[j % N_ROT].n_bytes for j in range(args.steps)) / (time.perf_counter() - o0) / 1e6)
        L.spl_set_option(tok.handle, b"memo", 1)
        off_rates.sort()
        memo_info = {"fills": int(ms_[0]), "chunks_put_in": int(ms_[1]), "chunks_beyond_an_entry": int(ms_[2]), "entries": int(ms_[3]),
                     "value_memo_off": round(off_rates[len(off_rates) // 2], 2), "unit": "MB/s",
                     "note": "`value` is measured with the chunk memo WARM: the rotation's 8 batches were each encoded by the verification pass and the warm-up steps before "
                             "the timed regions, so every chunk the vocabulary does not hold as one token is in the memo (the reference's LRU of encoded chunks, "
                             "src/core/tokenizer.rs:707-722, is warm in its own benchmark in the same way).  value_memo_off: the same rotation with "
                             "spl_set_option(memo, 0), median of 5 regions -- what a batch of text never seen before costs, minus the one-off fills"}

    # ---- the lexically wide variant of the same mix, in rotation, timed the same way (single GPU) ---------
    # C2's 925 distinct words flatter the vocabulary probe (every whole-chunk probe hits); c2_wide draws the same mix from a
    # >= 20 000-word lexicon.  Reported in `c2_wide_rotation` and named in config.workload next to the headline.
    c2_wide_rot = None
    if rank == 0 and world == 1 and not use_dist and args.corpus == "c2" and not args.no_c2_wide:
        w_sets = [corpus.c2_wide(args.docs, seed=2002 + k) for k in range(N_ROT)]
        w_batches = [DeviceBatch(t, dev) for t in w_sets]
        reserve(tok, max(b.n_bytes for b in w_batches + batches), max(b.n_docs for b in w_batches + batches))
        for b, texts in zip(w_batches, w_sets):
            encode_device(tok, b)
            torch.cuda.synchronize()
            ids, off = result_csr(b)
            text_np, _ = _packed(texts)
            o_ids, o_off = orc.encode_packed(text_np, b.host_offsets, threads=ncpu)
            if not (np.array_equal(ids, o_ids) and np.array_equal(off, o_off)):
                raise SystemExit("c2_wide: HIP result differs from the oracle")
        for j in range(args.warmup):
            encode_device(tok, w_batches[j % N_ROT])
        w_rates = []
        for _ in range(max(1, args.regions)):
            torch.cuda.synchronize()
            w0_ = time.perf_counter()
            for j in range(args.steps):
                encode_device(tok, w_batches[j % N_ROT])
            torch.cuda.synchronize()
            w_el = time.perf_counter() - w0_
            w_rates.append(sum(w_batches[j % N_ROT].n_bytes for j in range(args.steps)) / w_el / 1e6)
        w_rates.sort()
        c2_wide_rot = {"value": round(w_rates[len(w_rates) // 2], 2), "unit": "MB/s",
                       "p10_p90": [round(w_rates[int(0.1 * len(w_rates))], 2), round(w_rates[min(len(w_rates) - 1, int(0.9 * len(w_rates)))], 2)],
                       "workload": f"splintr_amd.corpus.c2_wide, {args.docs} x ~1 KB, {N_ROT} batches in rotation (seeds 2002 + k), kernel-only, "
                                   f"median of {len(w_rates)} regions of {args.steps} steps; bit-exact vs oracle"}
        del w_batches, w_sets

    # ---- supplementary: the same steps with several batches in flight --------------------------------
    pipelined = None
    if rank == 0 and world == 1 and not use_dist and args.inflight > 1 and not args.no_throughputs:
        toks = [tok] + [Tokenizer.from_pretrained("cl100k_base", device=local_rank) for _ in range(args.inflight - 1)]
        from splintr_amd.device import pick_stream
        strs = [torch.cuda.current_stream(dev)]
        while len(strs) < args.inflight:          # streams that really run side by side (spl_pick_stream: measured)
            strs.append(pick_stream(dev, strs))
        for t_ in toks[1:]:
            reserve(t_, max(b.n_bytes for b in batches), max(b.n_docs for b in batches))

        def run(k):
            for i in range(k):
                j = i % args.inflight
                with torch.cuda.stream(strs[j]):
                    encode_device(toks[j], batches[i % N_ROT])
        run(args.warmup + args.inflight)
        p_rates = []
        for _ in range(max(1, args.regions)):
            torch.cuda.synchronize()
            p0 = time.perf_counter()
            run(args.steps)
            torch.cuda.synchronize()
            pel = time.perf_counter() - p0
            p_rates.append((sum(batches[i % N_ROT].n_bytes for i in range(args.steps)) / pel / 1e6, pel))
        p_rates.sort()
        p_val, pel = p_rates[len(p_rates) // 2]
        pipelined = {"inflight": args.inflight,
                     "value": round(p_val, 2), "unit": "MB/s",
                     "ms_per_step": round(pel / args.steps * 1e3, 5),
                     "note": f"SUPPLEMENTARY (never `value`): the same rotation and steps, round-robin over {args.inflight} handles on their own HIP streams, "
                             f"median of {len(p_rates)} regions -- the next batch's tile kernel fills the CUs that the stragglers of this one and k_tile_out leave idle"}
        del toks

    # ---- per-kernel durations: measured live over the same rotation (separate pass) ---------------------
    roofline = roofline_valu = None
    kernels = {}
    if rank == 0:
        L.spl_profile_enable(tok.handle, 1)
        L.spl_profile_reset(tok.handle)
        for j in range(13 * N_ROT):
            encode_device(tok, batches[j % N_ROT])
        torch.cuda.synchronize()
        ms = (ctypes.c_double * _ffi.SPL_MAX_KERNELS)()
        cnt = (ctypes.c_uint64 * _ffi.SPL_MAX_KERNELS)()
        L.spl_profile_read(tok.handle, ms, cnt)
        L.spl_profile_enable(tok.handle, 0)
        for i in range(_ffi.SPL_MAX_KERNELS):
            nm = L.spl_kernel_name(i)
            if nm and cnt[i]:
                kernels[nm.decode()] = round(ms[i] / cnt[i] * 1e3, 3)      # us per launch
        dom = max(kernels, key=kernels.get)
        # algorithmic bytes of one batch: text read once, u32 ids written once,
        # input and output offset arrays (u64 each); average over the rotation
        b_alg = (bytes_rot + 4 * sum(n_tokens) + 16 * sum(b.n_docs + 1 for b in batches)) / N_ROT
        achieved = b_alg / (kernels[dom] * 1e-6) / 1e9
        traffic, tsrc = None, None
        for cand in ("r06_hbm_traffic.json", "r05_hbm_traffic.json", "r04_hbm_traffic.json", "r03_hbm_traffic.json", "r02_hbm_traffic.json", "hbm_traffic.json"):
            tpath = os.path.join(ROOT, "profiles", cand)
            if os.path.exists(tpath):
                try:
                    traffic = json.load(open(tpath)).get(dom, {}).get("bytes_per_launch")
                    tsrc = "profiles/" + cand
                except Exception:
                    traffic = None
                if traffic is not None:
                    break
        roofline = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                    "traffic_source": (tsrc + " (rocprofv3 --pmc passes of this command, committed; NOT measured by this run)") if tsrc else None,
                    "algorithmic_bytes_per_launch": round(b_alg), "kernel_us": kernels[dom],
                    "all_kernels_us": kernels}
        # the roofline that binds this kernel: VALU issue.  Counters come from rocprofv3 --pmc (not
        # available inside a plain run): the committed pass over this very command.
        vname = next((n for n in ("r06_pmc_sq.json", "r05_pmc_sq.json", "r04_pmc_sq.json", "r03_pmc_sq.json", "r02_pmc_sq.json") if os.path.exists(os.path.join(ROOT, "profiles", n))), None)
        vpath = os.path.join(ROOT, "profiles", vname) if vname else ""
        if vname:
            try:
                v = json.load(open(vpath)).get(dom)
                if v:
                    roofline_valu = {"bound": "valu_issue", "kernel": dom,
                                     "SQ_INSTS_VALU": v["SQ_INSTS_VALU"], "SQ_ACTIVE_INST_VALU": v["SQ_ACTIVE_INST_VALU"],
                                     "kernel_cycles": v["kernel_cycles"], "simds": 1024,
                                     "frac": round(v["SQ_ACTIVE_INST_VALU"] * 4 / (1024 * v["kernel_cycles"]), 4),
                                     "lane_utilisation": v.get("lane_utilisation"),
                                     "source": f"profiles/{vname} (rocprofv3 --pmc passes of this command, committed; NOT measured by this run)"}
            except Exception:
                roofline_valu = None

    # ---- kernels only / C ABI host->host / Python surface, for C2 and C3 -------------------
    throughputs = None
    if rank == 0 and world == 1 and not use_dist and not args.no_throughputs:
        import host_path_bench
        throughputs = {}
        for cfg in ("c2", "c2_wide", "c3"):
            try:
                throughputs[cfg] = host_path_bench.measure(cfg)
            except Exception as e:          # a failure here must not cost the headline line
                throughputs[cfg] = {"error": repr(e)}
        try:
            throughputs["c2_custom_pattern"] = host_path_bench.measure_custom()
        except Exception as e:
            throughputs["c2_custom_pattern"] = {"error": repr(e)}
        throughputs["note"] = ("MB/s of input bytes. kernel_hbm: corpus resident in HBM (one batch, re-encoded); c_abi_host: "
                               "spl_encode_batch host bytes -> host CSR incl. H2D/D2H, input from spl_host_alloc; "
                               "c_abi_host_pageable: the same from pageable memory (one extra host copy into pinned staging); "
                               "python_surface: Tokenizer.encode_batch(list[str]) -> list[list[int]]; decode_host: spl_decode_batch, ids CSR on the host -> "
                               "bytes CSR on the host (pinned in and out), MB/s of decoded bytes; c2_wide: C2's mix over a >= 20 000-word "
                               "lexicon (splintr_amd.corpus.c2_wide); encode_one_call_us: Tokenizer.encode(text) on the batch's first document -- one GPU "
                               "round trip per call, a latency figure; c2_custom_pattern: the C2 batch through a handle with GPT-2's split pattern, which the "
                               "GPU scanner does not implement -- split_host: the host splitter alone, split_device: the device splitter alone (k_rx_match + "
                               "k_rx_mark, text in HBM), kernel_hbm_given_boundaries: the tile kernel on given chunk boundaries "
                               "(spl_encode_chunks_device), kernel_hbm_device_split: spl_encode_batch_device (device splitter + tile kernel), c_abi_host / "
                               "python_surface: the calls a user makes (device splitter), c_abi_host_host_split: the same with the split kept on the host cores")

    # ---- BASELINE config 4: llama3, 1 M short prompts, doc-sharded over the ranks (strong scaling) -------
    c4 = c5 = None
    if not args.no_c4:
        c4 = run_c4(args, rank, world, local_rank, dev, use_dist, c4_texts)
        del c4_texts
    if not args.no_c5:
        c5 = run_c5(args, rank, world, local_rank, dev, use_dist, c5_pieces)
        del c5_pieces

    # ---- CPU baseline: the oracle (a port of the reference's Rayon path) on the host cores ---------
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        # `value`: ONE batch of the rotation per call (1000 documents, ~1 MB: what the GPU's step is and what
        # north_star's ">= 10x on 1000 x 1 KB" is about), a FIXED number of repetitions per configuration, the
        # MEDIAN.  Beside it the whole rotation as ONE call (8000 documents, ~8 MB: a call of ~10 ms, where the
        # pool's wake-up no longer shows) and the 24-thread figure SURVEY 8d asks for.  Thread counts 8 .. all
        # host CPUs x memo on/off; the best median of each is reported (the GPU is compared with the strongest
        # configuration of the port on this host).
        from oracle.coracle import pool_pin, pool_set_cpus, idle_cpus
        from splintr_amd import _ffi as _f
        shim = _f.shim()
        one_np, one_off = _packed(text_sets[0])
        all_texts = [t for ts in text_sets for t in ts]
        all_np, all_off = _packed(all_texts)
        # the pool's CPUs: the ones this process may run on MINUS those that are busy or that this process's other threads (the GPU
        # runtime's among them) last ran on -- a worker pinned onto an occupied CPU waits a time slice, and a 1 ms call takes 10
        # (BENCH_r05: 64 threads at 99.8 MB/s on the driver's box, 1 106 on the builder's)
        cpus = idle_cpus()
        pool_set_cpus(cpus)
        cands = sorted({t for t in (8, 16, 24, 32, 64, 128, len(cpus)) if t <= len(cpus)})
        pool_pin(True)                 # one CPU per pool thread: unpinned, a 1 ms call scattered 50x between repetitions

        def pct(ts_, nb_):
            ts_ = sorted(ts_)
            q = lambda f: nb_ / ts_[min(len(ts_) - 1, int(f * len(ts_)))] / 1e6
            return {"p50": q(0.5), "p10": q(0.9), "p90": q(0.1)}       # (MB/s: the 10th percentile of the RATE is the 90th of the time)

        REPS1, REPS8 = 60, 15

        def run_cfg(th, memo):
            """Both legs of one configuration, INTERLEAVED: REPS8 rounds of (REPS1 / REPS8 calls on one batch, one call on the whole rotation)."""
            orc_t = COracle("cl100k_base", memo=memo)
            for _ in range(3):
                orc_t.encode_packed(one_np, one_off, threads=th)
            orc_t.encode_packed(all_np, all_off, threads=th)
            t_one, t_all = [], []
            for _ in range(REPS8):
                for _ in range(REPS1 // REPS8):
                    c0 = time.perf_counter()
                    orc_t.encode_packed(one_np, one_off, threads=th)
                    t_one.append(time.perf_counter() - c0)
                c0 = time.perf_counter()
                orc_t.encode_packed(all_np, all_off, threads=th)
                t_all.append(time.perf_counter() - c0)
            return pct(t_one, int(one_off[-1])), pct(t_all, int(all_off[-1]))

        t1, t8, unstable = {}, {}, []
        cfgs = [(th, memo) for memo in (False, True) for th in cands if not (memo and th > 24)]   # (the mutex-guarded memo only loses ground with more threads)
        for cfg_ in cfgs:
            t1[cfg_], t8[cfg_] = run_cfg(*cfg_)
        # a configuration whose median is below a third of a neighbour's (the next thread count down or up, same memo) is a scheduling
        # accident, not a property of the port: measured once more, and listed
        for tb, leg in ((t1, "one_batch"), (t8, "rotation_as_one_call")):
            for (th, memo) in cfgs:
                nb = [tb[(t_, memo)]["p50"] for t_ in cands if (t_, memo) in tb and abs(cands.index(t_) - cands.index(th)) == 1]
                if nb and tb[(th, memo)]["p50"] < max(nb) / 3.0:
                    first = tb[(th, memo)]["p50"]
                    r1, r8 = run_cfg(th, memo)
                    again = (r1 if tb is t1 else r8)["p50"]
                    if again > first:
                        t1[(th, memo)], t8[(th, memo)] = r1, r8
                    unstable.append({"config": f"{th}t{'+memo' if memo else ''}", "leg": leg, "first_MBps": round(first, 1), "rerun_MBps": round(again, 1)})

        def surface(texts_, th, memo):
            """The port at the surface the reference's published numbers are quoted on (benchmarks/benchmark_batch.py:45-83:
            3 warm-ups, 10 timed calls of encode_batch(list[str]) -> list[list[int]], mean): the product's own shim halves
            (UTF-8 packing, list building) around the oracle's batch call."""
            orc_t = COracle("cl100k_base", memo=memo)

            def call():
                b_, o_ = shim.pack_bytes(texts_)
                ids_, off_ = orc_t.encode_packed(np.frombuffer(b_, dtype=np.uint8), np.frombuffer(o_, dtype=np.uint64), threads=th)
                return shim.lists_from_csr(ids_.tobytes(), off_.tobytes())
            for _ in range(3):
                call()
            ts_ = []
            for _ in range(10):
                c0 = time.perf_counter()
                r_ = call()
                ts_.append(time.perf_counter() - c0)
                del r_
            nb_ = sum(len(t.encode("utf-8")) for t in texts_)
            return {"mean": round(nb_ / (sum(ts_) / len(ts_)) / 1e6, 1), **{k_: round(v_, 1) for k_, v_ in pct(ts_, nb_).items()}}
        (bth, bmemo), bv = max(t1.items(), key=lambda kv: kv[1]["p50"])
        (b8th, b8memo), b8v = max(t8.items(), key=lambda kv: kv[1]["p50"])
        # `value`: the BEST median over every configuration AND both legs (the GPU is compared with the strongest form of the port on this host)
        best_leg = "one_batch" if bv["p50"] >= b8v["p50"] else "rotation_as_one_call"
        best_v, best_th = (bv, bth) if best_leg == "one_batch" else (b8v, b8th)
        t24 = max((t1[(24, m)]["p50"] for m in (False, True) if (24, m) in t1), default=None)
        fmt = lambda tb: {f"{th}t{'+memo' if m else ''}": round(v["p50"], 1) for (th, m), v in sorted(tb.items())}
        py_c2 = surface(text_sets[0], bth, bmemo)
        # BASELINE config 1 (the reference's own CPU-runnable case): 1000 short English texts, CSR level and Python surface
        c1_texts = corpus.c1(1000)
        c1_np, c1_off = _packed(c1_texts)
        orc_c1 = COracle("cl100k_base", memo=bmemo)
        for _ in range(3):
            orc_c1.encode_packed(c1_np, c1_off, threads=bth)
        ts_c1 = []
        for _ in range(31):
            c0 = time.perf_counter()
            orc_c1.encode_packed(c1_np, c1_off, threads=bth)
            ts_c1.append(time.perf_counter() - c0)
        c1_csr = pct(ts_c1, int(c1_off[-1]))
        py_c1 = surface(c1_texts, bth, bmemo)
        pool_pin(False)
        pool_set_cpus([])
        gpu_py = (throughputs or {}).get("c2", {}).get("python_surface")
        gpu_host = (throughputs or {}).get("c2", {}).get("c_abi_host")
        cpu = {"value": round(best_v["p50"], 2), "unit": "MB/s", "cores": best_th, "kind": "port", "host_cpus": ncpu,
               "value_leg": best_leg,
               "p10_p50_p90": [round(best_v["p10"], 2), round(best_v["p50"], 2), round(best_v["p90"], 2)],
               "one_batch": {"value": round(bv["p50"], 2), "cores": bth, "memo": bmemo},
               "threads_pinned": True, "pool_cpus": len(cpus), "unstable": unstable,
               "gpu_over_cpu": {"kernel_only": round(value / best_v["p50"], 2),
                                "c_abi_host": round(gpu_host / best_v["p50"], 2) if gpu_host else None,
                                "python_surface": round(gpu_py / py_c2["p50"], 2) if gpu_py else None,
                                "note": "kernel_only: `value` of this line / cpu_baseline.value; c_abi_host: throughputs.c2.c_abi_host (host bytes -> host CSR, PCIe "
                                        "included) / cpu_baseline.value (CSR in, CSR out); python_surface: list[str] -> list[list[int]] on both sides"},
               "threads_24": round(t24, 2) if t24 is not None else None,
               "python_surface": {"value": py_c2["p50"], "unit": "MB/s", "stats": py_c2, "cores": bth,
                                  "recipe": "list[str] -> list[list[int]]: shim.pack_bytes + the port's batch call + shim.lists_from_csr; 3 warm-ups, 10 timed calls as "
                                            "benchmarks/benchmark_batch.py:45-83; `value` is the MEDIAN call (that script reports the mean, kept in `stats`: on the 256-CPU host "
                                            "one or two calls of ten stall for milliseconds and drag it down)",
                                  "gpu_python_surface": gpu_py,
                                  "gpu_over_cpu": round(gpu_py / py_c2["p50"], 2) if gpu_py else None},
               "c1": {"workload": "BASELINE config 1: cl100k_base, 1000 short English texts (splintr_amd.corpus.c1, seed 1001)",
                      "bytes": int(c1_off[-1]), "csr_level": {k_: round(v_, 1) for k_, v_ in c1_csr.items()}, "python_surface": py_c1,
                      "cores": bth, "memo": bmemo},
               "rotation_as_one_call": {"value": round(b8v["p50"], 2), "cores": b8th, "memo": b8memo, "docs": len(all_texts),
                                        "bytes": int(all_off[-1]), "repetitions": REPS8, "all_medians": fmt(t8)},
               "all_medians": fmt(t1),
               "sample": f"two legs per configuration, interleaved: the first batch of the rotation ({batches[0].n_docs} docs, {batches[0].n_bytes} B) per call "
                         f"({REPS1} timed calls) and the whole rotation as ONE call ({len(all_texts)} docs, {REPS8} timed calls), 3 warm-ups, MEDIANS; `value` = the best "
                         f"median over all configurations and both legs ({best_leg}); pool threads pinned one per CPU on the {len(cpus)} CPUs of this process's mask "
                         f"that were idle and not used by its other threads; a configuration below a third of its neighbour is measured once more (`unstable`); "
                         f"threads in {cands} x memo on/off -- best: {best_th} threads "
                         f"{'with' if bmemo else 'without'} the mutex-guarded 4096-entry memo that stands in for the "
                         f"reference's LRU; persistent pool pulling documents off a shared counter, CSR in/out"}

    # ---- tripwire: every rate of this line against the last committed line of an earlier round (profiles/rNN_bench.json) ------
    # (two surfaces had become 9 % and 37 % slower without anybody noticing.)  vs_prev = this / previous for rates, previous /
    # this for latencies; anything at or below 0.95 (0.90 for host <-> device figures, 0.75 for the figures that are the host cores') is listed in `regressions`.
    vs_prev = regressions = prev_name = None
    if rank == 0 and world == 1 and not use_dist:
        import glob
        import re
        cands_ = sorted((int(re.search(r"r(\d+)_bench\.json$", f).group(1)), f) for f in glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_bench.json")))
        prev = None
        for _, f in reversed(cands_):
            try:
                prev = json.load(open(f)); prev_name = "profiles/" + os.path.basename(f)
                break
            except Exception:
                prev = None
        if prev:
            vs_prev, regressions = {}, []

            # (figures that are the HOST cores' -- the host splitter, list building in CPython -- move by 20 % from one GPU box to the next:
            #  1 106 / 1 533 / 1 597 / 1 283 MB/s for split_host in four runs of one build; their threshold is 0.75)
            host_bound = ("split_host", "c_abi_host_host_split", "python_surface")

            def cmp_(name, now, before, lower_is_better=False):
                if not isinstance(now, (int, float)) or not isinstance(before, (int, float)) or not now or not before:
                    return
                r = (before / now) if lower_is_better else (now / before)
                vs_prev[name] = round(r, 3)
                # (what crosses PCIe or is a single call's latency moves by +-4 % from box to box -- 27.9 .. 30.2 us for the 1 KB call, 11.7 .. 12.4 GB/s
                #  host to host for one build: 0.90)
                leaf = name.rsplit(".", 1)[-1]
                limit = 0.75 if leaf in host_bound else 0.90 if leaf in ("encode_one_call_us", "c_abi_host", "c_abi_host_pageable", "decode_host") else 0.95
                if r <= limit:
                    regressions.append({"what": name, "now": now, "previous": before, "ratio": round(r, 3), "limit": limit})
            cmp_("value", value, prev.get("value"))
            cmp_("c2_wide_rotation", (c2_wide_rot or {}).get("value"), (prev.get("c2_wide_rotation") or {}).get("value"))
            cmp_("pipelined", (pipelined or {}).get("value"), (prev.get("pipelined") or {}).get("value"))
            cmp_("c4_strong", (c4 or {}).get("value"), (prev.get("c4_strong") or {}).get("value"))
            cmp_("c5_strong", (c5 or {}).get("value"), (prev.get("c5_strong") or {}).get("value"))
            for cfg, ent in (throughputs or {}).items():
                pent = (prev.get("throughputs") or {}).get(cfg)
                if not isinstance(ent, dict) or not isinstance(pent, dict):
                    continue
                for k_, v_ in ent.items():
                    if k_ in ("docs", "bytes", "tokens", "host_threads", "encode_one_call_bytes", "split_device_status", "device_split_fallbacks") or isinstance(v_, bool):
                        continue
                    cmp_(f"throughputs.{cfg}.{k_}", v_, pent.get(k_), lower_is_better=k_.endswith("_us"))

    out = None
    if rank == 0:
        out = {
            "metric": "encode_batch MB/s (bytes in)", "value": round(value, 2), "unit": "MB/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 5),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8",
            "data": "synthetic",
            "config": {"workload": f"cl100k_base, {args.docs} x ~1 KB mixed English/code per batch and GPU, {N_ROT} distinct batches in rotation "
                                   f"(splintr_amd.corpus.{args.corpus}, seeds {seed0_} + 100 rank + k{'' if args.corpus == 'c2' else '; NOT the BASELINE corpus: the lexically wide variant'}); KERNEL-ONLY: corpus HBM-resident, CSR left in HBM "
                                   f"(the host->host and Python-surface rates of the same batch are in `throughputs`); chunk memo warm (`memo.value_memo_off`: the same rotation without it)"
                                   + (f"; the same mix over a >= 20 000-word lexicon (c2_wide, in rotation, same timing): {c2_wide_rot['value']} MB/s" if c2_wide_rot else "")
                                   + (f"; + RCCL all-gatherv of the ragged ids (slab written by the encoder's last kernel, ONE exchange per bucket of {gv.depth} batches on its own stream -- {'ncclAllGather' if gv.collective == 'allgather' else 'grouped ncclSend / ncclRecv'}, chosen by the start-up calibration in `dist` --, overlapped with the following encodes; every rank gets every batch's global CSR, handed to the consumer per bucket)" if use_dist else ""),
                       "vocab": "cl100k_base", "docs_per_batch": args.docs, "bytes_per_batch": round(bytes_rot / N_ROT),
                       "tokens_per_batch": round(sum(n_tokens) / N_ROT), "distinct_batches": N_ROT,
                       "parallelism": f"doc-shard x{world}"},
            "parity": "bit-exact vs oracle (untimed verification pass over every batch of the rotation)",
            "roofline": roofline, "roofline_valu": roofline_valu, "throughputs": throughputs, "c4_strong": c4, "c5_strong": c5,
            "cpu_baseline": cpu, "pipelined": pipelined, "dist": dist_info, "timing": region_stats, "c2_wide_rotation": c2_wide_rot,
            "memo": memo_info, "vs_prev": vs_prev, "vs_prev_source": prev_name, "regressions": regressions,
        }
        if rehearsal:
            out["rehearsal"] = True
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    # the JSON line is the LAST thing on stdout: whatever the C runtime still buffers (RCCL prints a
    # version banner through stdio) goes out first
    try:
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    if rank == 0:
        print(json.dumps(out), flush=True)


C5_PIECES = 16          # distributed runs: every C5 document is cut into this many pieces at context-free boundaries (128 KiB each)
# Distributed strong-scaling legs: the batch is exchanged in N_WAVES TAPERED waves (splintr_amd.distributed.wave_fractions: wave k gets
# taper ** k of the batch).  What the taper should be depends on X / E -- the time the links need for the whole result over the time the
# rank needs to encode its share: a wave's exchange hides behind the NEXT wave's encode only if that one is not shorter, so the waves may
# shrink by at most X / E from one to the next, and the last wave -- whose exchange nothing hides -- is then as small as it can be
# (DESIGN.md section 6 has the arithmetic; tools/dev/wave_taper.py the encode side measured at a rank's slice sizes).  With round 6's
# encoder (a rank's eighth of config 4 in 0.59 - 0.63 ms) and 300 GB/s of all-gather bandwidth X / E is 0.85 - 0.9 with ids packed three
# bytes each: 8 waves at 0.85 are 20.6 / 17.5 / 14.9 / 12.7 / 10.8 / 9.1 / 7.8 / 6.6 % -- the exposed exchange half of what 8 equal waves
# expose, and no wave's exchange waits for the one before it.  Faster links want a smaller taper: SPL_BENCH_WAVE_TAPER / SPL_BENCH_WAVES.
N_WAVES = int(os.environ.get("SPL_BENCH_WAVES", "8"))
WAVE_TAPER = float(os.environ.get("SPL_BENCH_WAVE_TAPER", "0.85"))


def wave_count_bounds(n_items, n_waves=None, taper=None):
    """n_items equal-sized items (prompts, pieces) in their order as tapered waves: n_waves + 1 item indices"""
    from splintr_amd.distributed import wave_fractions
    fr = wave_fractions(N_WAVES if n_waves is None else n_waves, WAVE_TAPER if taper is None else taper)
    b, acc = [0], 0.0
    for f in fr[:-1]:
        acc += f
        b.append(min(max(int(round(n_items * acc)), b[-1]), n_items))
    b.append(n_items)
    return b


def _stream_picks():
    """every stream this process picked by measurement (spl_pick_stream) and what the kept candidate still lost to its neighbours: a leg whose
    `least_bad_pick` is true ran with an exchange or encode stream that shares a hardware queue or pipe -- its figure is suspect (a bad pick
    costs up to 2 x, profiles/r05_wave_exchange.txt section 3)"""
    from splintr_amd.device import stream_picks
    return stream_picks()


def _dist_run(world):
    return world > 1 or os.environ.get("SPL_BENCH_FORCE_DIST") == "1"


def c4_wave_slices(rank, world):
    """[(first prompt, one past the last)] of this rank's slice of every wave, over the C4_PARTS * C4_PART_DOCS prompts in their order"""
    b = wave_count_bounds(C4_PARTS * C4_PART_DOCS)
    return [(b[k] + (b[k + 1] - b[k]) * rank // world, b[k] + (b[k + 1] - b[k]) * (rank + 1) // world) for k in range(len(b) - 1)]


def _c4_slice(args_):
    """what this rank holds of part k: for every wave, the prompts of its slice that lie in the part"""
    k, rank, world = args_
    part = _c4_part(k)
    lo_p = k * C4_PART_DOCS
    return [part[max(a, lo_p) - lo_p:max(min(b_, lo_p + C4_PART_DOCS), lo_p) - lo_p] if b_ > lo_p and a < lo_p + C4_PART_DOCS else []
            for a, b_ in c4_wave_slices(rank, world)]


def gen_c4_texts(rank, world):
    """BASELINE config 4 (1 000 000 prompts = 8 parts of 125 000, seeds 1004 + part; document order = part order).  One GPU: the whole
    batch.  Distributed: the batch is exchanged in N_WAVES tapered WAVES over the prompts in their order (splintr_amd.distributed.plan_waves'
    layout by prompt counts: the prompts are of one size distribution): this rank's contiguous slice of EVERY wave, as a list of lists."""
    from multiprocessing import Pool
    procs = max(1, min(C4_PARTS, (os.cpu_count() or 1) // max(world, 1)))
    if not _dist_run(world):
        with Pool(procs) as pool:
            return [t for part in pool.map(_c4_part, range(C4_PARTS)) for t in part]
    with Pool(procs) as pool:
        per_part = pool.map(_c4_slice, [(k, rank, world) for k in range(C4_PARTS)])       # [part][wave] -> prompts
    return [[t for k in range(C4_PARTS) for t in per_part[k][w]] for w in range(len(per_part[0]))]


def _c5_cut(doc_bytes, frac_num, frac_den):
    """Byte offset of a cut inside a document: the first position at or behind the proportional target that follows a newline
    and holds an ASCII letter or digit -- a context-free match boundary of every built-in split pattern
    (splintr_amd.distributed.plan_shards, spl_api.hip encode_host use the same rule)."""
    i = len(doc_bytes) * frac_num // frac_den
    while True:
        j = doc_bytes.find(b"\n", max(i - 1, 0))
        if j < 0 or j + 1 >= len(doc_bytes):
            return len(doc_bytes)
        c = doc_bytes[j + 1]
        if (48 <= c <= 57) or (65 <= c <= 90) or (97 <= c <= 122):
            return j + 1
        i = j + 2


def _c5_doc_pieces(d):
    """document d of config 5 as C5_PIECES pieces whose ids concatenate to the document's (cuts at context-free boundaries)"""
    raw = _c5_doc(d).encode("utf-8")
    cuts = [0] + [_c5_cut(raw, j, C5_PIECES) for j in range(1, C5_PIECES)] + [len(raw)]
    cuts = sorted(set(cuts))
    out = [raw[a:b].decode("utf-8") for a, b in zip(cuts, cuts[1:]) if b > a]      # (cuts sit in front of ASCII bytes)
    while len(out) < C5_PIECES:
        out.append("")                                  # (a document without enough boundaries: empty pieces keep the counts aligned)
    return out


def c5_wave_slices(n_docs, world, n_waves, pieces=C5_PIECES, taper=None):
    """[(first piece, one past the last)] per wave and rank over the n_docs * pieces pieces in document order: TAPERED waves by piece
    counts (wave_count_bounds), every wave cut into `world` slices of equal piece counts (the pieces are of about equal size)."""
    b = wave_count_bounds(n_docs * pieces, n_waves, taper)
    out = []
    for k in range(n_waves):
        lo, hi = b[k], b[k + 1]
        out.append([(lo + (hi - lo) * r // world, lo + (hi - lo) * (r + 1) // world) for r in range(world)])
    return out


def gen_c5_pieces(rank, world):
    """BASELINE config 5 (deepseek_v3, 100 documents of 2 MiB, seeds 1005 + d).  One GPU: the 100 documents as they are.  Distributed:
    every document is cut into 16 pieces at context-free boundaries (a newline in front of an ASCII letter or digit: the ids of the
    pieces concatenate to the ids of the document -- the intra-document parallelism encode_rayon stands for,
    src/core/tokenizer.rs:815-837), the 1600 pieces in document order are the batch, exchanged in N_WAVES tapered waves; this rank's slice of every
    wave, as a list of lists.  Every rank generates only the documents it holds pieces of."""
    from multiprocessing import Pool
    procs = max(1, (os.cpu_count() or 1) // max(world, 1))
    if not _dist_run(world):
        with Pool(min(C5_DOCS, procs)) as pool:
            return pool.map(_c5_doc, range(C5_DOCS))
    n_waves = min(N_WAVES, C5_DOCS)
    slices = c5_wave_slices(C5_DOCS, world, n_waves)
    need = sorted({p // C5_PIECES for k in range(n_waves) for p in range(*slices[k][rank])})
    with Pool(max(1, min(len(need), procs))) as pool:
        docs = dict(zip(need, pool.map(_c5_doc_pieces, need)))
    return [[docs[p // C5_PIECES][p % C5_PIECES] for p in range(*slices[k][rank])] for k in range(n_waves)]


def run_c4(args, rank, world, local_rank, dev, use_dist, texts):
    """BASELINE config 4 as stated: llama3, 1 000 000 short chat prompts (8 x 125 000, seeds 1004..1011),
    ONE global batch doc-sharded over the ranks (strong scaling), HBM-resident; with W > 1 the ragged ids are all-gathered over RCCL
    inside the timed step, wave by wave behind the encodes.  Returns the sub-object for the JSON line (rank 0), None elsewhere."""
    return run_strong("llama3", texts, args.c4_steps, rank, world, local_rank, dev, use_dist, 20000,
                      "llama3, {docs} short chat prompts ({bytes} B, {tokens} tokens) as ONE batch doc-sharded over {world} GPU(s), HBM-resident",
                      "bit-exact vs oracle on the first and last 20 000 prompts of every rank's shard")


def run_c5(args, rank, world, local_rank, dev, use_dist, pieces):
    """BASELINE config 5 as stated: deepseek_v3, 100 documents of 2 MiB as ONE batch sharded over the ranks, HBM-resident; distributed:
    documents cut into pieces at context-free boundaries (gen_c5_pieces), the ragged ids all-gathered over RCCL inside the timed
    step, wave by wave behind the encodes."""
    return run_strong("deepseek_v3", pieces, args.c5_steps, rank, world, local_rank, dev, use_dist, 1,
                      "deepseek_v3, 100 x 2 MiB documents as {docs} piece(s) ({bytes} B, {tokens} tokens): ONE batch sharded over {world} GPU(s), "
                      "cut inside a document at a newline + ASCII letter/digit (distributed runs: 16 pieces per document), HBM-resident",
                      "bit-exact vs oracle on the first and last piece of every rank's shard")


def run_strong(vocab, texts, steps, rank, world, local_rank, dev, use_dist, nchk, workload, parity):
    """One GPU: `texts` is the batch, ONE encode per step.  Distributed: `texts` is this rank's slice of every wave (a list of
    lists); a step encodes them in order while an exchange stream all-gathers wave k behind encode k + 1
    (splintr_amd.device.WaveGather): every rank ends the step with the CSR of the whole batch in document order."""
    from splintr_amd import Tokenizer
    from splintr_amd.device import Comm, DeviceBatch, WaveGather, encode_device, reserve, result_csr
    from oracle.coracle import COracle
    tok = Tokenizer.from_pretrained(vocab, device=local_rank)
    waves = texts if use_dist else [texts]
    subs = [DeviceBatch(w, dev) for w in waves]
    reserve(tok, max(b.n_bytes for b in subs), max(b.n_docs for b in subs))
    tok2 = None
    if use_dist:
        # a second handle: WaveGather encodes consecutive waves on two streams in alternation (a rank's slice of a wave is 3.4 MB at 8 ranks:
        # launches of that size one after the other run at 30 GB/s, in alternation at 39 -- tools/dev/wave_overlap.py)
        tok2 = Tokenizer.from_pretrained(vocab, device=local_rank)
        reserve(tok2, max(b.n_bytes for b in subs), max(b.n_docs for b in subs))
    orc = COracle(vocab)
    csr = []
    for bi, (b, w) in enumerate(zip(subs, waves)):
        encode_device(tok, b)
        torch.cuda.synchronize()
        ids, off = result_csr(b)
        csr.append((ids, off))
        if bi not in (0, len(subs) - 1):
            continue
        # parity on a bounded sample (the oracle needs seconds per 100 MB): the first and last `nchk` documents of the first and last wave
        nc = min(nchk, len(w))
        for sl in ((slice(0, nc), slice(len(w) - nc, len(w))) if nc else ()):
            t_np, t_off = _packed(w[sl])
            o_ids, o_off = orc.encode_packed(t_np, t_off, threads=os.cpu_count() or 1)
            a, b_ = int(off[sl.start]), int(off[sl.stop])
            if not (np.array_equal(ids[a:b_], o_ids) and np.array_equal(off[sl.start:sl.stop + 1] - off[sl.start], o_off)):
                raise SystemExit(f"rank {rank}: {vocab} strong-scaling result differs from the oracle")
    n_tok = sum(int(off[-1]) for _, off in csr)
    my_bytes, my_docs = sum(b.n_bytes for b in subs), sum(b.n_docs for b in subs)
    wg = comm = None
    cal = {}
    if use_dist:
        # slab capacity from the largest (rank, wave) slice, result capacity from the totals (one-time, untimed)
        mx = torch.tensor([max(int(off[-1]) for _, off in csr), max(b.n_docs for b in subs)], dtype=torch.int64, device=dev)
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        tot = torch.tensor([n_tok, my_docs], dtype=torch.int64, device=dev)
        dist.all_reduce(tot)
        comm = Comm.from_torch_group(dev)
        K = len(subs)

        def make(form):
            return WaveGather(tok, dev, comm, K, max_docs=int(mx[1].item()), max_tokens=int(int(mx[0].item()) * 1.02) + 64,
                              total_tokens_cap=int(tot[0].item()) + 64, total_docs_cap=int(tot[1].item()), collective=form.split("+")[0],
                              pack24="pack24" in form, tok2=tok2 if "+2s" in form else None, enc_pair=int(form.split("@")[1]) if "@" in form else 0)

        def step_with(g_):
            g_.begin()
            for b in subs:
                g_.encode_and_submit(b)
            return g_.finish()
        # calibration of the collective form (untimed): ncclAllGather of the wave's slabs against grouped send / recv
        # ... and one or two encode streams ("+2s": consecutive waves on two handles in alternation -- fewer idle CUs between the launches of
        # small waves, but the exchange then has no idle CUs to hide in)
        # (two streams that land on one pipe of the command processor take turns: which do is not knowable, so three pairs are tried)
        for form in ("allgather", "p2p", "allgather+pack24") + tuple(f"{f_}+2s@{p_}" for p_ in range(3) for f_ in ("allgather", "allgather+pack24")):
            g_ = make(form)
            for _ in range(2):
                step_with(g_)
            torch.cuda.synchronize()
            dist.barrier()
            c0 = time.perf_counter()
            for _ in range(steps):                                   # as the timed region: `steps` steps back to back
                step_with(g_)
            torch.cuda.synchronize()
            ct = torch.tensor([time.perf_counter() - c0], dtype=torch.float64, device=dev)
            dist.all_reduce(ct, op=dist.ReduceOp.MAX)
            cal[form] = float(ct.item()) / steps * 1e3
            del g_
            torch.cuda.empty_cache()
        wg = make(min(cal, key=cal.get))

    def step():
        if wg is None:
            encode_device(tok, subs[0])
        else:
            wg.begin()
            for b in subs:
                wg.encode_and_submit(b)
            wg.finish()
    for _ in range(2):
        step()
    if True:
        # "memo warm" means what it says: a cold pass over 200 MB logs more missed chunks than one fill takes (65 536 a fill, duplicates among
        # them), so the fills of the first passes run on until every chunk is in -- steps until two in a row have run without a fill (12 at most)
        import ctypes as _ct
        from splintr_amd import _ffi as _f1
        _L1 = _f1.lib()
        _L1.spl_memo_stats.argtypes = [_ct.c_void_p, _ct.POINTER(_ct.c_uint64)]
        _ms, _prev, _quiet = (_ct.c_uint64 * 4)(), -1, 0
        for _ in range(12):
            step()
            torch.cuda.synchronize()
            _fills = 0
            for _t in (tok, tok2):                      # (the distributed form: two handles, consecutive waves in alternation -- a memo each)
                if _t is not None:
                    _L1.spl_memo_stats(_t.handle, _ms)
                    _fills += int(_ms[0])
            _quiet = _quiet + 1 if _fills == _prev else 0
            _prev = _fills
            _done = 1 if _quiet >= 2 else 0
            if use_dist:                                # (every rank runs the same number of steps: a step holds a collective)
                _dv = torch.tensor([_done], dtype=torch.int32, device=dev)
                dist.all_reduce(_dv, op=dist.ReduceOp.MIN)
                _done = int(_dv.item())
            if _done:
                break
    if use_dist:
        # every rank holds the whole result: totals, and this rank's slice of every wave at its place
        torch.cuda.synchronize()
        assert not wg.overflowed()
        run = wg.run.cpu().numpy()
        assert (int(run[0]), int(run[1])) == (int(tot[0].item()), int(tot[1].item())), (run, tot)
        mine = torch.tensor([[int(off[-1]), b.n_docs] for (_, off), b in zip(csr, subs)], dtype=torch.int64, device=dev)
        allc = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allc, mine)
        allc = torch.stack(allc).cpu().numpy()                       # [rank, wave, {T, N}]
        g_ids, g_off = wg.all_ids.cpu().numpy().view(np.uint32), wg.all_off.cpu().numpy()
        for k, ((ids, off), b) in enumerate(zip(csr, subs)):
            t0_ = int(allc[:, :k, 0].sum() + allc[:rank, k, 0].sum())
            d0_ = int(allc[:, :k, 1].sum() + allc[:rank, k, 1].sum())
            assert np.array_equal(g_ids[t0_:t0_ + int(off[-1])], ids), (rank, k)
            assert np.array_equal((g_off[d0_:d0_ + b.n_docs + 1] - t0_).astype(np.uint64), off), (rank, k)
        for _ in range(2):                                           # the check above left the GPU idle for a while: warm up again
            step()
        torch.cuda.synchronize()
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    t_issue = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_sync = time.perf_counter() - t0
    if use_dist:
        dist.barrier()
    el = time.perf_counter() - t0
    if os.environ.get("SPL_BENCH_DEBUG"):
        print(f"[debug] {vocab}: issue {t_issue * 1e3:.3f} ms, synchronised {t_sync * 1e3:.3f} ms, after the barrier {el * 1e3:.3f} ms", file=sys.stderr)
    tot_b, tot_d, tot_t = my_bytes, my_docs, n_tok
    dist_info = None
    memo_off_ms = None
    if not use_dist:
        # the same steps with the chunk memo OFF: every step of this leg re-encodes the SAME batch, so the memo is warm from the verification
        # pass on (as the reference's LRU is in its own benchmark, which times repeated calls on one batch); this is what a batch costs cold
        from splintr_amd import _ffi as _f2
        _f2.lib().spl_set_option(tok.handle, b"memo", 0)
        step()
        torch.cuda.synchronize()
        m0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        memo_off_ms = (time.perf_counter() - m0) / steps * 1e3
        _f2.lib().spl_set_option(tok.handle, b"memo", 1)
    if use_dist:
        local_ms = el / steps * 1e3
        # untimed repeats: the encodes alone, and what the exchange stream spent in collective + unpack (events around them)
        from splintr_amd.device import encode_streams
        es_ = encode_streams(dev, wg.enc_pair)
        e0 = time.perf_counter()
        for _ in range(steps):
            for k_, b in enumerate(subs):                          # as WaveGather encodes them: on one stream, or two handles and streams in alternation
                if wg.enc is None:
                    encode_device(tok, b)
                else:
                    with torch.cuda.stream(es_[k_ & 1]):
                        encode_device((tok, tok2)[k_ & 1], b)
        torch.cuda.synchronize()
        enc_ms = (time.perf_counter() - e0) / steps * 1e3
        dist.barrier()
        wg.enable_timing()
        for _ in range(steps):
            step()
        ex_total, _n = wg.exchange_ms()
        wg.enable_timing(False)
        mine = torch.tensor([local_ms, enc_ms, ex_total / steps], dtype=torch.float64, device=dev)
        allm = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allm, mine)
        allm = torch.stack(allm).cpu().numpy()
        v = torch.tensor([el], dtype=torch.float64, device=dev)
        dist.all_reduce(v, op=dist.ReduceOp.MAX)
        el = float(v.item())
        s_ = torch.tensor([my_bytes, my_docs, n_tok], dtype=torch.int64, device=dev)
        dist.all_reduce(s_)
        tot_b, tot_d, tot_t = (int(x) for x in s_.tolist())
        step_ms, exposed = float(allm[:, 0].max()), float(allm[:, 0].max() - allm[:, 1].max())
        dist_info = {"rccl_ranks": comm.ranks(), "torch_world": world, "waves": len(subs), "collective": wg.collective, "pack24": wg.pack24,
                     "per_rank": {"step_ms": [round(float(x), 4) for x in allm[:, 0]],
                                  "encode_only_ms": [round(float(x), 4) for x in allm[:, 1]],
                                  "exchange_stream_ms": [round(float(x), 4) for x in allm[:, 2]]},
                     "exposed_exchange_ms": round(exposed, 4), "exposed_over_step": round(exposed / step_ms, 4) if step_ms else None,
                     "calibration_ms_per_step": {k_: round(v_, 4) for k_, v_ in cal.items()},
                     "slab_bytes_sent_per_wave": wg.cap_words * 4, "bytes_received_per_rank": wg.cap_words * 4 * world * len(subs),
                     "ids_bytes_4T": 4 * tot_t,
                     "encode_streams": 1 if wg.enc is None else 2,
                     "stream_picks": _stream_picks(), "least_bad_pick": any(p_["least_bad"] for p_ in _stream_picks()),
                     "note": "WaveGather: the batch is exchanged in `waves` waves; rank r encodes its slice of wave k into a slab (written by the encoder's "
                             "last kernel; with encode_streams 2 consecutive waves on two handles and streams in alternation), ONE exchange of the wave's slabs + an unpack behind what the earlier waves left run on an exchange stream while "
                             "wave k + 1 encodes; no host synchronisation inside a step.  exposed = slowest step - slowest encodes-only"}
    del subs, tok, tok2, comm, wg
    torch.cuda.empty_cache()
    if rank != 0:
        return None
    return {"workload": workload.format(docs=tot_d, bytes=tot_b, tokens=tot_t, world=world)
                        + (f"; RCCL all-gatherv of the ragged result inside the step, pipelined: {dist_info['waves']} waves, wave k on the links while wave k + 1 encodes "
                           f"(splintr_amd.device.WaveGather over spl_allgather_slabs{'_p2p' if dist_info['collective'] == 'p2p' else ''} + spl_gatherv_unpack_at)" if use_dist else ""),
            "value": round(tot_b * steps / el / 1e6, 1), "unit": "MB/s", "ms_per_step": round(el / steps * 1e3, 3),
            "steps": steps, "scaling": "strong", "parity": parity, "dist": dist_info,
            "memo": None if memo_off_ms is None else {"value_memo_off": round(tot_b / (memo_off_ms * 1e-3) / 1e6, 1), "ms_per_step_memo_off": round(memo_off_ms, 3),
                                                       "note": "`value` is with the chunk memo WARM (every step re-encodes the same batch); value_memo_off: the same steps with "
                                                               "spl_set_option(memo, 0) -- what the batch costs cold"}}


if __name__ == "__main__":
    main()
