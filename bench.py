#!/usr/bin/env python3
"""encode_batch throughput on MI355X (BASELINE.json metric: MB/s of input bytes, MB = 1e6 B).

A "step" is ONE pass of the hot path (Tokenizer.encode_batch semantics through the C ABI,
spl_encode_batch_device) over ONE batch of synthetic text already resident in HBM:
config[1] of BASELINE.json = cl100k_base, 1000 x ~1 KB mixed English/code (splintr_amd.corpus.c2).
With --gpus N > 1 every rank holds its own 1000-document shard (weak scaling) and the step also
all-gathers the ragged ids over RCCL so that every rank ends up with the whole CSR result.

    python bench.py --gpus 1 --steps 500 --warmup 50
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Prints one JSON line on rank 0.  The oracle (oracle/) is used only as the checker of the
untimed verification pass and as the timed CPU baseline ("port"); it is never on the measured path.
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E peak (MI355X_MICROARCH.md)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--inflight", type=int, default=1,
                    help="N > 1: also report the steps with N batches in flight on their own handles and streams "
                         "(\"pipelined\"; off by default so that a profile of the default command sees one batch at a time)")
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--docs", type=int, default=1000, help="documents per GPU (BASELINE config: 1000)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched through torch.distributed.run (one rank per GPU)")
        args.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product has no CPU path)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or os.environ.get("SPL_BENCH_FORCE_DIST") == "1"   # the latter: 1-rank smoke of the RCCL path
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    import __graft_entry__ as entry
    if rank == 0:
        entry.build()
    if use_dist:
        dist.barrier()
    from splintr_amd import Tokenizer, corpus, _ffi
    from splintr_amd.device import DeviceBatch, GatherV, encode_device, reserve, result_csr

    tok = Tokenizer.from_pretrained("cl100k_base", device=local_rank)
    texts = corpus.c2(args.docs, seed=1002 + rank)          # rank-distinct shard, same distribution
    batch = DeviceBatch(texts, dev)
    reserve(tok, batch.n_bytes, batch.n_docs)
    L = _ffi.lib()

    gv = None

    def step():
        if gv is None:
            encode_device(tok, batch)
        else:
            # encode with the slab written by the encoder's last kernel; every 8th batch: one RCCL
            # all-gather of the bucket + one unpack launch, on a stream of their own
            gv.encode_and_submit(batch)

    # ---- untimed verification pass: bit-exact vs the oracle on this very batch -----------------
    step()
    torch.cuda.synchronize()
    ids, off = result_csr(batch)
    n_tokens = int(off[-1])
    from oracle.coracle import COracle
    orc = COracle("cl100k_base")
    text_np = np.frombuffer(b"".join(t.encode("utf-8") for t in texts), dtype=np.uint8)
    ncpu = os.cpu_count() or 1
    o_ids, o_off = orc.encode_packed(text_np, batch.host_offsets, threads=ncpu)
    if not (np.array_equal(ids, o_ids) and np.array_equal(off, o_off)):
        raise SystemExit(f"rank {rank}: HIP result differs from the oracle -- refusing to report a throughput")

    if use_dist:
        # size the slabs from the largest shard (one-time, untimed), then check that the exchange
        # reproduces this rank's ids and offsets at its place in the global CSR
        mx = torch.tensor([n_tokens, batch.n_docs], dtype=torch.int64, device=dev)
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        gv = GatherV(tok, dev, max_docs=int(mx[1].item()), max_tokens=int(int(mx[0].item()) * 1.02) + 64)
        step()
        g_ids, g_off = gv.finish()
        torch.cuda.synchronize()
        assert not gv.overflowed()
        cnt = torch.tensor([n_tokens, batch.n_docs], dtype=torch.int64, device=dev)
        allc = [torch.zeros_like(cnt) for _ in range(world)]
        dist.all_gather(allc, cnt)
        allc = torch.stack(allc).cpu().numpy()
        t_before, d_before = int(allc[:rank, 0].sum()), int(allc[:rank, 1].sum())
        assert int(g_off[int(allc[:, 1].sum())].item()) == int(allc[:, 0].sum())
        assert np.array_equal(g_ids[t_before:t_before + n_tokens].cpu().numpy().view(np.uint32), ids)
        assert np.array_equal((g_off[d_before:d_before + batch.n_docs + 1] - t_before).cpu().numpy().astype(np.uint64), off)

    # ---- timed region ----------------------------------------------------------------------------
    for _ in range(args.warmup):
        step()
    if gv is not None:
        gv.finish()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    if gv is not None:
        gv.finish()                   # the last batch's exchange completes inside the timed region
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if use_dist:
        et = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(et, op=dist.ReduceOp.MAX)
        elapsed = float(et.item())
        nb = torch.tensor([batch.n_bytes], dtype=torch.int64, device=dev)
        dist.all_reduce(nb)
        total_bytes = int(nb.item())
    else:
        total_bytes = batch.n_bytes
    ms_per_step = elapsed / args.steps * 1e3
    value = total_bytes * args.steps / elapsed / 1e6

    # ---- supplementary: the same steps with several batches in flight --------------------------------
    # (one handle, workspace and stream each; `value` above stays the one-batch-at-a-time figure that
    #  the per-kernel durations and the roofline below belong to.  Consecutive steps overlap: the
    #  next batch's tile kernel fills the CUs that the stragglers of this one and k_tile_out leave idle)
    pipelined = None
    if rank == 0 and world == 1 and not use_dist and args.inflight > 1:
        toks = [tok] + [Tokenizer.from_pretrained("cl100k_base", device=local_rank) for _ in range(args.inflight - 1)]
        bats = [batch] + [DeviceBatch(texts, dev) for _ in range(args.inflight - 1)]
        strs = [torch.cuda.Stream(dev) for _ in range(args.inflight)]
        for t_, b_ in zip(toks[1:], bats[1:]):
            reserve(t_, b_.n_bytes, b_.n_docs)

        def run(k):
            for i in range(k):
                j = i % args.inflight
                with torch.cuda.stream(strs[j]):
                    encode_device(toks[j], bats[j])
        run(args.warmup + args.inflight)
        torch.cuda.synchronize()
        p0 = time.perf_counter()
        run(args.steps)
        torch.cuda.synchronize()
        pel = time.perf_counter() - p0
        for b_ in bats[1:]:
            i2, o2 = result_csr(b_)
            if not (np.array_equal(i2, o_ids) and np.array_equal(o2, o_off)):
                raise SystemExit("pipelined run: result differs from the oracle")
        pipelined = {"inflight": args.inflight, "value": round(batch.n_bytes * args.steps / pel / 1e6, 2), "unit": "MB/s",
                     "ms_per_step": round(pel / args.steps * 1e3, 5),
                     "note": "same batch and steps, round-robin over handles on their own HIP streams; every handle's result bit-exact"}
        del toks, bats

    # ---- per-kernel durations: HIP events on the launch stream (separate pass) ---------------------
    roofline = None
    kernels = {}
    if rank == 0:
        L.spl_profile_enable(tok.handle, 1)
        L.spl_profile_reset(tok.handle)
        nprof = 100
        for _ in range(nprof):
            encode_device(tok, batch)
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
        # input and output offset arrays (u64 each)
        b_alg = batch.n_bytes + 4 * n_tokens + 16 * (batch.n_docs + 1)
        achieved = b_alg / (kernels[dom] * 1e-6) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get(dom, {}).get("bytes_per_launch")
            except Exception:
                traffic = None
        roofline = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                    "algorithmic_bytes_per_launch": b_alg, "kernel_us": kernels[dom],
                    "all_kernels_us": kernels}

    # ---- CPU baseline: the oracle (a port of the reference's Rayon path) on the host cores ---------
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        # sweep the thread count and the LRU-style memo, keep the best: the GPU is compared with the
        # strongest configuration of the port on this host, not with an arbitrary one
        best = None
        cands = sorted({t for t in (8, 16, 32, 64, 128, ncpu) if t <= ncpu} | {min(8, ncpu)})
        for memo in (False, True):
            orc_t = COracle("cl100k_base", memo=memo)
            for th in cands:
                orc_t.encode_packed(text_np, batch.host_offsets, threads=th)
                reps, c0 = 0, time.perf_counter()
                while time.perf_counter() - c0 < 1.2 or reps < 3:
                    orc_t.encode_packed(text_np, batch.host_offsets, threads=th)
                    reps += 1
                rate = batch.n_bytes * reps / (time.perf_counter() - c0) / 1e6
                if best is None or rate > best[0]:
                    best = (rate, th, memo, reps)
        cpu = {"value": round(best[0], 2), "unit": "MB/s", "cores": best[1], "kind": "port",
               "host_cpus": ncpu,
               "sample": f"the full bench batch ({batch.n_docs} docs, {batch.n_bytes} B) x {best[3]} repetitions; "
                         f"best of threads in {cands} x memo on/off (best: {best[1]} threads, "
                         f"{'with' if best[2] else 'without'} the mutex-guarded 4096-entry memo that stands in for the "
                         f"reference's LRU); persistent pool pulling documents off a shared counter, CSR in/out"}

    if rank == 0:
        out = {
            "metric": "encode_batch MB/s (bytes in)", "value": round(value, 2), "unit": "MB/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 5),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8",
            "data": "synthetic",
            "config": {"workload": f"cl100k_base, {args.docs} x ~1 KB mixed English/code per GPU "
                                   f"(splintr_amd.corpus.c2, seed 1002+rank), HBM-resident, CSR out"
                                   + ("; + RCCL all-gatherv of the ragged ids (slabs packed per batch, ONE all-gather per bucket of 8 batches on its own stream, overlapped with the following encodes; every rank ends with every batch's global CSR)" if use_dist else ""),
                       "vocab": "cl100k_base", "docs_per_gpu": args.docs, "bytes_per_gpu": batch.n_bytes,
                       "tokens_per_gpu": n_tokens, "parallelism": f"doc-shard x{world}"},
            "parity": "bit-exact vs oracle (untimed verification pass on the bench batch)",
            "roofline": roofline, "cpu_baseline": cpu, "pipelined": pipelined,
        }
        print(json.dumps(out))
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
