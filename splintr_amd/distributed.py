"""Document sharding across the GPUs of one node and the ragged all-gather of the result.

The reference has nothing distributed (Rayon on one host, src/core/tokenizer.rs:932-934);
documents are independent, so the only exchange is reassembly: one process per GPU encodes a
contiguous, byte-balanced range of documents, then an all-gatherv over RCCL/xGMI gives every
rank the whole CSR result.  RCCL has no all-gatherv: it is composed from an all-gather of the
counts and an all-gather of slices padded to the largest count.  The functions work on whatever
backend the default process group uses (nccl = RCCL on GPU tensors, gloo on CPU tensors), which
is how the CPU tests cover this logic.
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import numpy as np
import torch
import torch.distributed as dist


def shard_bounds(doc_bytes: Sequence[int], world: int) -> List[int]:
    """Cut N documents into `world` contiguous ranges of about equal bytes.
    Returns world+1 document indices; rank r owns [b[r], b[r+1])."""
    n = len(doc_bytes)
    csum = np.concatenate([[0], np.cumsum(np.asarray(doc_bytes, dtype=np.int64))])
    total = int(csum[-1])
    bounds = [0]
    for r in range(1, world):
        target = total * r / world
        i = int(np.searchsorted(csum, target, side="left"))
        # nearest document boundary, monotone
        if i > 0 and abs(csum[i - 1] - target) <= abs(csum[min(i, n)] - target):
            i -= 1
        bounds.append(min(max(i, bounds[-1]), n))
    bounds.append(n)
    return bounds


def wave_fractions(n_waves: int, taper: float = 1.0) -> List[float]:
    """Shares of a batch for `n_waves` waves: wave k gets taper ** k (normalised).  taper = 1: equal waves.  taper < 1: TAPERED -- the early
    waves are large (a rank's slice of them is a launch big enough to run at the tile kernel's full rate, and their exchange hides behind
    the encodes that follow), the last wave -- whose exchange nothing hides -- is small: 7 waves at taper 0.7 are 33 / 23 / 16 / 11 / 8 /
    5.5 / 3.8 % of the batch, where 8 equal waves expose 12.5 %."""
    if n_waves < 1 or not (0.0 < taper <= 1.0):
        raise ValueError("wave_fractions: n_waves >= 1, 0 < taper <= 1")
    w = [taper ** k for k in range(n_waves)]
    t = sum(w)
    return [x / t for x in w]


def fraction_bounds(doc_bytes: Sequence[int], fractions: Sequence[float]) -> List[int]:
    """Cut N documents into len(fractions) contiguous ranges whose bytes are about the given shares.  Returns len + 1 document indices."""
    n = len(doc_bytes)
    csum = np.concatenate([[0], np.cumsum(np.asarray(doc_bytes, dtype=np.int64))])
    total = int(csum[-1])
    tot_f = float(sum(fractions))
    bounds, acc = [0], 0.0
    for f in list(fractions)[:-1]:
        acc += f / tot_f
        target = total * acc
        i = int(np.searchsorted(csum, target, side="left"))
        if i > 0 and abs(csum[i - 1] - target) <= abs(csum[min(i, n)] - target):
            i -= 1
        bounds.append(min(max(i, bounds[-1]), n))
    bounds.append(n)
    return bounds


def plan_waves(doc_bytes: Sequence[int], world: int, n_waves: int, taper: float = 1.0,
               fractions: Sequence[float] = None) -> List[List[Tuple[int, int]]]:
    """ONE batch for the pipelined strong-scaling exchange (splintr_amd.device.WaveGather): the documents, in their order, are cut into
    `n_waves` contiguous waves -- of about equal bytes, or TAPERED (`taper` < 1, wave_fractions; or explicit `fractions`) -- and every
    wave into `world` contiguous slices of about equal bytes.  Returns waves[k][r] = (first document, one past the last) of rank r's
    slice of wave k.  The result of the batch is the concatenation over k of (the concatenation over r of slice (k, r)) -- document
    order (what Rayon's order-preserving collect gives the reference, src/core/tokenizer.rs:932-942) -- so wave k can be exchanged, and
    land at its final place, while wave k + 1 is still being encoded: its place depends only on the waves before it."""
    fr = list(fractions) if fractions is not None else wave_fractions(n_waves, taper)
    if len(fr) != n_waves:
        raise ValueError("plan_waves: one fraction per wave")
    wb = fraction_bounds(doc_bytes, fr)
    waves = []
    for k in range(n_waves):
        lo, hi = wb[k], wb[k + 1]
        rb = shard_bounds(doc_bytes[lo:hi], world)
        waves.append([(lo + rb[r], lo + rb[r + 1]) for r in range(world)])
    return waves


def encode_batch_waves(encode_csr, texts: Sequence[str], device: torch.device, n_waves: int = 4, group=None, taper: float = 1.0):
    """encode_batch over the process group in waves (whole documents; the host-tensor form of WaveGather, on whatever backend the
    group uses -- gloo on CPU tensors in the tests): rank r encodes its slice of every wave with `encode_csr`, wave k's ragged
    result is all-gathered and lands behind the waves before it; the offsets rebase per wave.  Returns (ids, off) numpy arrays
    for ALL documents, in their order, on every rank."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    lens = [len(t.encode("utf-8")) for t in texts]
    waves = plan_waves(lens, world, n_waves, taper)
    ids_parts, off_parts, t_done = [], [np.zeros(1, dtype=np.int64)], 0
    for k in range(n_waves):
        lo, hi = waves[k][rank]
        ids, off = encode_csr(list(texts[lo:hi]))
        t_ids = torch.from_numpy(ids.astype(np.int32, copy=False).copy()).to(device)
        dc = torch.from_numpy(np.diff(off.astype(np.int64))).to(device)
        w_ids, w_off = all_gather_csr(t_ids, int(off[-1]), dc, group)
        ids_parts.append(w_ids.cpu().numpy().view(np.uint32))
        off_parts.append(w_off.cpu().numpy().astype(np.int64)[1:] + t_done)          # the wave's offsets, rebased by what came before
        t_done += int(w_off[-1])
    return np.concatenate(ids_parts) if ids_parts else np.zeros(0, np.uint32), np.concatenate(off_parts).astype(np.uint64)


def all_gather_csr(ids: torch.Tensor, n_tokens, doc_counts: torch.Tensor,
                   group=None) -> Tuple[torch.Tensor, torch.Tensor]:
    """Ragged all-gather.  ids[:n_tokens] = this rank's token ids (int32), doc_counts = tokens per
    local document (int64); n_tokens may be an int or a 0-dim device tensor (no host sync needed
    before the exchange).  Every rank returns (all_ids int32[T_total], all_off int64[N_total+1])
    in rank order, i.e. in global document order."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    dev = ids.device
    # 1. counts
    if torch.is_tensor(n_tokens):
        mine = torch.stack([n_tokens.to(torch.int64).reshape(()),
                            torch.tensor(doc_counts.numel(), dtype=torch.int64, device=dev)])
    else:
        mine = torch.tensor([int(n_tokens), doc_counts.numel()], dtype=torch.int64, device=dev)
    allc = [torch.zeros(2, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(allc, mine, group=group)
    counts = torch.stack(allc).cpu().numpy()          # the one host sync of the exchange
    n_tokens = int(counts[rank, 0])
    tmax, dmax = int(counts[:, 0].max()), int(counts[:, 1].max())
    # 2. padded slices (equal sizes: one collective each, every link busy at once over xGMI)
    pad_ids = torch.zeros(max(tmax, 1), dtype=torch.int32, device=dev)
    pad_ids[:n_tokens] = ids[:n_tokens]
    g_ids = [torch.empty_like(pad_ids) for _ in range(world)]
    dist.all_gather(g_ids, pad_ids, group=group)
    pad_dc = torch.zeros(max(dmax, 1), dtype=torch.int64, device=dev)
    pad_dc[: doc_counts.numel()] = doc_counts
    g_dc = [torch.empty_like(pad_dc) for _ in range(world)]
    dist.all_gather(g_dc, pad_dc, group=group)
    # 3. strip the padding, rebase offsets
    all_ids = torch.cat([g_ids[r][: int(counts[r, 0])] for r in range(world)])
    all_dc = torch.cat([g_dc[r][: int(counts[r, 1])] for r in range(world)])
    all_off = torch.zeros(all_dc.numel() + 1, dtype=torch.int64, device=dev)
    torch.cumsum(all_dc, 0, out=all_off[1:])
    return all_ids, all_off


_ALNUM = frozenset(b"0123456789abcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZ")


def plan_shards(docs: Sequence[bytes], world: int, split_docs: bool = True,
                search: int = 1 << 20) -> List[List[Tuple[int, int, int]]]:
    """Cut a batch into `world` contiguous shards of about equal BYTES.  Returns, per rank, a list of
    pieces (doc index, lo, hi): whole documents, except that a shard boundary falling inside a large
    document is moved forward to the next newline that is followed by an ASCII letter or digit -- a
    context-free match boundary of the three BUILT-IN split patterns (cl100k, o200k / llama3 / deepseek_v3,
    mistral_v3: SURVEY 8e; spl_scan.h is_sync rule (d)), so the ids of the two pieces concatenate to the ids of
    the document.  It is NOT one for an arbitrary pattern (`[^\n]+\n*\p{L}+` matches across it): callers whose
    tokenizer has a custom pattern pass split_docs=False (encode_batch_sharded does so by itself; the C host
    path likewise, `may_cut`, spl_api.hip).  This is what lets
    100 equal 2 MiB documents (BASELINE config 5), or ONE huge document (the reference's
    encode_rayon case, src/core/tokenizer.rs:815-837), spread evenly over 8 GPUs."""
    n = len(docs)
    lens = np.fromiter((len(d) for d in docs), dtype=np.int64, count=n)
    csum = np.concatenate([[0], np.cumsum(lens)])
    total = int(csum[-1])
    cuts: List[Tuple[int, int]] = [(0, 0)]                  # (doc, offset inside it); (n, 0) = the end
    for r in range(1, world):
        target = total * r // world
        d = int(np.searchsorted(csum, target, side="right")) - 1       # csum[d] <= target < csum[d + 1]
        d = min(max(d, 0), n - 1) if n else 0
        lo_b, hi_b = int(csum[d]), int(csum[d + 1]) if n else 0
        tol = total // world // 16
        near = (d, 0) if target - lo_b <= hi_b - target else (d + 1, 0)
        dist_b = min(target - lo_b, hi_b - target)
        cut = near
        if split_docs and n and dist_b > tol and lo_b < target < hi_b:
            doc = docs[d]
            i = target - lo_b
            end = min(len(doc), i + search)
            while True:
                j = doc.find(b"\n", i, end)
                if j < 0 or j + 1 >= len(doc):
                    break
                if doc[j + 1] in _ALNUM:
                    cut = (d, j + 1)
                    break
                i = j + 1
        cut = max(cut, cuts[-1])                            # monotone
        cuts.append(cut)
    cuts.append((n, 0))
    shards: List[List[Tuple[int, int, int]]] = []
    for r in range(world):
        (d0, o0), (d1, o1) = cuts[r], cuts[r + 1]
        pieces: List[Tuple[int, int, int]] = []
        for d in range(d0, min(d1 + (1 if o1 else 0), n)):
            lo = o0 if d == d0 else 0
            hi = o1 if (d == d1 and o1) else len(docs[d])
            if d == d1 and not o1:
                break
            pieces.append((d, lo, hi))
        shards.append(pieces)
    return shards


def encode_batch_sharded(encode_csr, texts: Sequence[str], device: torch.device, group=None, split_docs: bool = True,
                         special_literals: Sequence[str] = (), context_free_cuts=None):
    """encode_batch over the process group (strong scaling of ONE batch): rank r encodes its
    byte-balanced shard -- whole documents and, at the shard's ends, pieces of documents cut at
    context-free boundaries (plan_shards) -- with `encode_csr(list[str]) -> (ids uint32 ndarray,
    off uint64 ndarray)`, and the ragged result is all-gathered.  Returns (ids, off) numpy arrays
    for ALL documents on every rank.

    `special_literals`: when `encode_csr` encodes WITH special tokens, pass the literals of its map.  A cut
    sits directly behind a newline, so it can only fall inside a literal that contains one; if any does,
    documents are not cut at all (what spl_encode_batch's host path does: `special_newline`, spl_api.hip).

    `context_free_cuts`: whether "behind a newline, in front of an ASCII letter or digit" is a match boundary of the
    encoder's split pattern whatever surrounds it.  True for the three built-in patterns, unknown for a custom one
    (src/core/tokenizer.rs:410-456 compiles any pattern), so False there: documents then stay whole.  None (default)
    asks the encoder: a bound method of a `Tokenizer` (or any object with `has_custom_pattern`), also behind functools.partial or a
    decorator that sets __wrapped__, answers for itself; for an opaque callable the documents stay whole (pass True to allow cuts)."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    if context_free_cuts is None:
        # (a functools.partial or a decorated function has no __self__ of its own -- unwrap; an encoder whose owner cannot be
        #  found is NOT assumed to use a built-in pattern: its documents stay whole, which is always correct)
        fn = encode_csr
        for _ in range(8):
            if hasattr(fn, "func") and callable(getattr(fn, "func")):
                fn = fn.func
            elif hasattr(fn, "__wrapped__"):
                fn = fn.__wrapped__
            else:
                break
        owner = getattr(fn, "__self__", None)
        context_free_cuts = owner is not None and hasattr(owner, "has_custom_pattern") and not bool(owner.has_custom_pattern)
    if not context_free_cuts or any("\n" in lit for lit in special_literals):
        split_docs = False
    docs = [t.encode("utf-8") for t in texts]
    shards = plan_shards(docs, world, split_docs)
    local = [docs[d][lo:hi].decode("utf-8") for d, lo, hi in shards[rank]]     # cuts sit in front of ASCII bytes
    ids, off = encode_csr(local)
    t_ids = torch.from_numpy(ids.astype(np.int32, copy=False).copy()).to(device)
    dc = torch.from_numpy(np.diff(off.astype(np.int64))).to(device)
    all_ids, all_off = all_gather_csr(t_ids, int(off[-1]), dc, group)
    # one offset entry per PIECE so far; a piece that continues a document contributes none
    keep = np.ones(sum(len(s) for s in shards) + 1, dtype=bool)
    k = 0
    for s in shards:
        for _, lo, _ in s:
            if lo != 0:
                keep[k] = False
            k += 1
    return all_ids.cpu().numpy().view(np.uint32), all_off.cpu().numpy().astype(np.uint64)[keep]
