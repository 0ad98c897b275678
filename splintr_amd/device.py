"""Device-resident batch encode: torch tensors in HBM in, CSR tensors in HBM out.

PyTorch is plumbing here (device memory, streams, torch.distributed); the work is done by
`spl_encode_batch_device` (include/splintr_hip.h) on torch's current HIP stream.
"""
from __future__ import annotations

import ctypes
from typing import Optional, Sequence, Tuple

import numpy as np
import torch

from . import _ffi
from .tokenizer import Tokenizer, _pack


class DeviceBatch:
    """A packed corpus resident on one GPU plus preallocated output buffers."""

    def __init__(self, texts: Sequence[str], device: torch.device):
        buf, off = _pack(texts)
        self.n_docs = len(off) - 1
        self.n_bytes = len(buf)
        pad = (-self.n_bytes) % 16 + 16
        host = np.frombuffer(buf + b"\0" * pad, dtype=np.uint8)
        self.text = torch.from_numpy(host.copy()).to(device)
        self.doc_off = torch.from_numpy(off.astype(np.int64)).to(device)
        self.ids = torch.empty(max(self.n_bytes, 1), dtype=torch.int32, device=device)
        self.out_off = torch.zeros(self.n_docs + 1, dtype=torch.int64, device=device)
        self.host_offsets = off


def encode_device(tok: Tokenizer, batch: DeviceBatch, with_special: bool = False) -> None:
    """One pass of the hot path over `batch`, asynchronous on torch's current stream.
    Results land in batch.ids / batch.out_off (out_off[-1] = token count)."""
    L = _ffi.lib()
    stream = torch.cuda.current_stream(batch.text.device).cuda_stream
    rc = L.spl_encode_batch_device(tok.handle, batch.text.data_ptr(), batch.n_bytes, batch.doc_off.data_ptr(),
                                   batch.n_docs, _ffi.SPL_WITH_SPECIAL if with_special else 0,
                                   batch.ids.data_ptr(), batch.ids.numel(), batch.out_off.data_ptr(), stream)
    if rc != 0:
        raise RuntimeError(f"spl_encode_batch_device failed ({rc}): {_ffi.last_error()}")


def reserve(tok: Tokenizer, n_bytes: int, n_docs: int) -> None:
    if _ffi.lib().spl_reserve(tok.handle, n_bytes, n_docs) != 0:
        raise RuntimeError(_ffi.last_error())


def result_csr(batch: DeviceBatch) -> Tuple[np.ndarray, np.ndarray]:
    off = batch.out_off.cpu().numpy().astype(np.uint64)
    ids = batch.ids[: int(off[-1])].cpu().numpy().view(np.uint32)
    return ids, off


class Comm:
    """One rank of the library's own RCCL communicator (include/splintr_hip.h: spl_comm_*).  The 128-byte id is
    made by rank 0 and handed to the others by whatever the caller has; `from_torch_group` uses an existing
    torch.distributed group (any backend: it only carries 128 bytes) for that."""

    def __init__(self, unique_id: bytes, rank: int, world: int, device: int):
        L = _ffi.lib()
        self._h = L.spl_comm_create(unique_id, rank, world, device)
        if not self._h:
            raise RuntimeError(f"spl_comm_create failed: {_ffi.last_error()}")
        self.rank, self.world, self.device = rank, world, device

    @staticmethod
    def unique_id() -> bytes:
        buf = ctypes.create_string_buffer(128)
        if _ffi.lib().spl_comm_unique_id(buf) != 0:
            raise RuntimeError(f"spl_comm_unique_id failed: {_ffi.last_error()}")
        return buf.raw

    @classmethod
    def from_torch_group(cls, device: torch.device, group=None) -> "Comm":
        import torch.distributed as dist
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        box = [cls.unique_id() if rank == 0 else None]
        if world > 1:
            dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        return cls(box[0], rank, world, device.index if device.index is not None else torch.cuda.current_device())

    @property
    def handle(self) -> int:
        return self._h

    def allgatherv_csr(self, ids: torch.Tensor, out_off: torch.Tensor, n_docs: int, all_ids: torch.Tensor,
                       all_off: torch.Tensor) -> Tuple[int, int]:
        """spl_allgatherv_csr on torch's current stream: exactly T_r ids and N_r offsets per rank, straight to
        their place of the global CSR.  Returns (total tokens, total documents); one host synchronisation."""
        nt, nd = ctypes.c_uint64(), ctypes.c_uint64()
        stream = torch.cuda.current_stream(ids.device).cuda_stream
        rc = _ffi.lib().spl_allgatherv_csr(self._h, ids.data_ptr(), out_off.data_ptr(), n_docs, all_ids.data_ptr(),
                                           all_ids.numel(), all_off.data_ptr(), all_off.numel(), ctypes.byref(nt),
                                           ctypes.byref(nd), stream)
        if rc != 0:
            raise RuntimeError(f"spl_allgatherv_csr failed ({rc}): {_ffi.last_error()}")
        return int(nt.value), int(nd.value)

    def ranks(self) -> int:
        """spl_comm_world: the number of ranks of the RCCL communicator the LIBRARY holds (what a scaling run checks
        against torch's world size)."""
        return int(_ffi.lib().spl_comm_world(self._h))

    def close(self) -> None:
        if getattr(self, "_h", None):
            _ffi.lib().spl_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _slab_words(max_tokens: int, max_docs: int, pack24: bool) -> int:
    """words of one slab: header, offsets, then the ids -- u32 each, or three bytes each (+ one word of slack for the unpacker's reads)"""
    return ((3 * max_tokens + 3) // 4 + 1 if pack24 else max_tokens) + max_docs + 4


_EXCH_STREAMS = {}


def exchange_stream(device: torch.device) -> "torch.cuda.Stream":
    """THE exchange stream of a device: one per process and GPU, shared by every GatherV / WaveGather on it.
    HIP maps streams onto a handful of hardware queues by what else the process has created: the n-th stream may land on the queue
    (or the command-processor pipe) the encoder's stream uses, and then the exchange runs BEHIND the encodes instead of beside them
    (measured: the fourth WaveGather of a process lost the whole overlap, 5.47 -> 6.05 ms per step).  The stream is therefore picked
    by measurement beside the stream that is current when it is first asked for (spl_pick_stream)."""
    key = (device.type, device.index if device.index is not None else torch.cuda.current_device())
    st = _EXCH_STREAMS.get(key)
    if st is None:
        st = _EXCH_STREAMS[key] = pick_stream(device, [torch.cuda.current_stream(device)])
    return st


def pick_stream(device: torch.device, beside) -> "torch.cuda.Stream":
    """A new stream that really runs beside the streams in `beside` (spl_pick_stream: measured, not assumed)."""
    import ctypes
    idx = device.index if device.index is not None else torch.cuda.current_device()
    arr = (ctypes.c_void_p * max(1, len(beside)))(*[ctypes.c_void_p(int(s.cuda_stream)) for s in beside])
    out, left = ctypes.c_void_p(), ctypes.c_double()
    if _ffi.lib().spl_pick_stream(idx, arr, len(beside), ctypes.byref(out), ctypes.byref(left)) != 0:
        raise RuntimeError(_ffi.last_error())
    st = torch.cuda.ExternalStream(out.value, device=device)
    st.conflict_us = left.value
    _PICKS.append({"beside": len(beside), "conflict_us": round(float(left.value), 1), "least_bad": bool(left.value >= PICK_CONFLICT_US)})
    return st


# Every stream this process picked by measurement, in order: what the candidate that was kept still lost to its neighbours (microseconds
# for four empty kernels beside a spinning stream, minus the same alone).  spl_pick_stream keeps the first candidate without a conflict and
# otherwise the LEAST BAD one -- silently, up to round 5; a scaling run records the picks and flags a leg that ran on a least-bad one.
PICK_CONFLICT_US = 15.0
_PICKS = []


def stream_picks():
    return list(_PICKS)


_ENC_STREAMS = {}


def encode_streams(device: torch.device, pair: int = 0):
    """Two streams per process and GPU on which WaveGather encodes consecutive waves in alternation -- pair `pair` of three pairs created
    once.  HIP multiplexes a process's streams onto a few hardware queues, and hardware queues onto four pipes of the command processor:
    two busy streams that end up on one pipe take turns instead of running side by side (kernel trace: every small kernel of such a stream
    takes 40 - 55 us instead of 4 - 13, the tile kernels 215 us instead of 130).  Which streams share is decided by what else the process has
    created; nothing in the HIP API tells.  The streams are therefore PICKED by measurement (spl_pick_stream) beside the exchange stream and
    each other; a caller that can afford a calibration still tries the pairs and keeps the fastest (bench.py)."""
    key = (device.type, device.index if device.index is not None else torch.cuda.current_device())
    st = _ENC_STREAMS.setdefault(key, {})
    p = pair % 3
    if p not in st:
        ex = exchange_stream(device)
        a = pick_stream(device, [ex])
        st[p] = (a, pick_stream(device, [ex, a]))
    return st[p]


def _set_pack24(tok: Optional[Tokenizer], on: bool, owner=None) -> None:
    """The slab format is the HANDLE's (spl_set_option "slab_pack24"), read by the encoder at every launch: a gatherer (re)asserts ITS format
    in front of each of its encodes and packs (one C call that stores an int), so that gatherers of both formats can live on one Tokenizer
    (one thread) without one silently changing the wire format of the other."""
    if tok is None:                              # the bucket logic driven without an encoder (CPU tests): slabs come from the caller
        return
    if _ffi.lib().spl_set_option(tok.handle, b"slab_pack24", 1 if on else 0) != 0:
        raise RuntimeError(_ffi.last_error())


class GatherV:
    """Pipelined, bucketed ragged all-gather of per-rank CSR results.

    Every batch is packed into a slab right after its encode (HIP kernel, same stream).  `depth`
    consecutive slabs form one bucket; a full bucket is handed to an exchange stream of its own:
    ONE all-gather of world x depth equal slabs (RCCL over xGMI: spl_allgather_slabs on the library's own
    communicator when `comm` is given, torch.distributed's all_gather_into_tensor otherwise) and ONE unpack launch
    that rebuilds the global CSR of each of the bucket's batches on every rank.  Two bucket SETS
    alternate -- each with its own send, receive and result buffers -- so the exchange of one bucket
    overlaps the encodes of the next and no cross-stream wait sits between two encodes.  Fewer,
    larger collectives on purpose: xGMI rings are per-link bound and a collective costs tens of
    microseconds of launch work, as much as encoding a whole 1 MB batch.

    Results: `on_bucket(results)` -- if set -- is called for every exchanged bucket with a list of
    (ids, off) views, one per batch in submission order (ids int32, the first off[-1] entries valid;
    off int64[world * max_docs + 1], entries past the global document count unspecified).  It runs
    with the exchange stream current: whatever it enqueues is ordered behind the unpack, and the
    set's buffers are reused only after that work.  `finish()` flushes a partial bucket, drains,
    and returns the views of the LAST submitted batch.  No host synchronisation per batch.
    """

    def __init__(self, tok: Tokenizer, device: torch.device, max_docs: int, max_tokens: int, group=None, depth: int = 8,
                 comm: "Comm" = None, collective: str = "allgather", pack24: bool = False):
        import torch.distributed as dist
        self.tok, self.dev, self.group, self.dist, self.comm = tok, device, group, dist, comm
        self.pack24 = bool(pack24)                # slab ids travel three bytes each (ids < 2**21): a quarter less on the links; EVERY rank alike
        _set_pack24(tok, self.pack24, self)
        if collective not in ("allgather", "p2p") or (collective == "p2p" and comm is None):
            raise ValueError("GatherV: collective is 'allgather' or -- with the library's communicator -- 'p2p'")
        self.collective = collective             # ncclAllGather of the bucket's slabs, or grouped send / recv of the same slabs
        self.world = comm.world if comm is not None else dist.get_world_size(group)
        self.depth = int(depth)
        self.max_docs = int(max_docs)
        self.max_tokens = int(max_tokens)
        self.cap_words = _slab_words(self.max_tokens, self.max_docs, self.pack24)
        if self.cap_words >= 1 << 32:
            raise ValueError("GatherV: a slab must stay below 2**32 words")
        self.off_stride = self.world * self.max_docs + 1
        self.ids_stride = self.world * self.max_tokens
        kw = dict(device=device)
        self.send = [torch.zeros(self.depth * self.cap_words, dtype=torch.int32, **kw) for _ in range(2)]
        self.recv = [torch.zeros(self.world * self.depth * self.cap_words, dtype=torch.int32, **kw) for _ in range(2)]
        # global CSR of each batch of the set's last exchanged bucket
        self.all_ids = [torch.zeros(self.depth * self.ids_stride, dtype=torch.int32, **kw) for _ in range(2)]
        self.all_off = [torch.zeros(self.depth * self.off_stride, dtype=torch.int64, **kw) for _ in range(2)]
        self.status = torch.zeros(1, dtype=torch.int32, **kw)
        self.exch = self._new_stream()
        self.packed = [self._new_event() for _ in range(2)]       # the bucket in set s is fully packed
        self.drained = [self._new_event() for _ in range(2)]      # the exchange of the bucket in set s is through
        self.in_flight = [False, False]
        self.cur, self.fill = 0, 0
        self.last = None                                          # (set, batches) of the last exchanged bucket
        self.on_bucket = None
        self._timing = None                                       # enable_timing(): [(start, end)] events per exchanged bucket

    # (overridable: the CPU test drives the bucket / event logic with a stub encoder on gloo)
    def _new_stream(self):
        return exchange_stream(self.dev)

    def _new_event(self):
        return torch.cuda.Event()

    def _main_stream(self):
        return torch.cuda.current_stream(self.dev)

    def _stream_ctx(self, st):
        return torch.cuda.stream(st)

    def enable_timing(self, on: bool = True) -> None:
        """Diagnostics for a scaling run: bracket every bucket's exchange (collective + unpack) with events on the
        exchange stream.  `exchange_ms()` then gives the time that stream spent in them -- beside the step time with
        and without the exchange that tells whether a rank is encode-bound or link-bound."""
        self._timing = [] if on else None

    def exchange_ms(self):
        """(total milliseconds, buckets) of the exchanges recorded since enable_timing(); synchronises."""
        if not self._timing:
            return 0.0, 0
        torch.cuda.synchronize(self.dev)
        total = sum(a.elapsed_time(b) for a, b in self._timing)
        n = len(self._timing)
        self._timing = []
        return float(total), n

    def bytes_per_bucket(self):
        """(bytes every rank SENDS per full bucket, bytes it RECEIVES): depth slabs of cap_words words out, world x that in."""
        sent = self.depth * self.cap_words * 4
        return sent, sent * self.world

    def _open_slab(self) -> torch.Tensor:
        main = self._main_stream()
        s = self.cur
        if self.fill == 0 and self.in_flight[s]:
            main.wait_event(self.drained[s])      # the set's previous exchange: normally long through
            self.in_flight[s] = False
        return self.send[s][self.fill * self.cap_words:(self.fill + 1) * self.cap_words]

    def _close_slab(self) -> None:
        self.fill += 1
        if self.fill == self.depth:
            self._exchange()

    def _pack(self, batch, slab, stream_ptr) -> None:
        _set_pack24(self.tok, self.pack24)
        rc = _ffi.lib().spl_gatherv_pack(self.tok.handle, batch.ids.data_ptr(), batch.out_off.data_ptr(), batch.n_docs,
                                         slab.data_ptr(), self.cap_words, self.max_docs, stream_ptr)
        if rc != 0:
            raise RuntimeError(_ffi.last_error())

    def _encode_packed(self, batch, slab, with_special, stream_ptr) -> None:
        _set_pack24(self.tok, self.pack24)
        rc = _ffi.lib().spl_encode_batch_device_packed(
            self.tok.handle, batch.text.data_ptr(), batch.n_bytes, batch.doc_off.data_ptr(), batch.n_docs,
            _ffi.SPL_WITH_SPECIAL if with_special else 0, batch.ids.data_ptr(), batch.ids.numel(),
            batch.out_off.data_ptr(), slab.data_ptr(), self.cap_words, self.max_docs, stream_ptr)
        if rc != 0:
            raise RuntimeError(f"spl_encode_batch_device_packed failed ({rc}): {_ffi.last_error()}")

    def _unpack(self, s, n, stream_ptr) -> None:
        rc = _ffi.lib().spl_gatherv_unpack_group(self.tok.handle, self.recv[s].data_ptr(), self.world, self.depth, n,
                                                 self.cap_words, self.max_docs, self.all_ids[s].data_ptr(),
                                                 self.ids_stride, self.all_off[s].data_ptr(), self.off_stride,
                                                 self.status.data_ptr(), stream_ptr)
        if rc != 0:
            raise RuntimeError(_ffi.last_error())

    def _allgather(self, s) -> None:
        """world x depth slabs, on the exchange stream (current here)."""
        if self.comm is not None:             # the collective behind the C ABI (librccl bound by the library)
            fn = _ffi.lib().spl_allgather_slabs_p2p if self.collective == "p2p" else _ffi.lib().spl_allgather_slabs
            rc = fn(self.comm.handle, self.send[s].data_ptr(), self.recv[s].data_ptr(), self.depth * self.cap_words, self.exch.cuda_stream)
            if rc != 0:
                raise RuntimeError(f"spl_allgather_slabs{'_p2p' if self.collective == 'p2p' else ''} failed ({rc}): {_ffi.last_error()}")
            return
        work = self.dist.all_gather_into_tensor(self.recv[s], self.send[s], group=self.group, async_op=True)
        work.wait()                           # the exchange stream (not the encode stream) waits for the collective

    def submit(self, batch: "DeviceBatch") -> None:
        """Pack `batch`'s current result into the open bucket (call right after encode_device on the
        same stream); a full bucket goes out."""
        slab = self._open_slab()
        self._pack(batch, slab, self._main_stream().cuda_stream)
        self._close_slab()

    def encode_and_submit(self, batch: "DeviceBatch", with_special: bool = False) -> None:
        """encode_device + submit in ONE call: the encoder's last kernel writes the slab itself
        (spl_encode_batch_device_packed), no separate pack launch."""
        slab = self._open_slab()
        self._encode_packed(batch, slab, with_special, self._main_stream().cuda_stream)
        self._close_slab()

    def _views(self, s, j):
        return (self.all_ids[s][j * self.ids_stride:(j + 1) * self.ids_stride],
                self.all_off[s][j * self.off_stride:(j + 1) * self.off_stride])

    def _exchange(self) -> None:
        main = self._main_stream()
        s, n = self.cur, self.fill
        self.packed[s].record(main)
        with self._stream_ctx(self.exch):
            self.exch.wait_event(self.packed[s])
            if self._timing is not None:
                ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                ev[0].record(self.exch)
            self._allgather(s)
            self._unpack(s, n, self.exch.cuda_stream)
            if self._timing is not None:
                ev[1].record(self.exch)
                self._timing.append(ev)
            if self.on_bucket is not None:
                self.on_bucket([self._views(s, j) for j in range(n)])
            self.drained[s].record(self.exch)
        self.in_flight[s] = True
        self.last = (s, n)
        self.cur ^= 1
        self.fill = 0

    def finish(self):
        """Flush and drain.  Returns (all_ids, all_off) of the LAST submitted batch: int32 ids (the
        first all_off[-1] entries are valid) and int64 offsets [world * max_docs + 1]."""
        if self.fill:
            self._exchange()
        main = self._main_stream()
        for s in (0, 1):
            if self.in_flight[s]:
                main.wait_event(self.drained[s])
                self.in_flight[s] = False
        if self.last is None:
            return self._views(0, 0)
        s, n = self.last
        return self._views(s, max(n - 1, 0))

    def overflowed(self) -> bool:
        return bool(self.status.item())


class WaveGather:
    """ONE batch, doc-sharded over the ranks (strong scaling), exchanged in WAVES so that the ids of wave k travel while wave k + 1 encodes.

    The batch's documents, in their order, are cut into `n_waves` waves and every wave into one contiguous slice per rank
    (splintr_amd.distributed.plan_waves); rank r encodes its slice of wave k straight into a slab (the encoder's last kernel writes it),
    an exchange stream of its own all-gathers the wave's `world` slabs and unpacks them BEHIND what the earlier waves left
    (spl_gatherv_unpack_at: running totals in device memory, advanced in stream order) -- no host synchronisation anywhere, the
    only exposed exchange is the last wave's.  After `finish()` every rank holds the CSR of the whole batch in document order.
    Order it replaces: Rayon's order-preserving collect, src/core/tokenizer.rs:932-942."""

    def __init__(self, tok: Tokenizer, device: torch.device, comm: "Comm", n_waves: int, max_docs: int, max_tokens: int,
                 total_tokens_cap: int, total_docs_cap: int, collective: str = "allgather", pack24: bool = False,
                 tok2: Optional[Tokenizer] = None, enc_pair: int = 0):
        self.tok, self.dev, self.comm, self.world = tok, device, comm, comm.world
        self.n_waves, self.max_docs, self.max_tokens = int(n_waves), int(max_docs), int(max_tokens)
        self.pack24 = bool(pack24)
        _set_pack24(tok, self.pack24, self)
        # A second handle of the same vocabulary (a workspace of its own): consecutive waves are then encoded on two streams in alternation,
        # wave k + 1's tile kernel starting while the stragglers of wave k's finish.  A rank's slice of a wave is small at high rank counts
        # (3.4 MB at 8 ranks x 8 waves), and launches of that size one after the other leave the GPU to every launch's ramp and tail: eight
        # of them took 0.90 ms on one stream, 0.69 ms on two -- one launch of the 27 MB: 0.63 (profiles/r05_wave_exchange.txt).
        self.toks = (tok,) if tok2 is None else (tok, tok2)
        if tok2 is not None:
            _set_pack24(tok2, self.pack24, self)
        self.enc = encode_streams(device, enc_pair) if tok2 is not None else None
        self.enc_pair = int(enc_pair)
        self.cap_words = _slab_words(self.max_tokens, self.max_docs, self.pack24)
        if self.cap_words >= 1 << 32:
            raise ValueError("WaveGather: a slab must stay below 2**32 words")
        self.collective = collective
        kw = dict(device=device)
        self.send = [torch.zeros(self.cap_words, dtype=torch.int32, **kw) for _ in range(self.n_waves)]
        self.recv = [torch.zeros(self.world * self.cap_words, dtype=torch.int32, **kw) for _ in range(self.n_waves)]
        self.all_ids = torch.zeros(int(total_tokens_cap), dtype=torch.int32, **kw)
        self.all_off = torch.zeros(int(total_docs_cap) + 1, dtype=torch.int64, **kw)
        self.run = torch.zeros(2, dtype=torch.int64, **kw)            # tokens, documents landed so far
        self.status = torch.zeros(1, dtype=torch.int32, **kw)
        self.exch = exchange_stream(device)
        self.encoded = [torch.cuda.Event() for _ in range(self.n_waves)]
        self.k = 0
        self._timing = None

    def enable_timing(self, on: bool = True) -> None:
        self._timing = [] if on else None

    def exchange_ms(self):
        if not self._timing:
            return 0.0, 0
        torch.cuda.synchronize(self.dev)
        total, n = sum(a.elapsed_time(b) for a, b in self._timing), len(self._timing)
        self._timing = []
        return float(total), n

    def begin(self) -> None:
        """Start a new batch: the running totals back to zero (on the exchange stream, behind the previous batch's last unpack;
        the main stream waits for it only through finish())."""
        self.k = 0
        with torch.cuda.stream(self.exch):
            self.run.zero_()
        if self.enc is not None:                 # what the caller has queued so far (the batches' text) comes first
            main = torch.cuda.current_stream(self.dev)
            for st in self.enc:
                st.wait_stream(main)

    def encode_and_submit(self, batch: "DeviceBatch", with_special: bool = False) -> None:
        """This rank's slice of the next wave: encode (slab written by the encoder's last kernel) on the current stream, exchange and
        unpack on the exchange stream."""
        L = _ffi.lib()
        k = self.k
        main = torch.cuda.current_stream(self.dev) if self.enc is None else self.enc[k & 1]
        _set_pack24(self.toks[k % len(self.toks)], self.pack24)
        rc = L.spl_encode_batch_device_packed(self.toks[k % len(self.toks)].handle, batch.text.data_ptr(), batch.n_bytes, batch.doc_off.data_ptr(), batch.n_docs,
                                              _ffi.SPL_WITH_SPECIAL if with_special else 0, batch.ids.data_ptr(), batch.ids.numel(),
                                              batch.out_off.data_ptr(), self.send[k].data_ptr(), self.cap_words, self.max_docs, main.cuda_stream)
        if rc != 0:
            raise RuntimeError(f"spl_encode_batch_device_packed failed ({rc}): {_ffi.last_error()}")
        self.encoded[k].record(main)
        with torch.cuda.stream(self.exch):
            self.exch.wait_event(self.encoded[k])
            if self._timing is not None:
                ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                ev[0].record(self.exch)
            fn = L.spl_allgather_slabs_p2p if self.collective == "p2p" else L.spl_allgather_slabs
            rc = fn(self.comm.handle, self.send[k].data_ptr(), self.recv[k].data_ptr(), self.cap_words, self.exch.cuda_stream)
            if rc != 0:
                raise RuntimeError(f"spl_allgather_slabs failed ({rc}): {_ffi.last_error()}")
            rc = L.spl_gatherv_unpack_at(self.tok.handle, self.recv[k].data_ptr(), self.world, self.cap_words, self.max_docs,
                                         self.all_ids.data_ptr(), self.all_ids.numel(), self.all_off.data_ptr(), self.all_off.numel(),
                                         self.run.data_ptr(), self.status.data_ptr(), self.exch.cuda_stream)
            if rc != 0:
                raise RuntimeError(f"spl_gatherv_unpack_at failed ({rc}): {_ffi.last_error()}")
            if self._timing is not None:
                ev[1].record(self.exch)
                self._timing.append(ev)
        self.k += 1

    def finish(self):
        """The main stream waits for the last wave's unpack.  Returns (all_ids, all_off, run): int32 ids, int64 offsets of the whole batch in
        document order, run = [total tokens, total documents] (device tensors; no host synchronisation here)."""
        torch.cuda.current_stream(self.dev).wait_stream(self.exch)
        return self.all_ids, self.all_off, self.run

    def overflowed(self) -> bool:
        return bool(self.status.item())
