"""Device-resident batch encode: torch tensors in HBM in, CSR tensors in HBM out.

PyTorch is plumbing here (device memory, streams, torch.distributed); the work is done by
`spl_encode_batch_device` (include/splintr_hip.h) on torch's current HIP stream.
"""
from __future__ import annotations

from typing import Sequence, Tuple

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


class GatherV:
    """Pipelined ragged all-gather of per-rank CSR results (one RCCL collective per batch).

    pack (HIP kernel) -> all_gather_into_tensor of equal slabs (async, RCCL over xGMI) -> unpack
    (HIP kernel).  Two slab sets alternate, so the collective of batch i overlaps the encode of
    batch i+1; `finish()` drains the pipeline.  No host synchronisation per batch."""

    def __init__(self, tok: Tokenizer, device: torch.device, max_docs: int, max_tokens: int, group=None):
        import torch.distributed as dist
        self.tok, self.dev, self.group, self.dist = tok, device, group, dist
        self.world = dist.get_world_size(group)
        self.max_docs = int(max_docs)
        self.cap_words = int(max_tokens) + self.max_docs + 4
        self.send = [torch.zeros(self.cap_words, dtype=torch.int32, device=device) for _ in range(2)]
        self.recv = [torch.zeros(self.world * self.cap_words, dtype=torch.int32, device=device) for _ in range(2)]
        self.all_ids = torch.zeros(self.world * int(max_tokens), dtype=torch.int32, device=device)
        self.all_off = torch.zeros(self.world * self.max_docs + 1, dtype=torch.int64, device=device)
        self.status = torch.zeros(1, dtype=torch.int32, device=device)
        self.pending = None       # (work, slot)
        self.slot = 0

    def _unpack(self, slot: int) -> None:
        L = _ffi.lib()
        stream = torch.cuda.current_stream(self.dev).cuda_stream
        rc = L.spl_gatherv_unpack(self.tok.handle, self.recv[slot].data_ptr(), self.world, self.cap_words, self.max_docs,
                                  self.all_ids.data_ptr(), self.all_ids.numel(), self.all_off.data_ptr(),
                                  self.status.data_ptr(), stream)
        if rc != 0:
            raise RuntimeError(_ffi.last_error())

    def submit(self, batch: "DeviceBatch") -> None:
        """Queue the exchange of `batch`'s current result; completes the previous one first."""
        L = _ffi.lib()
        stream = torch.cuda.current_stream(self.dev).cuda_stream
        slot = self.slot
        rc = L.spl_gatherv_pack(self.tok.handle, batch.ids.data_ptr(), batch.out_off.data_ptr(), batch.n_docs,
                                self.send[slot].data_ptr(), self.cap_words, self.max_docs, stream)
        if rc != 0:
            raise RuntimeError(_ffi.last_error())
        if self.pending is not None:
            work, pslot = self.pending
            work.wait()                       # current stream waits for the collective of the previous batch
            self._unpack(pslot)
        work = self.dist.all_gather_into_tensor(self.recv[slot], self.send[slot], group=self.group, async_op=True)
        self.pending = (work, slot)
        self.slot ^= 1

    def finish(self):
        """Drain: returns (all_ids int32 view, all_off int64) of the LAST submitted batch."""
        if self.pending is not None:
            work, pslot = self.pending
            work.wait()
            self._unpack(pslot)
            self.pending = None
        return self.all_ids, self.all_off

    def overflowed(self) -> bool:
        return bool(self.status.item())
