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
