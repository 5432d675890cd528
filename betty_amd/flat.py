"""Flat fp32 state vectors over a list of parameter-shaped tensors.

The reference keeps CG/Neumann state as Python lists of per-parameter tensors and rebuilds a
flat copy with ``to_vec`` (``torch.cat``) four times per CG iteration
(/root/reference betty/utils.py:117-118, betty/hypergradient/cg.py:42-44,51).  Here the state
lives in flat HBM buffers for the whole solve; per-parameter *views* of the flat direction
vector are what autograd receives as ``grad_outputs``, and the HVP tensors autograd returns are
consumed in place by the fused kernels through a pointer table.

Layout (built by ``bhg_layout_build`` in libbhg, so C and Python agree): tensor ``t`` occupies
``flat[start_t : start_t + numel_t]`` with ``start_t`` a multiple of 64 elements (256 B); padding
is zero and never written.  Work is cut into chunks of at most 4096 elements of one tensor.
"""
from __future__ import annotations

import ctypes
from typing import List, Sequence, Tuple

import numpy as np
import torch

from . import _native


class FlatLayout:
    """Chunk table + flat offsets for one list of tensor sizes on one device."""

    def __init__(self, numels: Sequence[int], device: torch.device):
        lib = _native.load()
        self.numels: Tuple[int, ...] = tuple(int(n) for n in numels)
        self.T = len(self.numels)
        self.device = torch.device(device)
        arr = (ctypes.c_int64 * max(self.T, 1))(*self.numels)
        self.flat_size = int(lib.bhg_layout_flat_size(arr, self.T))
        self.n_chunks = int(lib.bhg_layout_num_chunks(arr, self.T))
        if self.flat_size < 0 or self.n_chunks < 0:
            raise ValueError("invalid tensor sizes for a flat layout")
        starts = (ctypes.c_int64 * max(self.T, 1))()
        chunks = (_native.Chunk * max(self.n_chunks, 1))()
        _native.check(lib.bhg_layout_build(arr, self.T, starts, chunks), "bhg_layout_build")
        self.starts: Tuple[int, ...] = tuple(int(starts[i]) for i in range(self.T))
        self.total = int(sum(self.numels))
        # chunk table as raw bytes, uploaded once
        raw = np.frombuffer(chunks, dtype=np.uint8, count=ctypes.sizeof(_native.Chunk) * self.n_chunks).copy()
        self.chunks_host = raw
        self.chunks_dev = torch.from_numpy(raw).to(self.device) if self.n_chunks > 0 else torch.empty(0, dtype=torch.uint8, device=self.device)
        self.workspace_bytes = int(lib.bhg_workspace_bytes(self.T))
        self.workspace = torch.zeros(self.workspace_bytes, dtype=torch.uint8, device=self.device)
        self._pool: List[torch.Tensor] = []

    # -- buffers ---------------------------------------------------------------------------
    def new_flat(self) -> torch.Tensor:
        """A zero-initialised flat vector (padding must stay zero)."""
        return torch.zeros(self.flat_size, dtype=torch.float32, device=self.device)

    def state(self, n: int) -> List[torch.Tensor]:
        """``n`` cached flat state vectors (reused across calls; padding is zero, payload is
        overwritten by the init kernels)."""
        while len(self._pool) < n:
            self._pool.append(self.new_flat())
        return self._pool[:n]

    def views(self, flat: torch.Tensor, like: Sequence[torch.Tensor]) -> List[torch.Tensor]:
        """Per-tensor views of ``flat`` shaped like ``like`` (no copy)."""
        return [flat[s : s + n].view(t.shape) for s, n, t in zip(self.starts, self.numels, like)]


_LAYOUTS = {}


def layout_for(tensors: Sequence[torch.Tensor]) -> FlatLayout:
    key = (tuple(int(t.numel()) for t in tensors), str(tensors[0].device) if len(tensors) else "cpu")
    lay = _LAYOUTS.get(key)
    if lay is None:
        dev = tensors[0].device if len(tensors) else torch.device("cpu")
        lay = FlatLayout(key[0], dev)
        _LAYOUTS[key] = lay
    return lay
