"""Flat-buffer exchange of an upper-level hypergradient across data-parallel ranks.

In the reference the exchange is implicit: the last hop runs ``torch.autograd.backward`` so that
DistributedDataParallel's reducer all-reduces (mean) the upper gradients in 25 MB buckets and the
compute stream waits for it (betty/problems/problem.py:220-224, cg.py:58-63).  That is what
``sync=True`` still does here.  For problems whose UPPER parameter count is of the order of the inner
one (iMAML: M = N = 10.4 M in 122 tensors) this module offers the explicit form the
north-star asks for: the hypergradient returned by a ``sync=False`` call is gathered into ONE flat
HBM buffer by a single multi-tensor kernel, ONE asynchronous all-reduce (RCCL over xGMI: ``backend=
"nccl"`` on ROCm) runs on the communication stream while the next task's CG matvecs run on the compute
stream, and the averaged result is scattered back only when the optimizer needs it.

    handle = exchange_async(grads)        # after get_grads(..., do_sync=False)
    ...                                   # next task's cg(...): overlaps with the collective
    avg = handle.wait()                   # list of tensors shaped like `grads`, mean over ranks
"""
from __future__ import annotations

import weakref
from typing import List, Optional, Sequence

import torch
import torch.distributed as dist

from .backend import get_backend


class ExchangeHandle:
    def __init__(self, layout, flat, like, work, world, backend=None):
        self._layout, self._flat, self._like, self._work, self._world = layout, flat, like, work, world
        self._be = backend
        self._counted = None
        if work is not None and backend is not None:
            backend.collectives_in_flight += 1   # resident CG kernel is not eligible while RCCL kernels hold CUs
            # a handle that is dropped without wait() (an exception between issue and use) must not disable the resident
            # kernel for the rest of the process: the finalizer gives the count back
            self._counted = weakref.finalize(self, _release, backend)

    def wait(self) -> List[torch.Tensor]:
        """Block the CURRENT stream (not the host) on the collective and return the averaged tensors
        (views of the flat buffer).  The resident CG kernel becomes eligible again for launches on THIS stream
        (ordered behind the collective by the wait); solves on other streams must wait on the handle themselves."""
        if self._work is not None:
            self._work.wait()
            self._work = None
            if self._counted is not None:
                self._counted()   # runs _release once (and detaches the finalizer)
        return self._layout.views(self._flat, self._like)


def _release(backend):
    backend.collectives_in_flight = max(0, backend.collectives_in_flight - 1)


def exchange_async(grads: Sequence[torch.Tensor], group: Optional[dist.ProcessGroup] = None) -> ExchangeHandle:
    """Mean over the ranks of ``group`` of a list of tensors, as one flat asynchronous all-reduce."""
    grads = list(grads)
    be = get_backend()
    layout = be.layout(grads)
    flat = layout.new_flat()
    world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
    be.flatten(layout, grads, flat, 1.0 / world)  # pre-scaled: SUM of (g / world) = mean
    work = None
    if world > 1:
        work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group, async_op=True)
    return ExchangeHandle(layout, flat, grads, work, world, be if hasattr(be, "collectives_in_flight") else None)
