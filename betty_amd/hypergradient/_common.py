"""Shared pieces of cg/neumann: inner-gradient graph, HVP providers, final mixed VJP."""
from __future__ import annotations

import contextlib
import os
import warnings
from typing import List, Sequence

import torch


def inner_gradient(curr):
    """``g = d L_in / d w`` with a graph (cg.py:27-32, neumann.py:31-36): re-evaluates the inner
    loss on the inner problem's last batch at the current inner weights."""
    in_loss = curr.training_step_exec(curr.cur_batch)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        in_grad = torch.autograd.grad(in_loss, curr.trainable_parameters(), create_graph=True)
    return in_grad


class AutogradHVP:
    """Default Hessian-vector product: PyTorch-ROCm double backward through the user's opaque
    ``training_step`` (cg.py:39-41, neumann.py:62)."""

    def __init__(self, in_grad, params):
        self.in_grad = in_grad
        self.params = list(params)

    def __call__(self, direction_views: Sequence[torch.Tensor]):
        return torch.autograd.grad(self.in_grad, self.params, grad_outputs=direction_views, retain_graph=True)


# ---- hipGraph replay of an opaque Hessian-vector product ----------------------------------------------------------------
# The double backward of a user's training_step is hundreds to thousands of small ATen launches (cfg 2's MLP: ~1.1 ms per
# HVP, launch-bound): the K HVPs of one solve run the SAME launch sequence on the SAME addresses — the autograd graph of
# `in_grad` is fixed for the step and the direction lives in one persistent flat vector — so the sequence is captured once
# (HIP graph) and replayed K - 2 times.
#   call 1: eager (settles library handles, workspaces, autotuning on the capture stream)
#   call 2: captured (hipStreamBeginCapture .. EndCapture around torch.autograd.grad), then replayed
#   call 3..K: replayed
# Autograd runs a backward node on the stream its forward ran on, so the graph of `in_grad` must be BUILT on the stream
# that later captures: `solve_stream` moves the whole solve to a library-owned side stream when the caller sits on the
# default (legacy) stream, which HIP cannot capture.  A capture that raises (a host sync inside the double backward, ...)
# leaves the wrapper eager for the rest of the solve.
# OPT-IN (measured on the MI355X, round 3): cfg 2 with the opaque HVP goes from 43.9 to 72.4 steps/s and seven solves of
# seven captures each replay cleanly, a ResNet-12 (GPU-bound double backward) gains nothing — and hipStreamEndCapture
# SEGFAULTS inside ROCm 7.2 on the tiny logistic-regression graph of the reference's regression scenario, which no
# try / except can contain.  So nothing is captured unless the user says so: `curr.hypergradient_graph = True` on the inner
# problem, or BHG_HVP_GRAPH=1 in the environment (=0 forces it off).
_SOLVE_STREAMS = {}
GRAPH_STATS = {"captures": 0, "replays": 0, "fallbacks": 0}


def hvp_graph_wanted(K: int, tensors, curr=None) -> bool:
    mode = os.environ.get("BHG_HVP_GRAPH", "")
    if mode == "0" or K < 3 or not tensors or not tensors[0].is_cuda:
        return False
    return mode == "1" or bool(getattr(curr, "hypergradient_graph", False))


@contextlib.contextmanager
def solve_stream(device, enabled: bool):
    """Run the body on a capturable stream (see above); joins the caller's stream on both sides."""
    if not enabled:
        yield None
        return
    cur = torch.cuda.current_stream(device)
    if cur != torch.cuda.default_stream(device):
        yield cur       # the caller already works on a stream of its own
        return
    side = _SOLVE_STREAMS.get(device.index)
    if side is None:
        side = _SOLVE_STREAMS[device.index] = torch.cuda.Stream(device)
    side.wait_stream(cur)
    try:
        with torch.cuda.stream(side):
            yield side
    finally:
        cur.wait_stream(side)


class GraphedHVP:
    """Wraps an eager HVP callable ``fn(direction_views) -> tuple of tensors`` (pure device work on fixed addresses)."""

    def __init__(self, fn):
        self.fn, self.calls, self.graph, self.out, self.key, self.dead = fn, 0, None, None, None, False

    @staticmethod
    def _key(views):
        return tuple((v.data_ptr(), tuple(v.shape), v.dtype) for v in views)

    def __call__(self, views):
        self.calls += 1
        if self.dead:
            return self.fn(views)
        key = self._key(views)
        if self.graph is not None:
            if key == self.key:
                self.graph.replay()
                GRAPH_STATS["replays"] += 1
                return self.out
            self.dead = True            # the caller moved the direction: the captured addresses are stale
            return self.fn(views)
        dev = views[0].device
        if self.calls == 1 or torch.cuda.current_stream(dev) == torch.cuda.default_stream(dev):
            self.key = key
            return self.fn(views)
        if key != self.key:
            self.dead = True
            return self.fn(views)
        graph = torch.cuda.CUDAGraph()
        try:
            # a private memory pool per graph (released with the graph at the end of the solve): a pool handle shared
            # across solves is dropped by the caching allocator when the previous graph dies (round-3 measurement: the second
            # capture then trips an internal assert of HIPCachingAllocator)
            graph.capture_begin(capture_error_mode="thread_local")
            try:
                out = self.fn(views)
            finally:
                graph.capture_end()
            graph.replay()
        except Exception as exc:   # not capturable on this stack: eager for the rest of the solve
            self.dead = True
            GRAPH_STATS["fallbacks"] += 1
            warnings.warn(f"betty_amd: hipGraph capture of the Hessian-vector product failed ({type(exc).__name__}: {exc}); "
                          "continuing with eager launches", RuntimeWarning)
            torch.cuda.synchronize(dev)
            return self.fn(views)
        GRAPH_STATS["captures"] += 1
        GRAPH_STATS["replays"] += 1
        self.graph, self.out = graph, tuple(out)
        return self.out


def mixed_vjp(in_grad, prev, neg_x_views: List[torch.Tensor], sync: bool):
    """Final hop to the upper parameters (cg.py:58-68, neumann.py:44-54).

    ``neg_x_views`` already holds ``-(alpha * x)``; by linearity of the VJP
    ``-(d(g.x)/d lambda) == d(g.(-x))/d lambda`` bit for bit, so no extra negation pass is needed.
    ``sync=True`` accumulates into ``.grad`` through ``backward`` (DDP reducer hooks fire) and
    returns None; ``sync=False`` returns the list for ``Problem.set_grads``."""
    upper = prev.trainable_parameters()
    if sync:
        torch.autograd.backward(in_grad, inputs=upper, grad_tensors=neg_x_views)
        return None
    return list(torch.autograd.grad(in_grad, upper, grad_outputs=neg_x_views))
