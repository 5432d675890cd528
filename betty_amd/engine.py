"""A minimal ``Engine``: dependency graph -> hypergradient paths -> run loop.

Just enough of betty/engine.py (find_paths/dfs 232-264, parse_dependency 266-291, train_step/run
86-121) to drive the hot path end to end without the reference installed: paths are
``[upper, lower_1, ..., upper]`` exactly as ``Engine.find_paths`` builds them (checked in
tests/test_engine_shim.py against test/test_engine.py:124-130).  Validation, logging, early stopping,
roll-back and multi-node launch are NOT here — use ``betty.Engine`` + ``betty_amd.install()`` for
those; the hypergradient kernels are the same either way.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List

import torch


@dataclass
class EngineConfig:
    """Subset of betty/configs/engine_dataclass.py:5-27 that this engine honours."""

    train_iters: int = 50000
    valid_step: int = 500
    strategy: str = "default"  # "default" | "distributed" (DDP over RCCL; one process per GPU)
    backend: str = "nccl"
    roll_back: bool = False  # warm-start roll-back (engine_dataclass.py; problem.py:417-436)
    # strategy "distributed" only (extension, not in the reference): an upper problem with at least this many
    # parameters exchanges its hypergradient as ONE flat asynchronous all-reduce per path (betty_amd.distributed) —
    # the exchange of path i runs on the communication stream while the CG solve of path i+1 runs — instead of
    # through the DDP reducer's 25 MB buckets; its module is then NOT DDP-wrapped (its parameters AND buffers are
    # broadcast from rank 0 once, as DDP's constructor would; buffers are not re-broadcast every forward — use DDP for
    # modules with batch-statistics layers).  0 = always use DDP — the reference's behaviour and the default; opt in
    # with e.g. 1_000_000 for upper problems as large as the inner one (iMAML: M = N).
    flat_exchange_min_params: int = 0


class Engine:
    def __init__(self, problems, config=None, dependencies=None, device=None):
        self.config = config if config is not None else EngineConfig()
        self.problems: List = list(problems)
        self.dependencies: Dict = dependencies or {"u2l": {}, "l2u": {}}
        self.leaves: List = []
        self.global_step = 0
        self.device = device
        self._setup_device()
        self._parse_dependency()
        for a in self.problems:  # name attributes (engine.py:319-331)
            for b in self.problems:
                if a is not b:
                    a.set_problem_attr(b)

    # ---- device / distributed ---------------------------------------------------------------------------
    def _setup_device(self):
        import os

        strategy = self.config.strategy
        if strategy == "distributed":
            import torch.distributed as dist

            if not dist.is_initialized():
                os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
                dist.init_process_group(self.config.backend)
            if self.device is None and torch.cuda.is_available():
                local = int(os.environ.get("LOCAL_RANK", dist.get_rank() % max(torch.cuda.device_count(), 1)))
                torch.cuda.set_device(local)
                self.device = torch.device("cuda", local)
        if self.device is None:
            self.device = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")
        world = 1
        if strategy == "distributed":
            import torch.distributed as dist

            world = dist.get_world_size()
        for p in self.problems:
            p.device = self.device
            p._strategy = strategy
            p._world_size = world
            if p.module is not None:
                p.module.to(self.device)
                p.fwd = p.module
                n_params = sum(q.numel() for q in p.module.parameters())
                thr = self.config.flat_exchange_min_params
                p._flat_exchange = bool(strategy == "distributed" and world > 1 and thr > 0 and n_params >= thr
                                        and p.config.gradient_accumulation == 1)
                if p._flat_exchange:
                    # what DDP's constructor would have broadcast: parameters and buffers
                    p.synchronize_params(list(p.module.parameters()) + [b for b in p.module.buffers() if b.is_floating_point()])
                elif strategy == "distributed":
                    from torch.nn.parallel import DistributedDataParallel as DDP

                    # same wrapper arguments as problem.py:220-224
                    ids = [self.device.index] if self.device.type == "cuda" else None
                    p.fwd = DDP(p.module, device_ids=ids, gradient_as_bucket_view=True, find_unused_parameters=True)

    # ---- graph ---------------------------------------------------------------------------------------------
    def _lower_to_upper(self, problem):
        return self.dependencies.get("l2u", {}).get(problem, [])

    def find_paths(self, src, dst):
        """All backpropagation paths for the upper-to-lower edge dst -> src, each returned as
        [dst, ..., src, dst]: walk the lower-to-upper edges from src until dst is reached."""
        found = []

        def walk(node, trail):
            if node is dst:
                assert len(trail) > 1
                found.append(list(trail))
                return
            for nxt in self._lower_to_upper(node):
                trail.append(nxt)
                walk(nxt, trail)
                trail.pop()

        walk(src, [src])
        assert found, f"No path from {src.name} to {dst.name}!"
        return [list(reversed(t)) + [dst] for t in found]

    def _parse_dependency(self):
        self.leaves = []
        for p in self.problems:
            p.leaf = False
            p.clear_dependencies()
        for upper, lowers in self.dependencies.get("u2l", {}).items():
            for lower in lowers:
                upper.add_paths(self.find_paths(src=lower, dst=upper))
        for lower, uppers in self.dependencies.get("l2u", {}).items():
            for upper in uppers:
                lower.add_parent(upper)
                upper.add_child(lower)
        for p in self.problems:  # configure_roll_back (problem.py:681-689): only problems that have parents
            p._roll_back = bool(self.config.roll_back and len(p.parents) > 0)
        fed_by_someone = {u for uppers in self.dependencies.get("l2u", {}).values() for u in uppers}
        for p in self.problems:
            if p not in fed_by_someone:
                p.leaf = True
                self.leaves.append(p)

    # ---- run --------------------------------------------------------------------------------------------------
    def train_step(self):
        for leaf in self.leaves:
            leaf.step(global_step=self.global_step)

    def run(self):
        for p in self.problems:
            p.train()
        for _ in range(self.config.train_iters):
            self.global_step += 1
            self.train_step()
