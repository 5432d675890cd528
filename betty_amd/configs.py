"""``Config`` — per-problem options, field for field the reference's dataclass
(/root/reference betty/configs/problem_dataclass.py:5-48): same names, same defaults, so a
reference user's ``Config(type="cg", cg_iterations=20, cg_alpha=1.0)`` means the same here.
Only the hypergradient knobs are interpreted by this package; the rest are carried for the
problem/engine shim in :mod:`betty_amd.problems`.
"""
from dataclasses import dataclass


@dataclass
class Config:
    # which best-response-Jacobian approximation the LOWER problem on a path uses
    type: str = "darts"  # "darts" | "neumann" | "cg"
    unroll_steps: int = 1
    first_order: bool = True
    retain_graph: bool = False
    allow_unused: bool = True

    gradient_accumulation: int = 1
    gradient_clipping: float = 0.0

    precision: str = "fp32"
    initial_dynamic_scale: float = 4096.0
    scale_factor: float = 2.0

    warmup_steps: int = 0

    log_step: int = -1
    log_local_step: bool = False

    # finite difference (darts.py:29)
    darts_alpha: float = 0.01
    darts_multitask: bool = False

    # carried for API compatibility (SAMA is a §8(f) "next" row)
    sama_adam_alpha: float = 1.0
    sama_multitask: bool = False

    # Neumann series (neumann.py:41-42)
    neumann_iterations: int = 1
    neumann_alpha: float = 1.0

    # conjugate gradient (cg.py:38,42,56)
    cg_iterations: int = 1
    cg_alpha: float = 1.0
