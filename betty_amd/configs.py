"""``Config`` — the per-problem options object of the reference (betty/configs/problem_dataclass.py:5-48):
same names, same defaults, same positional order, so a reference user's
``Config(type="cg", cg_iterations=20, cg_alpha=1.0)`` means the same here.  Only the hypergradient knobs are
interpreted by this package; the others are carried for the caller slice in :mod:`betty_amd.problems`.
Every field records who reads it (``Config.describe()`` prints the table).
"""
from dataclasses import dataclass, field, fields


def _opt(default, read_by):
    return field(default=default, metadata={"read_by": read_by})


@dataclass
class Config:
    """Per-problem options (see module docstring)."""

    type: str = _opt("darts", "get_grads: key into jvp_fn_mapping for the LOWER problem of a hop — darts | neumann | cg | sama")
    unroll_steps: int = _opt(1, "Problem.step: inner steps between two hypergradient steps of the parent")
    first_order: bool = _opt(True, "Problem.backward: skip the best-response Jacobian when there is no lower path")
    retain_graph: bool = _opt(False, "get_grads: first-hop autograd.grad keeps the upper graph")
    allow_unused: bool = _opt(True, "Problem.backward: direct gradient tolerates unused parameters")
    gradient_accumulation: int = _opt(1, "Problem.step: the sync=True hop only fires on accumulation boundaries")
    gradient_clipping: float = _opt(0.0, "Problem.optimizer_step: max grad norm, 0 = off")
    precision: str = _opt("fp32", "Problem.training_step_exec: fp32 | fp16 | bf16 autocast")
    initial_dynamic_scale: float = _opt(4096.0, "fp16 loss scaler: initial scale (ImplicitProblem.optimizer_step)")
    scale_factor: float = _opt(2.0, "fp16 loss scaler: growth factor (back-off = 1 / scale_factor)")
    warmup_steps: int = _opt(0, "Problem.step: steps before the first upper-level update")
    log_step: int = _opt(-1, "logging cadence, -1 = off")
    log_local_step: bool = _opt(False, "logging")
    darts_alpha: float = _opt(0.01, "darts: finite-difference radius R, eps = R / ||v||")
    darts_multitask: bool = _opt(False, "darts: leave the inner weights perturbed after the call")
    sama_adam_alpha: float = _opt(1.0, "sama: finite-difference radius on the preconditioned direction")
    sama_multitask: bool = _opt(False, "sama: leave the inner weights perturbed after the call")
    neumann_iterations: int = _opt(1, "neumann: number of series terms K")
    neumann_alpha: float = _opt(1.0, "neumann: step of the series, also the final scale")
    cg_iterations: int = _opt(1, "cg: number of CG iterations K (no convergence test)")
    cg_alpha: float = _opt(1.0, "cg: scales Hp in the step length and the final x (not in the residual update)")

    @classmethod
    def describe(cls) -> str:
        """One line per option: name, default and the code that reads it."""
        return "\n".join(f"{f.name:24s} = {f.default!r:10}  {f.metadata['read_by']}" for f in fields(cls))
