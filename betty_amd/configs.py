"""``Config`` — the per-problem options object of the reference (betty/configs/problem_dataclass.py:5-48),
rebuilt from a field table: same names, same defaults, same positional order, so a reference user's
``Config(type="cg", cg_iterations=20, cg_alpha=1.0)`` means the same here.  Only the hypergradient knobs are
interpreted by this package; the others are carried for the caller slice in :mod:`betty_amd.problems`.
The table records, per field, who reads it — ``Config.describe()`` prints it.
"""
from dataclasses import field, make_dataclass

# (name, type, default, read by)
_SPEC = (
    ("type", str, "darts",
     "get_grads: key into jvp_fn_mapping for the LOWER problem of a hop — darts | neumann | cg | sama"),
    ("unroll_steps", int, 1, "Problem.step: inner steps between two hypergradient steps of the parent"),
    ("first_order", bool, True, "Problem.backward: skip the best-response Jacobian when there is no lower path"),
    ("retain_graph", bool, False, "get_grads: first-hop autograd.grad keeps the upper graph"),
    ("allow_unused", bool, True, "Problem.backward: direct gradient tolerates unused parameters"),
    ("gradient_accumulation", int, 1, "Problem.step: the sync=True hop only fires on accumulation boundaries"),
    ("gradient_clipping", float, 0.0, "Problem.optimizer_step: max grad norm, 0 = off"),
    ("precision", str, "fp32", "Problem.training_step_exec: fp32 | fp16 | bf16 autocast"),
    ("initial_dynamic_scale", float, 4096.0, "fp16 loss scaler"),
    ("scale_factor", float, 2.0, "fp16 loss scaler"),
    ("warmup_steps", int, 0, "Problem.step: steps before the first upper-level update"),
    ("log_step", int, -1, "logging cadence, -1 = off"),
    ("log_local_step", bool, False, "logging"),
    ("darts_alpha", float, 0.01, "darts: finite-difference radius R, eps = R / ||v||"),
    ("darts_multitask", bool, False, "darts: leave the inner weights perturbed after the call"),
    ("sama_adam_alpha", float, 1.0, "sama: finite-difference radius on the preconditioned direction"),
    ("sama_multitask", bool, False, "sama: leave the inner weights perturbed after the call"),
    ("neumann_iterations", int, 1, "neumann: number of series terms K"),
    ("neumann_alpha", float, 1.0, "neumann: step of the series, also the final scale"),
    ("cg_iterations", int, 1, "cg: number of CG iterations K (no convergence test)"),
    ("cg_alpha", float, 1.0, "cg: scales Hp in the step length and the final x (not in the residual update)"),
)


def _describe(cls):
    """One line per option: name, default and the code that reads it."""
    return "\n".join(f"{n:24s} = {d!r:10}  {doc}" for n, _t, d, doc in _SPEC)


Config = make_dataclass(
    "Config",
    [(n, t, field(default=d, metadata={"read_by": doc})) for n, t, d, doc in _SPEC],
    namespace={"describe": classmethod(_describe), "__doc__": "Per-problem options (see module docstring)."},
)
Config.__module__ = __name__
