"""betty_amd — MI355X-native hypergradient backend behind Betty's plug-in API.

``betty_amd.hypergradient`` mirrors ``betty.hypergradient`` (``get_grads``, ``jvp_fn_mapping`` with
``cg`` / ``neumann`` / ``darts``), ``betty_amd.Config`` mirrors ``betty.configs.Config``.
``betty_amd.install()`` registers the HIP-backed functions in a live ``betty`` installation.
"""
from .configs import Config  # noqa: F401
from ._native import NativeLibraryError  # noqa: F401

__version__ = "0.1.0"


def install(betty_hypergradient=None, auto_structure=None):
    """Drop the MI355X implementations into the reference's registry
    (betty/hypergradient/__init__.py:13-19 is looked up at call time, line 35, so replacing
    entries in place is the documented extension point; ``betty.problems.problem`` bound
    ``get_grads`` by name at import, problem.py:16, and that function reads the same dict).

    ``auto_structure=True`` (opt-in; ``False`` switches it off again, ``None`` leaves it as it is): inner problems WITHOUT a
    ``hypergradient_structure`` declaration are examined once — a Linear / ReLU stack under a sample-weighted cross-entropy that passes
    the declaration's own check against autograd takes the fused solver, everything else stays opaque
    (betty_amd/hypergradient/structured.py: ``_auto_structure``)."""
    from . import hypergradient as hg
    from .hypergradient import structured as _structured

    if auto_structure is not None:
        _structured.AUTO_STRUCTURE = bool(auto_structure)

    if betty_hypergradient is None:
        import betty.hypergradient as betty_hypergradient  # noqa: PLC0415
    for key in ("cg", "neumann", "darts", "sama", "cg_global", "neumann_global"):   # *_global: extension keys (global-batch mode)
        betty_hypergradient.jvp_fn_mapping[key] = hg.jvp_fn_mapping[key]
    return betty_hypergradient.jvp_fn_mapping
