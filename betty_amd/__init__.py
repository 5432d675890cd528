"""betty_amd — MI355X-native hypergradient backend behind Betty's plug-in API.

``betty_amd.hypergradient`` mirrors ``betty.hypergradient`` (``get_grads``, ``jvp_fn_mapping`` with
``cg`` / ``neumann`` / ``darts``), ``betty_amd.Config`` mirrors ``betty.configs.Config``.
``betty_amd.install()`` registers the HIP-backed functions in a live ``betty`` installation.
"""
from .configs import Config  # noqa: F401
from ._native import NativeLibraryError  # noqa: F401

__version__ = "0.1.0"


def install(betty_hypergradient=None):
    """Drop the MI355X implementations into the reference's registry
    (betty/hypergradient/__init__.py:13-19 is looked up at call time, line 35, so replacing
    entries in place is the documented extension point; ``betty.problems.problem`` bound
    ``get_grads`` by name at import, problem.py:16, and that function reads the same dict)."""
    from . import hypergradient as hg

    if betty_hypergradient is None:
        import betty.hypergradient as betty_hypergradient  # noqa: PLC0415
    for key in ("cg", "neumann", "darts", "sama", "cg_global"):   # cg_global: extension key (global-HVP mode)
        betty_hypergradient.jvp_fn_mapping[key] = hg.jvp_fn_mapping[key]
    return betty_hypergradient.jvp_fn_mapping
