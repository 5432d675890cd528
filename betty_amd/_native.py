"""ctypes binding of libbhg.so (the C ABI declared in include/bhg.h).

The library is the product: there is no CPU fallback.  If the shared object is missing the
import of anything that needs it raises :class:`NativeLibraryError` with the build command.
Symbols declared in ``include/bhg.h`` are listed in :data:`SYMBOLS`; tests check that every one
of them resolves.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, c_char_p, c_double, c_float, c_int, c_int32, c_int64, c_size_t, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("BHG_LIB") or os.path.join(_HERE, "csrc", "libbhg.so")  # BHG_LIB: another build of the same ABI
# the measurement build (the same sources with the A/B table compiled in, -DBHG_AB): what `bhg_debug_set` needs; see use_ab()
AB_LIB_PATH = os.environ.get("BHG_AB_LIB") or os.path.join(_HERE, "csrc", "libbhg_ab.so")

BHG_CHUNK_ELEMS = 4096
BHG_FLAT_ALIGN = 64
BHG_CG_AUTO, BHG_CG_STREAM, BHG_CG_RESIDENT = 0, 1, 2
BHG_CG_GLOBAL_CHAIN, BHG_CG_GLOBAL_UPDATE, BHG_CG_GLOBAL_DOTS = 0, 1, 2   # phases of bhg_mlp_cg_global_phase
BHG_CG_FX_BEGIN, BHG_CG_FX_CHAIN, BHG_CG_FX_GRAM, BHG_CG_FX_END = 0, 1, 2, 3   # phases of bhg_mlp_cg_fx_phase


class NativeLibraryError(RuntimeError):
    """libbhg.so is missing or a call into it failed."""


class Chunk(ctypes.Structure):
    """Mirror of ``bhg_chunk`` (include/bhg.h)."""

    _fields_ = [
        ("flat_off", c_int64),
        ("src_off", c_int64),
        ("tensor", c_int32),
        ("len", c_int32),
    ]


BHG_MLP_MAX_LAYERS = 32


class Mlp(ctypes.Structure):
    """Mirror of ``bhg_mlp`` (include/bhg.h)."""

    _fields_ = [
        ("L", c_int32),
        ("B", c_int32),
        ("Bp", c_int32),
        ("dims", c_int32 * (BHG_MLP_MAX_LAYERS + 1)),
        ("W", c_void_p * BHG_MLP_MAX_LAYERS),
        ("h", c_void_p * BHG_MLP_MAX_LAYERS),
        ("mask", c_void_p * BHG_MLP_MAX_LAYERS),
        ("delta", c_void_p * BHG_MLP_MAX_LAYERS),
        ("prob", c_void_p),
        ("sd", c_void_p),
        ("Rh", c_void_p * BHG_MLP_MAX_LAYERS),
        ("Rd", c_void_p * BHG_MLP_MAX_LAYERS),
        ("partial", c_void_p),
        ("partial_floats", c_size_t),
        ("ridge2", c_float),
        ("prepacked", c_int32),
    ]


_PP = POINTER(c_void_p)  # const void* const*
_CH = c_void_p  # const bhg_chunk* (device)

# name -> (restype, argtypes); mirrors include/bhg.h one to one.
SYMBOLS = {
    "bhg_version": (c_int, []),
    "bhg_last_error": (c_char_p, []),
    "bhg_layout_flat_size": (c_int64, [POINTER(c_int64), c_int]),
    "bhg_layout_num_chunks": (c_int64, [POINTER(c_int64), c_int]),
    "bhg_layout_build": (c_int, [POINTER(c_int64), c_int, POINTER(c_int64), POINTER(Chunk)]),
    "bhg_workspace_bytes": (c_size_t, [c_int]),
    "bhg_flatten": (c_int, [_PP, c_int, _CH, c_int, c_void_p, c_float, c_void_p, c_void_p]),
    "bhg_scatter": (c_int, [c_void_p, _PP, c_int, _CH, c_int, c_float, c_void_p, c_void_p]),
    "bhg_neumann_init": (c_int, [_PP, c_int, _CH, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "bhg_neumann_step": (
        c_int,
        [_PP, c_int, _CH, c_int, c_void_p, c_void_p, c_float, c_float, c_float, c_void_p, c_void_p],
    ),
    "bhg_cg_init": (c_int, [_PP, c_int, _CH, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "bhg_cg_init_masked": (c_int, [_PP, c_int, _CH, c_int, c_void_p, c_void_p, c_void_p, ctypes.c_uint64, c_void_p, c_void_p]),
    "bhg_cg_step": (
        c_int,
        [_PP, c_int, _CH, c_int, c_void_p, c_void_p, c_void_p, c_float, c_int, c_float, c_float, c_int, c_void_p, c_void_p],
    ),
    "bhg_cg_phase": (
        c_int,
        [c_int, _PP, c_int, _CH, c_int, c_void_p, c_void_p, c_void_p, c_float, c_int, c_float, c_float, c_void_p, c_void_p],
    ),
    "bhg_cg_partials_dev": (c_void_p, [c_void_p, c_int, c_int]),
    "bhg_cg_partials_count": (c_int, []),
    "bhg_cg_resident_capacity_chunks": (c_int, []),
    "bhg_cg_resident_ok": (c_int, []),
    "bhg_cg_resident_usable": (c_int, [c_int]),
    "bhg_cg_scalars_dev": (c_void_p, [c_void_p]),
    "bhg_cg_timeout_flag_dev": (c_void_p, [c_void_p]),
    "bhg_scale_flat": (c_int, [c_void_p, c_int64, c_float, c_void_p]),
    "bhg_darts_eps": (c_int, [_PP, c_int, _CH, c_int, c_double, c_void_p, c_void_p, c_void_p, c_void_p]),
    "bhg_axpy_multi": (c_int, [_PP, _PP, c_int, _CH, c_int, c_void_p, c_float, c_void_p, c_void_p]),
    "bhg_sama_adam_precondition": (
        c_int,
        [_PP, _PP, _PP, _PP, c_int, _CH, c_int, c_void_p, c_double, c_double, c_double, c_double, c_void_p, c_void_p],
    ),
    "bhg_debug_set": (c_int, [c_char_p, c_int]),
    "bhg_debug_unset": (c_int, [c_char_p]),
    "bhg_debug_reset": (None, []),
    "bhg_debug_key_count": (c_int, []),
    "bhg_debug_key_name": (c_char_p, [c_int]),
    "bhg_is_ab_build": (c_int, []),
    "bhg_timing_enable": (c_int, [c_int]),
    "bhg_timing_read": (c_int, [c_int, POINTER(c_double), POINTER(c_int)]),
    "bhg_logreg_prepare": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "bhg_logreg_tmp_floats": (c_size_t, [c_int, c_int]),
    "bhg_logreg_hvp": (
        c_int,
        [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p],
    ),
    "bhg_mlp_partial_floats": (c_size_t, [POINTER(Mlp)]),
    "bhg_mlp_hvp": (c_int, [POINTER(Mlp), _PP, _PP, c_void_p]),
    "bhg_mlp_hvp_mode": (c_int, [POINTER(Mlp), _PP, _PP, c_int, c_void_p]),
    "bhg_mlp_supports_native_prepare": (c_int, [POINTER(Mlp)]),
    "bhg_mlp_forward": (c_int, [POINTER(Mlp), _PP, c_void_p, c_void_p, c_void_p]),
    "bhg_mlp_backward": (c_int, [POINTER(Mlp), c_void_p, c_void_p]),
    "bhg_mlp_mixed_coeff": (c_int, [POINTER(Mlp), _PP, c_void_p, c_void_p, c_void_p]),
    "bhg_mlp_supports_fused_solve": (c_int, [POINTER(Mlp)]),
    "bhg_mlp_fused_ws_bytes": (c_size_t, [POINTER(Mlp)]),
    "bhg_mlp_cg_solve": (
        c_int,
        [POINTER(Mlp), c_void_p, c_void_p, c_void_p, POINTER(c_int64), _CH, c_int, c_int, c_float, c_float, c_void_p,
         c_void_p, c_size_t, c_void_p],
    ),
    "bhg_mlp_cg_state_mask": (ctypes.c_uint64, [POINTER(Mlp), c_int]),
    "bhg_mlp_cg_solve_rhs": (
        c_int,
        [POINTER(Mlp), c_void_p, c_void_p, c_void_p, POINTER(c_int64), _CH, c_int, c_int, c_float, c_float, c_void_p,
         c_void_p, c_size_t, _PP, c_void_p],
    ),
    "bhg_mlp_cg_global_phase": (
        c_int,
        [POINTER(Mlp), c_void_p, c_void_p, c_void_p, POINTER(c_int64), _CH, c_int, c_int, c_int, c_int, c_int, c_void_p, c_float,
         c_float, c_void_p, c_void_p, c_size_t, c_void_p],
    ),
    "bhg_mlp_fx_supported": (c_int, [POINTER(Mlp), c_int]),
    "bhg_mlp_fx_ws_bytes": (c_size_t, [POINTER(Mlp), c_int]),
    "bhg_mlp_fx_const_floats": (c_size_t, [POINTER(Mlp)]),
    "bhg_mlp_fx_slab_floats": (c_size_t, [POINTER(Mlp)]),
    "bhg_mlp_fx_scal_doubles": (c_size_t, [POINTER(Mlp)]),
    "bhg_mlp_cg_fx_phase": (
        c_int,
        [POINTER(Mlp), _PP, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_float, c_float, c_void_p, c_size_t,
         c_void_p, c_size_t, c_void_p],
    ),
    "bhg_mlp_neumann_fx_phase": (
        c_int,
        [POINTER(Mlp), _PP, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_float, c_float, c_void_p, c_size_t,
         c_void_p, c_size_t, c_void_p],
    ),
    "bhg_mlp_cg_mixed_coeff": (c_int, [POINTER(Mlp), c_void_p, c_void_p, c_float, c_void_p, c_size_t, c_void_p]),
    "bhg_mlp_timeout_flag_dev": (c_void_p, [POINTER(Mlp), c_void_p]),
    "bhg_mlp_stage_batch": (c_int, [POINTER(Mlp), c_void_p, c_void_p, c_void_p, c_void_p]),
    "bhg_mlp_supports_packed_prepare": (c_int, [POINTER(Mlp)]),
    "bhg_mlp_forward_packed": (c_int, [POINTER(Mlp), _PP, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "bhg_mlp_backward_packed": (c_int, [POINTER(Mlp), c_void_p, c_void_p, c_size_t, c_void_p]),
    "bhg_mwn_max_hidden": (c_int, []),
    "bhg_mwn_forward": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    "bhg_mwn_backward": (
        c_int,
        [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p],
    ),
    "bhg_mlp_neumann_solve": (
        c_int,
        [POINTER(Mlp), c_void_p, c_void_p, c_void_p, POINTER(c_int64), c_int, c_float, c_float, c_void_p, c_size_t, POINTER(c_int), c_void_p],
    ),
    "bhg_mlp_wsk_launches": (c_int64, []),
    "bhg_mlp_hoist_launches": (c_int64, []),
    "bhg_mlp_proj_iterations": (c_int64, []),
    "bhg_mlp_lin_launches": (c_int64, []),
    "bhg_mlp_neumann_mixed_coeff": (c_int, [POINTER(Mlp), _PP, c_void_p, c_void_p, c_float, c_int, c_int, c_void_p, c_size_t, c_void_p]),
    "bhg_mlp_plan_describe": (c_int, [POINTER(Mlp), c_int, c_int, ctypes.c_char_p, c_size_t]),
    "bhg_copy2d": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_int64, c_int64, c_void_p]),
    "bhg_bn_ws_bytes": (c_size_t, [c_int]),
    "bhg_bn_backward_vjp": (
        c_int,
        [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p,
         c_void_p, c_size_t, c_void_p],
    ),
}

_libs = {}          # path -> bound CDLL
_current = [None]   # path of the library load() returns: LIB_PATH (the product) unless use_ab() switched


_PY_ENV = {"BHG_LIB", "BHG_AB_LIB", "BHG_HVP_GRAPH", "BHG_ALL_RANKS_ON_GPU0"}   # what the PYTHON side reads; the library reads nothing
_warned_env = [False]


def _warn_legacy_env() -> None:
    """Rounds 1-3 selected measurement arms through BHG_* environment variables; since round 4 the library reads none, and scripts
    that still export them would silently measure the default arm under an A/B label.  Say so, once."""
    if _warned_env[0]:
        return
    _warned_env[0] = True
    stale = sorted(k for k in os.environ if k.startswith("BHG_") and k not in _PY_ENV)
    if stale:
        import warnings

        warnings.warn("betty_amd: environment variable(s) " + ", ".join(stale) + " are IGNORED — libbhg reads no environment variable. "
                      "Measurement arms are selected with betty_amd._native.use_ab() + debug_set(key, int) (bench.py --debug KEY=INT) "
                      "on the measurement build libbhg_ab.so.", RuntimeWarning, stacklevel=3)


def _bind(path: str) -> ctypes.CDLL:
    lib = _libs.get(path)
    if lib is not None:
        return lib
    _warn_legacy_env()
    if not os.path.exists(path):
        raise NativeLibraryError(
            f"{path} not found: the HIP extension has not been built. Run "
            "`python -c 'import __graft_entry__ as g; g.build()'` or `make -C betty_amd/csrc`. "
            "betty_amd has no CPU fallback."
        )
    lib = ctypes.CDLL(path)
    for name, (restype, argtypes) in SYMBOLS.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as exc:  # pragma: no cover - build/ABI mismatch
            raise NativeLibraryError(f"{path} does not export {name}") from exc
        fn.restype = restype
        fn.argtypes = argtypes
    _libs[path] = lib
    return lib


def load() -> ctypes.CDLL:
    """The library every call goes through: libbhg.so (the product: one form per solver, no measurement arm in its code object)
    unless use_ab() switched this process to libbhg_ab.so.  Loaded once; raises loudly when it has not been built."""
    return _bind(_current[0] or LIB_PATH)


def current_lib_path() -> str:
    return _current[0] or LIB_PATH


def is_ab() -> bool:
    return bool(load().bhg_is_ab_build())


def use_ab(on: bool = True) -> None:
    """Route every subsequent call of this process through the measurement build libbhg_ab.so (on) or back through the product
    (off).  Both can be loaded side by side (separate shared objects, separate state: launch counters, side streams, debug table);
    device buffers are plain memory and serve either.  Used by the `bhg_debug` pytest fixture and `bench.py --debug`."""
    if on:
        _bind(AB_LIB_PATH)
        _current[0] = AB_LIB_PATH
    else:
        _current[0] = None


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = load().bhg_last_error()
        raise NativeLibraryError(f"{what} failed with code {rc}: {msg.decode() if msg else ''}")


def _debug_key(key: str) -> bytes:
    """'BHG_MLP_PROJ' / 'MLP_PROJ' / 'mlp_proj' -> b'mlp_proj' (the names of round 1-3's environment switches still work)."""
    k = key.strip()
    if k.upper().startswith("BHG_"):
        k = k[4:]
    return k.lower().encode()


def debug_set(key: str, value) -> None:
    """Select a measurement / test arm of libbhg (include/bhg.h: bhg_debug_set).  value None = back to the shipped behaviour.
    The library itself reads no environment variable."""
    lib = load()
    if value is not None and not lib.bhg_is_ab_build():
        raise NativeLibraryError(f"bhg_debug_set({key}): the product libbhg.so carries no measurement arm — call "
                                 "betty_amd._native.use_ab() first (libbhg_ab.so; the `bhg_debug` fixture and `bench.py --debug` do)")
    if value is None:
        check(lib.bhg_debug_unset(_debug_key(key)), f"bhg_debug_unset({key})")
    else:
        check(lib.bhg_debug_set(_debug_key(key), int(value)), f"bhg_debug_set({key})")


def debug_reset() -> None:
    load().bhg_debug_reset()


def debug_keys():
    lib = load()
    return [lib.bhg_debug_key_name(i).decode() for i in range(lib.bhg_debug_key_count())]


def ptr_array(ptrs):
    """Host array of device pointers (``const void* const*``)."""
    arr = (c_void_p * len(ptrs))(*ptrs)
    return ctypes.cast(arr, _PP), arr


def plan_describe(dims, B: int, algo: str = "cg", keep_solution: bool = False) -> dict:
    """The form the fused solvers take for an MLP of widths ``dims`` and a batch of ``B`` rows, as a dict (include/bhg.h:
    bhg_mlp_plan_describe — host logic only: works on a box without a GPU).  ``dims`` are the widths the KERNELS see: a network with
    widths that are not multiples of 32 runs on its zero-padded twin (hypergradient/_mlp_hip.py: padded_dims)."""
    lib = load()
    d = Mlp()
    d.L, d.B, d.Bp = len(dims) - 1, int(B), (int(B) + 127) // 128 * 128
    for i, v in enumerate(dims):
        d.dims[i] = int(v)
    buf = ctypes.create_string_buffer(2048)
    check(lib.bhg_mlp_plan_describe(ctypes.byref(d), {"cg": 0, "neumann": 1}[algo], 1 if keep_solution else 0, buf, 2048), "bhg_mlp_plan_describe")
    out, text = {}, buf.value.decode()
    import re  # noqa: PLC0415

    for key, quoted, plain in re.findall(r'(\w+)=(?:"([^"]*)"|(\S+))', text):
        val = quoted if quoted else plain
        out[key] = int(val) if re.fullmatch(r"-?\d+", val) else val
    return out
