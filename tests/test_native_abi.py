"""The C-ABI library loads on a CPU-only box and exports every symbol include/bhg.h declares;
host-only entry points (layout builder, version, error string) behave."""
import ctypes
import os
import re

import numpy as np
import pytest

from betty_amd import _native

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "bhg.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(bhg_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree():
    declared = _declared_symbols()
    assert declared, "no prototypes found in include/bhg.h"
    assert sorted(_native.SYMBOLS) == declared


def test_library_exports_every_declared_symbol():
    lib = _native.load()
    for name in _declared_symbols():
        assert hasattr(lib, name), name
    assert lib.bhg_version() == 1
    assert lib.bhg_last_error() is not None


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(_native, "_libs", {})
    monkeypatch.setattr(_native, "LIB_PATH", str(tmp_path / "libbhg.so"))
    with pytest.raises(_native.NativeLibraryError, match="no CPU fallback"):
        _native.load()


def test_the_product_has_no_measurement_arm_and_the_measurement_build_has_them_all():
    """Round 5: libbhg.so is built WITHOUT the A/B table (every dbg(key, dflt) is the constant dflt, the arm-only kernel instances are
    not in its code object); libbhg_ab.so — the same sources with -DBHG_AB — carries the table for the every-arm tests and the
    same-box A/B measurements.  Same ABI: one binding serves both."""
    try:
        _native.use_ab(False)
        lib = _native.load()
        assert not _native.is_ab() and lib.bhg_debug_key_count() == 0 and lib.bhg_debug_key_name(0) is None
        assert lib.bhg_debug_set(b"mlp_proj", 0) != 0 and b"product build" in lib.bhg_last_error()
        with pytest.raises(_native.NativeLibraryError, match="use_ab"):
            _native.debug_set("mlp_proj", 0)
        _native.debug_set("mlp_proj", None)   # un-setting is a no-op everywhere
        _native.use_ab(True)
        ab = _native.load()
        assert ab is not lib and _native.is_ab() and ab.bhg_debug_key_count() >= 50
        keys = _native.debug_keys()
        for k in ("mlp_proj", "packed_chain", "lin_withhold_beta", "neumann_vnew", "cg_rhs_direct", "packed_prepare"):
            assert k in keys
        _native.debug_set("mlp_proj", 0)
        _native.debug_reset()
        for name in _declared_symbols():
            assert hasattr(ab, name), name
        # the arms are code: the product's shared object is smaller by the kernels only they launch
        assert os.path.getsize(_native.LIB_PATH) < 0.9 * os.path.getsize(_native.AB_LIB_PATH)
    finally:
        _native.use_ab(False)


def test_product_backend_refuses_cpu():
    import torch

    from betty_amd.backend import HipBackend

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(_native.NativeLibraryError, match="no CPU fallback"):
        HipBackend()


@pytest.mark.parametrize(
    "numels",
    [[], [0], [1], [63, 64, 65], [4096], [4097, 3, 0, 8192], [10, 5000, 4096, 1], [3072 * 2048, 2048, 2048 * 1536, 1536, 1536 * 384, 384, 3840, 10]],
)
def test_layout_builder(numels):
    lib = _native.load()
    T = len(numels)
    arr = (ctypes.c_int64 * max(T, 1))(*numels)
    flat = lib.bhg_layout_flat_size(arr, T)
    nch = lib.bhg_layout_num_chunks(arr, T)
    assert nch == sum((n + 4095) // 4096 for n in numels)
    starts = (ctypes.c_int64 * max(T, 1))()
    chunks = (_native.Chunk * max(nch, 1))()
    assert lib.bhg_layout_build(arr, T, starts, chunks) == 0
    # every element of every tensor is covered exactly once, flat starts are 64-aligned, no overlap
    cover = np.zeros(max(flat, 1), dtype=np.int32)
    pos = 0
    for t, n in enumerate(numels):
        assert starts[t] % 64 == 0 and starts[t] >= pos
        pos = starts[t] + n
    assert flat % 64 == 0 and flat >= pos
    per_tensor = [np.zeros(n, dtype=np.int32) for n in numels]
    for c in range(nch):
        ck = chunks[c]
        assert 0 < ck.len <= 4096 and ck.src_off % 4096 == 0
        assert ck.flat_off == starts[ck.tensor] + ck.src_off
        per_tensor[ck.tensor][ck.src_off : ck.src_off + ck.len] += 1
        cover[ck.flat_off : ck.flat_off + ck.len] += 1
    for a in per_tensor:
        assert (a == 1).all()
    assert cover.max(initial=0) <= 1
    assert lib.bhg_workspace_bytes(T) >= 24576 + 16 * T


def test_layout_builder_rejects_bad_sizes():
    lib = _native.load()
    arr = (ctypes.c_int64 * 2)(5, -1)
    assert lib.bhg_layout_flat_size(arr, 2) == -1
    assert lib.bhg_layout_num_chunks(arr, 2) == -1
    starts = (ctypes.c_int64 * 2)()
    chunks = (_native.Chunk * 4)()
    assert lib.bhg_layout_build(arr, 2, starts, chunks) == -1
    assert b"negative" in lib.bhg_last_error()


def test_argument_validation_without_gpu():
    """Bad arguments are rejected before any HIP call: negative return code + message, no crash.
    (Runs on the CPU-only box: none of these paths touches the device.)"""
    lib = _native.load()
    one = (ctypes.c_void_p * 1)(0)
    tab = ctypes.cast(one, _native._PP)
    null_tab = ctypes.cast(None, _native._PP)
    # NULL tensor table with T > 0
    assert lib.bhg_neumann_step(null_tab, 1, None, 1, None, None, 0.1, 0.0, 0.0, None, None) == -1
    assert b"NULL" in lib.bhg_last_error()
    # negative sizes
    assert lib.bhg_flatten(tab, -1, None, 0, None, 1.0, None, None) == -1
    assert b"negative" in lib.bhg_last_error()
    # chunk table missing
    assert lib.bhg_cg_init(tab, 1, None, 3, None, None, None, None, None) == -1
    # workspace missing
    assert lib.bhg_cg_step(tab, 1, 1, 1, None, None, None, 1.0, 0, 0.0, 0.0, 0, None, None) == -1
    assert b"workspace" in lib.bhg_last_error()
    # negative iteration index
    assert lib.bhg_cg_step(tab, 1, 1, 1, None, None, None, 1.0, -1, 0.0, 0.0, 0, 1, None) == -1
    # empty problems are a no-op, not an error
    assert lib.bhg_flatten(null_tab, 0, None, 0, None, 1.0, None, None) == 0
    assert lib.bhg_scale_flat(None, 0, 2.0, None) == 0
    # MLP descriptor checks
    assert lib.bhg_mlp_partial_floats(None) == 0
    d = _native.Mlp()
    d.L, d.B, d.Bp = 0, 1, 128
    assert lib.bhg_mlp_hvp(ctypes.byref(d), tab, tab, None) == -1
    d.L, d.Bp = 2, 100  # Bp must be a multiple of 128
    assert lib.bhg_mlp_hvp(ctypes.byref(d), tab, tab, None) == -1
    assert b"multiple of 128" in lib.bhg_last_error()
    # global-batch CG phases: descriptor checked before anything else (a NULL descriptor never reaches a launch)
    assert lib.bhg_mlp_cg_global_phase(None, None, None, None, None, None, 1, 0, 1, _native.BHG_CG_GLOBAL_CHAIN, 1, None, 1.0, 0.0,
                                       None, None, 0, None) == -1
    d.L, d.Bp = 2, 100
    assert lib.bhg_mlp_cg_global_phase(ctypes.byref(d), None, None, None, None, None, 1, 0, 1, _native.BHG_CG_GLOBAL_DOTS, 2, None, 1.0,
                                       0.0, None, None, 0, None) == -1
    assert b"multiple of 128" in lib.bhg_last_error()
    # timing API
    tot, cnt = ctypes.c_double(1.0), ctypes.c_int(7)
    assert lib.bhg_timing_enable(0) == 0
    assert lib.bhg_timing_read(0, ctypes.byref(tot), ctypes.byref(cnt)) == 0 and cnt.value == 0
    assert lib.bhg_timing_read(0, None, None) == -1
