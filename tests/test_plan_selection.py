"""The map "shape -> solver form" (the plan logic had no unit test of its own beyond launch counters).

``bhg_mlp_plan_describe`` (include/bhg.h) evaluates the host logic that picks the form — hoist_plan's shape rules and cost model, the
solvers' set-up code (projection level, the chain's first product by linearity, the head launch with the recurrences, the closing
launch) — WITHOUT launching anything, so this file runs on the CPU box.  The GPU half ties the description to what the library then
does: the launch counters of a real solve on the same shapes."""
import pytest

from betty_amd import _native
from betty_amd.hypergradient import _mlp_hip

SIX = "six-launch (k_wskpl .. k_headu .. k_graw)"
SIX_CLASS = "six-launch-class (k_wskpl first, recurrences beside the chain)"
PSTEP = "fully-projected (k_pstep launch)"
NEU = "projected-neumann (update inside k_graw)"

# (dims the kernels see, batch) -> cg solution-free, cg with x, neumann solution-free, neumann with accumulator; closing launch of the projected forms
CASES = [
    # the metric workload and its family: four layers, narrow head (<= 12 classes, last hidden width <= 512): the six-launch form
    ([3072, 2048, 1536, 384, 10], 100, SIX, "projected-keep-state", NEU, "classic", "k_graw"),
    ([800, 512, 256, 128, 10], 100, SIX, "projected-keep-state", NEU, "classic", "k_graw"),         # the padded twin of 784-512-256-128-10
    ([512, 256, 256, 64, 10], 128, SIX, "projected-keep-state", NEU, "classic", "k_graw"),
    # batches beyond 128 rows: the K-looped closing launch
    ([512, 256, 256, 64, 10], 200, SIX, "projected-keep-state", NEU, "classic", "k_grawk"),
    ([256, 256, 128, 64, 10], 256, SIX, "projected-keep-state", NEU, "classic", "k_grawk"),
    # four layers whose head the head-with-recurrences launch does not take (last hidden layer > 512 wide; more than 12 classes)
    ([256, 256, 192, 640, 10], 100, SIX_CLASS, "projected-keep-state", NEU, "classic", "k_graw"),
    ([256, 256, 128, 64, 24], 100, SIX_CLASS, "projected-keep-state", NEU, "classic", "k_graw"),
    # deeper than four layers: k_wskpl with its update blocks behind the publisher
    ([256, 256, 256, 256, 128, 10], 100, SIX_CLASS, "projected-keep-state", NEU, "classic", "k_graw"),
    ([512, 512, 256, 256, 128, 10], 300, SIX_CLASS, "projected-keep-state", NEU, "classic", "k_grawk"),
    ([256, 192, 128, 64, 32, 10], 300, "hoisted", "hoisted", "hoisted", "classic", None),   # (the same depth, narrow: the cost model declines — B^2 Gram work)
    # three layers: no product between the first and the pre-head one -> the k_pstep launch stays
    ([256, 384, 128, 10], 100, PSTEP, "projected-keep-state", NEU, "classic", "k_graw"),
    # the cost model: a batch wider than the narrowest hidden layer, or eight narrow layers -> hoisted chain, no projection
    ([256, 384, 128, 10], 1024, "hoisted", "hoisted", "hoisted", "classic", None),
    ([64, 96, 64, 32, 64, 96, 32, 64, 10], 50, "hoisted", "hoisted", "hoisted", "classic", None),
    # two layers, or widths that are not multiples of 32 (as the kernels would see them WITHOUT the twin): the classic chain
    ([256, 192, 10], 128, "classic", "classic", "classic", "classic", None),
    ([784, 512, 256, 128, 10], 100, "classic", "classic", "classic", "classic", None),
    ([70, 130, 36, 10], 100, "classic", "classic", "classic", "classic", None),
]


@pytest.mark.parametrize("dims,B,cg_free,cg_x,neu_free,neu_acc,closing", CASES, ids=lambda v: str(v) if isinstance(v, (list, int)) else None)
def test_shape_to_form(dims, B, cg_free, cg_x, neu_free, neu_acc, closing):
    got = {(a, k): _native.plan_describe(dims, B, a, k) for a in ("cg", "neumann") for k in (False, True)}
    assert all(v["fused"] == 1 and v["narrow_head"] == 1 for v in got.values())
    assert got[("cg", False)]["form"] == cg_free and got[("cg", True)]["form"] == cg_x
    assert got[("neumann", False)]["form"] == neu_free and got[("neumann", True)]["form"] == neu_acc
    d = got[("cg", False)]
    assert d["lin"] == (1 if cg_free in (SIX, SIX_CLASS) else 0) and d["lin_head"] == (1 if cg_free == SIX else 0)
    assert d["upd_first"] == (1 if (cg_free == SIX_CLASS and len(dims) - 1 > 4) else 0)
    assert d["proj_level"] == (2 if cg_free in (SIX, SIX_CLASS, PSTEP) else 0)
    assert got[("cg", True)]["proj_level"] == (1 if cg_x == "projected-keep-state" else 0)
    if closing is not None:
        assert d["closing"] == closing == got[("neumann", False)]["closing"]
        assert d["gram_floats"] > 0
    else:
        assert d["gram_floats"] == 0, "no Gram region is carved when the plan does not project"
    # the packed once-per-step passes go with the hoisted plan
    assert d["packed_prepare"] == d["plan_ok"]
    # round 6: the global-batch solve (Config(type="cg_global")) exchanges batch-sized factors exactly where the plan projects and the
    # caller does not ask for x; the slab the ranks gather per iteration holds Rd_0 .. Rd_{L-1}, Rh_0 .. Rh_{L-2} of the padded batch
    assert d["global_form"] == ("factor-exchange" if d["proj_level"] == 2 else "one-pass")
    assert got[("cg", True)]["global_form"] == "one-pass"
    if d["proj_level"] == 2:
        Bp = (B + 127) // 128 * 128
        assert d["fx_slab_bytes"] >= 4 * Bp * (sum(dims[1:]) + sum(dims[1:-1])) and d["fx_ws_bytes_world8"] > 0
    else:
        assert d["fx_slab_bytes"] == 0 and d["fx_ws_bytes_world8"] == 0


def test_wide_head_has_no_fused_solver_and_the_twin_rounds_widths_up():
    d = _native.plan_describe([192, 256, 128, 1000], 72)               # beyond the head kernels' 256 classes
    assert (d["fused"], d["narrow_head"], d["form"]) == (0, 0, "unfused")
    d = _native.plan_describe([256, 384, 128, 100], 100)               # round 6: a 100-class head takes the fused forms
    assert (d["fused"], d["narrow_head"], d["form"]) == (1, 1, PSTEP)
    assert _native.plan_describe([512, 256, 256, 64, 100], 100)["form"] == SIX_CLASS
    assert _native.plan_describe([192, 256, 132, 10], 72)["form"] == "classic"      # feature width % 32 != 0 as the kernels see it ...
    assert _native.plan_describe([192, 256, 130, 10], 72)["form"] == "unfused"      # (% 4 != 0: not even the narrow-head kernels)
    assert _mlp_hip.padded_dims([192, 256, 130, 10]) == (192, 256, 160, 10)          # ... so the host side hands them the twin
    assert _native.plan_describe(_mlp_hip.padded_dims([192, 256, 130, 10]), 72)["form"] == PSTEP
    assert _mlp_hip.padded_dims([784, 512, 250, 100, 10]) == (800, 512, 256, 128, 10)
    assert _mlp_hip.padded_dims([3072, 2048, 1536, 384, 10]) == (3072, 2048, 1536, 384, 10)
    with pytest.raises(_native.NativeLibraryError):
        _native.plan_describe([10], 5)   # no layer


@pytest.mark.gpu
@pytest.mark.parametrize("dims,B", [([512, 256, 256, 64, 10], 128), ([256, 384, 128, 10], 100), ([784, 512, 256, 128, 10], 100),
                                    ([256, 256, 256, 256, 128, 10], 100), ([256, 192, 10], 128), ([256, 384, 128, 10], 1024)], ids=lambda v: str(v))
def test_the_description_is_what_the_library_then_does(dims, B):
    """A real solve on the same shapes: the launch counters agree with the description (cg without a solution vector)."""
    import sys
    import os

    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import test_gpu_parity as tgp

    lib = _native.load()
    K = 4
    d = _native.plan_describe(_mlp_hip.padded_dims(dims) if len(dims) - 1 >= 3 else dims, B)
    h0, p0, l0 = lib.bhg_mlp_hoist_launches(), lib.bhg_mlp_proj_iterations(), lib.bhg_mlp_lin_launches()
    tgp._run_solver("cg", dims, B, 0.5, K, 7, True, keep=False)
    dh, dp, dl = lib.bhg_mlp_hoist_launches() - h0, lib.bhg_mlp_proj_iterations() - p0, lib.bhg_mlp_lin_launches() - l0
    want_h = 0 if not d["hoist"] else (1 if d["proj_level"] >= 1 else K)
    assert (dh, dp, dl) == (want_h, (K - 1) if d["proj_level"] == 2 else 0, K if d["lin"] else 0), (d, dh, dp, dl)
