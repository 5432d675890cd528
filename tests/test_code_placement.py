"""Code placement of the K-loop kernels (csrc/bhg_mlp.hip: k_layout_anchor).  The largest kernel of a CG iteration, k_wskpl, is as big as the
instruction cache two CUs share, and WHERE its instruction stream starts is worth 0.4 us per launch (profiles/r06_code_placement_of_the_k_loop_kernels.txt).
The product therefore anchors its template kernels behind a 16 KB-aligned, never-launched kernel whose size was chosen by two sweeps.  This test
(host tools only: it reads the gfx950 code object of the built bhg_mlp.o) fails when an edit moves the K-loop kernels relative to that anchor —
the cue to re-run the sweep (scripts/gpu_layout_sweep.sh) rather than to ship an untested placement."""
import os
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
OBJ = os.path.join(ROOT, "betty_amd", "csrc", "build", "bhg_mlp.o")
# (what the sweep was run on: offsets of the four kernels of a CG iteration from the anchor, product build)
EXPECTED = {"k_wskpc<2>": 0x17400, "k_wskpl<2, 4, false>": 0x54200, "k_headu<4>": 0x76200, "k_graw<1>": 0xa5200}


def _symbols():
    with tempfile.TemporaryDirectory() as tmp:
        fat, co = os.path.join(tmp, "fat"), os.path.join(tmp, "co")
        subprocess.run([f"{LLVM}/llvm-objcopy", "--dump-section", f".hip_fatbin={fat}", OBJ], check=True)
        subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={fat}", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                        f"--output={co}"], check=True)
        out = subprocess.run([f"{LLVM}/llvm-readelf", "-s", "--demangle", co], check=True, capture_output=True, text=True).stdout
    syms = {}
    for line in out.splitlines():
        parts = line.split(None, 7)
        if len(parts) == 8 and parts[3] == "FUNC":
            name = parts[7].replace("bhg::(anonymous namespace)::", "").replace("void ", "")
            syms[name.split("(")[0]] = int(parts[1], 16)
    return syms


@pytest.mark.skipif(not (os.path.exists(OBJ) and os.path.exists(f"{LLVM}/llvm-readelf")), reason="needs the built object and the ROCm LLVM tools")
def test_k_loop_kernels_sit_where_the_sweep_measured_them():
    syms = _symbols()
    anchor = syms["k_layout_anchor"]
    assert anchor % 16384 == 0, hex(anchor)
    got = {k: syms[k] - anchor for k in EXPECTED}
    assert got == EXPECTED, {k: hex(v) for k, v in got.items()}
