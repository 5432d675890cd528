#!/usr/bin/env python
"""Build-time gate: no register spills in libbhg's kernels.

Reads the gfx950 code objects out of betty_amd/csrc/build/*.o (the product) and build_ab/*.o (the measurement build) (the .hip_fatbin section -> clang-offload-bundler ->
llvm-readelf --notes) and fails when any kernel reports a non-zero `.vgpr_spill_count` or `.private_segment_fixed_size`
(scratch memory).  SGPR spills go to VGPR lanes (no memory traffic) and are reported, not refused.  Called by
__graft_entry__.build(); prints the five fattest kernels.

    python scripts/check_spills.py [--list]
"""
import glob
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
TARGET = "hipv4-amdgcn-amd-amdhsa--gfx950"


def kernels_of(obj, tmp):
    fat = os.path.join(tmp, os.path.basename(obj) + ".fatbin")
    co = os.path.join(tmp, os.path.basename(obj) + ".co")
    r = subprocess.run([f"{LLVM}/llvm-objcopy", "--dump-section", f".hip_fatbin={fat}", obj], capture_output=True)
    if r.returncode != 0 or not os.path.exists(fat) or os.path.getsize(fat) == 0:
        return []   # host-only object
    subprocess.check_call([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--targets={TARGET}",
                           f"--input={fat}", f"--output={co}"])
    notes = subprocess.check_output([f"{LLVM}/llvm-readelf", "--notes", co], text=True)
    out, name, cur = [], None, {}
    for line in notes.splitlines():
        s = line.strip().lstrip("- ").strip()
        if s.startswith(".name:"):
            name = s.split(":", 1)[1].strip()
            cur = {}
        for key in (".private_segment_fixed_size", ".sgpr_spill_count", ".vgpr_count", ".agpr_count", ".vgpr_spill_count"):
            if s.startswith(key + ":"):
                cur[key] = int(s.split(":", 1)[1])
                if key == ".vgpr_spill_count":
                    out.append((name, dict(cur)))
    return out


def scratch_ops(obj, tmp, kernel):
    """scratch_* / stack buffer_* instructions between the kernel's symbol and the next one (-1: symbol not found)."""
    co = os.path.join(tmp, os.path.basename(obj) + ".co")
    dis = subprocess.check_output([f"{LLVM}/llvm-objdump", "-d", co], text=True)
    n, inside = -1, False
    for line in dis.splitlines():
        if line.endswith(">:"):
            if inside:
                break
            inside = f"<{kernel}>:" in line
            if inside:
                n = 0
        elif inside and ("scratch_" in line or ("buffer_" in line and " offen" in line and "s[0:3]" in line)):
            n += 1
    return n


def short(n):
    import re

    m = re.search(r"(k_[A-Za-z0-9_]+(?:<[^>]*>)?)", demangle(n))
    return m.group(1) if m else n


def demangle(n):
    for tool in (f"{LLVM}/llvm-cxxfilt", "c++filt"):
        try:
            return subprocess.check_output([tool, n], text=True).strip()
        except Exception:
            continue
    return n


# Kernels that may DECLARE a private segment: their frame is explained by SGPR spills that the compiler parked in VGPR lanes
# (.sgpr_spill_count > 0, .vgpr_spill_count == 0) and their code contains no scratch / stack instruction.  Everything else with a
# non-zero .private_segment_fixed_size fails the build (an allow-list per kernel, not a global relaxation).
FRAME_ALLOWED = ("k_wskpl<", "k_wskpu<", "k_headu<")


def check(build_dir):
    objs = sorted(glob.glob(os.path.join(ROOT, "betty_amd", "csrc", build_dir, "*.o")))
    if not objs:
        sys.exit(f"check_spills: no objects under betty_amd/csrc/{build_dir} — run make first")
    bad, allk, frames = [], [], []
    with tempfile.TemporaryDirectory() as tmp:
        for o in objs:
            for name, d in kernels_of(o, tmp):
                allk.append((name, d))
                if d.get(".vgpr_spill_count", 0):
                    bad.append((name, d))
                elif d.get(".private_segment_fixed_size", 0):
                    explained = d.get(".sgpr_spill_count", 0) > 0 and short(name).startswith(FRAME_ALLOWED)
                    n = scratch_ops(o, tmp, name) if explained else None
                    if not explained or n != 0:   # (n == -1: the kernel's symbol was not found in the disassembly — refuse, do not guess)
                        d["scratch_instructions"] = n
                        bad.append((name, d))
                    else:
                        frames.append((name, d))
    return objs, bad, allk, frames


def main():
    for build_dir in ("build", "build_ab"):
        if build_dir == "build_ab" and not os.path.isdir(os.path.join(ROOT, "betty_amd", "csrc", build_dir)):
            continue
        objs, bad, allk, frames = check(build_dir)
        report(build_dir, objs, bad, allk, frames)


def report(build_dir, objs, bad, allk, frames):
    allk.sort(key=lambda t: -t[1].get(".vgpr_count", 0))
    if "--list" in sys.argv:
        for name, d in allk:
            print(d, short(name))
    print(f"check_spills [{build_dir}]: {len(allk)} kernels in {len(objs)} objects; fattest: " +
          ", ".join(f"{short(n)}={d['.vgpr_count']}" for n, d in allk[:5]))
    if bad:
        for name, d in bad:
            print("  SPILL:", short(name), d, file=sys.stderr)
        sys.exit(f"check_spills: {len(bad)} kernel(s) spill registers / use scratch")
    for name, d in frames:
        print(f"  note: {short(name)} declares a {d['.private_segment_fixed_size']}-byte frame but has no scratch instruction "
              f"(SGPR spills to VGPR lanes: {d.get('.sgpr_spill_count', 0)})")
    print(f"check_spills [{build_dir}]: no VGPR spills, no scratch")


if __name__ == "__main__":
    main()
