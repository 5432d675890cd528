#!/usr/bin/env python
"""sha256 of the gfx950 device code (.hip_fatbin section) of every object under betty_amd/csrc/build, and of libbhg.so.
Two builds that differ only in host code (hoist_plan's conditions, argument plumbing) print the same device-code hashes:
how profiles/README.md ties the shipped library to the one the GPU suite ran on."""
import glob, hashlib, os, subprocess, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
with tempfile.TemporaryDirectory() as tmp:
    for o in sorted(glob.glob(os.path.join(ROOT, "betty_amd", "csrc", "build", "*.o"))):
        fat = os.path.join(tmp, "fat")
        if os.path.exists(fat):
            os.remove(fat)
        r = subprocess.run([f"{LLVM}/llvm-objcopy", "--dump-section", f".hip_fatbin={fat}", o], capture_output=True)
        if r.returncode == 0 and os.path.exists(fat) and os.path.getsize(fat):
            print(f"{os.path.basename(o):18s} device code {hashlib.sha256(open(fat, 'rb').read()).hexdigest()[:16]}")
lib = os.path.join(ROOT, "betty_amd", "csrc", "libbhg.so")
print(f"{'libbhg.so':18s} whole file  {hashlib.sha256(open(lib, 'rb').read()).hexdigest()[:16]}")
