#!/usr/bin/env python3
"""No device function that is not a kernel may exceed the reach of s_cbranch (+-128 KB).

Why (profiles/r03_nested_notes.txt): past that size LLVM's branch relaxation expands far branches to
s_getpc_b64 / s_setpc_b64 through a scavenged SGPR pair, and in a leaf device function of this toolchain (ROCm 7.2, gfx950) the pair is
s[30:31] -- the return address.  The function returns into itself with a stale EXEC mask and the wave dies with
HSA_STATUS_ERROR_MEMORY_APERTURE_VIOLATION.  Kernels have no return address and are exempt.

usage: check_leaf_functions.py [object or shared library ...]   (default: build/core/*.o)   exit code 1 on a violation
"""
import glob
import os
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
LIMIT = 128*1024


def device_functions(path):
    """(name, size, is_kernel) of every function in the gfx950 code object embedded in a host object / shared library."""
    with tempfile.TemporaryDirectory() as tmp:
        fat, co = os.path.join(tmp, "fat.bin"), os.path.join(tmp, "k.co")
        if subprocess.run([f"{LLVM}/llvm-objcopy", "--dump-section", f".hip_fatbin={fat}", path], capture_output=True).returncode != 0:
            return []
        listing = subprocess.run([f"{LLVM}/clang-offload-bundler", "--list", "--type=o", f"--input={fat}"], capture_output=True, text=True).stdout.split()
        out = []
        for target in (t for t in listing if "gfx950" in t):
            subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={fat}", f"--targets={target}", f"--output={co}"], check=True)
            syms = subprocess.run([f"{LLVM}/llvm-readelf", "-s", "--wide", co], capture_output=True, text=True, check=True).stdout.splitlines()
            rows = [l.split() for l in syms if len(l.split()) >= 8]
            descriptors = {r[7][:-3] for r in rows if r[3] == "OBJECT" and r[7].endswith(".kd")}
            seen = set()
            for r in rows:
                if r[3] == "FUNC" and r[7] not in seen:
                    seen.add(r[7])
                    out.append((r[7], int(r[2]), r[7] in descriptors))
        return out


def main():
    paths = sys.argv[1:] or sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "build", "core", "*.o")))
    bad = 0
    for p in paths:
        for name, size, kernel in device_functions(p):
            if not kernel:
                verdict = "TOO LARGE" if size >= LIMIT else "ok"
                print(f"{os.path.basename(p):32s} {size:8d} B  {verdict:9s} {name[:110]}")
                bad += size >= LIMIT
    print(f"{bad} device function(s) at or above {LIMIT} bytes")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
