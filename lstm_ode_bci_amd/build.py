"""Build liblob.so (the C-ABI library of include/lob.h) for gfx950 with hipcc, in-tree.

    python -m lstm_ode_bci_amd.build            # rebuild if any source is newer
    python -m lstm_ode_bci_amd.build --force

hipcc cross-compiles without a GPU.  The .so stays next to the package (git-ignored,
but it travels with gpurun snapshots) so the GPU box never needs a compiler run.
"""
from __future__ import annotations

import glob
import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
INCLUDE = os.path.join(ROOT, "include")
LIB = os.path.join(PKG, "liblob.so")
ARCH = "gfx950"


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = sources() + glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(INCLUDE, "*.h"))
    return any(os.path.getmtime(p) > t for p in deps)


def build(force=False, verbose=True):
    if not force and not _stale():
        return LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: cannot build liblob.so")
    objdir = os.path.join(PKG, "build")
    os.makedirs(objdir, exist_ok=True)
    objs = []
    procs = []
    for src in sources():
        obj = os.path.join(objdir, os.path.basename(src)[:-4] + ".o")
        objs.append(obj)
        cmd = [hipcc, f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-I", INCLUDE, "-I", CSRC,
               "-c", src, "-o", obj]
        procs.append((cmd, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for cmd, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError("hipcc failed: " + " ".join(cmd) + "\n" + out)
        if verbose and out.strip():
            print(out)
    cmd = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIB] + objs
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed: " + r.stdout)
    if verbose:
        print(f"built {LIB}")
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
