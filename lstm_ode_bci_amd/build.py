"""Build liblob.so (the C-ABI library of include/lob.h) for gfx950 with hipcc, in-tree.

    python -m lstm_ode_bci_amd.build            # rebuild if any source is newer
    python -m lstm_ode_bci_amd.build --force

hipcc cross-compiles without a GPU.  The .so stays next to the package (git-ignored,
but it travels with gpurun snapshots) so the GPU box never needs a compiler run.
"""
from __future__ import annotations

import glob
import hashlib
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


FLAGS = [f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC"]
ID_FILE = os.path.join(PKG, "liblob.build_id")


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def source_id():
    """Hash of every source the library is built from plus the compile flags: what `lob_build_id()` of a fresh
    build returns.  The loader (_lib.lib) refuses a library whose id differs from the sources it can see."""
    h = hashlib.sha256(" ".join(FLAGS).encode())
    deps = sources() + sorted(glob.glob(os.path.join(CSRC, "*.h"))) + sorted(glob.glob(os.path.join(INCLUDE, "*.h")))
    for p in deps:
        h.update(os.path.basename(p).encode())
        h.update(open(p, "rb").read())
    return h.hexdigest()[:16]


def built_id():
    try:
        return open(ID_FILE).read().strip()
    except OSError:
        return None


def _stale():
    return not os.path.exists(LIB) or built_id() != source_id()


def isa_checks():
    """Static checks of the hand-synchronised kernels (tools/isa_check.py): their correctness depends on what hipcc
    does around inline-asm loads and counted waits, so a build whose ISA violates them must not ship."""
    tools = os.path.join(ROOT, "tools")
    if tools not in sys.path:
        sys.path.insert(0, tools)
    import isa_check
    return (isa_check.main() + isa_check.check_dma_gemms() + isa_check.check_gate_ws() + isa_check.check_dx_ksplit() +
            isa_check.check_gemm_pp() + isa_check.check_h256_rec())


def build(force=False, verbose=True, check_isa=True):
    if not force and not _stale():
        return LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: cannot build liblob.so")
    objdir = os.path.join(PKG, "build")
    os.makedirs(objdir, exist_ok=True)
    # objects of sources that no longer exist (a dropped kernel) must not linger next to the ones that are linked
    live = {os.path.basename(src)[:-4] + ".o" for src in sources()}
    for f in os.listdir(objdir):
        if f.endswith(".o") and f not in live:
            os.remove(os.path.join(objdir, f))
    objs = []
    procs = []
    bid = source_id()
    for src in sources():
        obj = os.path.join(objdir, os.path.basename(src)[:-4] + ".o")
        objs.append(obj)
        cmd = [hipcc] + FLAGS + [f'-DLOB_BUILD_ID="{bid}"', "-I", INCLUDE, "-I", CSRC, "-c", src, "-o", obj]
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
    if check_isa:
        problems = isa_checks()
        if problems:
            os.remove(LIB)
            raise RuntimeError("ISA checks of the hand-synchronised kernels failed:\n  " + "\n  ".join(problems))
    with open(ID_FILE, "w") as f:
        f.write(bid + "\n")
    if verbose:
        print(f"built {LIB} (build id {bid})")
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
