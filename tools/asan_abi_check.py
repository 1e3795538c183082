#!/usr/bin/env python3
"""CPU-side AddressSanitizer build of the C-ABI's HOST code (the optional sanitizer pass).

GPU AddressSanitizer is not available on this pool, and the kernels are checked by the parity tests; what ASan can
check is the host layer of every entry point -- argument validation, launch-parameter arithmetic, the descriptor
tables copied into kernel arguments.  Every csrc/*.hip is compiled with `-Xarch_host -fsanitize=address` (the device
code is built as usual, uninstrumented) into a scratch library, which a child Python process loads with the ASan
runtime preloaded and drives through the argument-error paths of all entry points (each must return a negative
LOB_E_* code before any HIP call).  Exit code 0 and no ASan report = pass.

    python tools/asan_abi_check.py            # builds under a temp dir, ~2 min
"""
import glob
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "lstm_ode_bci_amd", "csrc")

CHILD = r'''
import ctypes as C, sys
L = C.CDLL(sys.argv[1])
NULL = None
bad = []
def expect_err(name, *args):
    fn = getattr(L, name)
    fn.restype = C.c_int
    rc = fn(*args)
    if rc >= 0:
        bad.append((name, rc))
i = C.c_int
f = C.c_float
u64 = C.c_uint64
rates = (C.c_double * 6)(*[0.1] * 6)
expect_err("lob_gemm_nt_f32", NULL, i(1), NULL, i(1), NULL, NULL, i(1), i(1), i(1), i(1), i(0), NULL)
expect_err("lob_gemm_tn_f32", NULL, i(1), NULL, i(1), NULL, i(1), i(1), i(1), i(1), NULL)
expect_err("lob_gate_gemm_x_f32", NULL, i(1), NULL, NULL, NULL, i(1), i(32), i(128), i(2), i(128), i(1), NULL, NULL)
expect_err("lob_lstm_rec_fwd_f32", NULL, NULL, NULL, NULL, i(1), i(32), i(128), i(2), i(0), NULL, NULL)
expect_err("lob_lstm_rec_bwd_f32", NULL, NULL, NULL, NULL, NULL, i(0), NULL, i(1), i(32), i(128), i(2), NULL)
expect_err("lob_gemm_nt_bf16", NULL, i(1), i(8), NULL, i(1), i(8), NULL, NULL, i(8), i(8), i(8), i(8), i(0), f(0.0), u64(0), NULL)
expect_err("lob_gate_gemm_x_bf16", NULL, i(1), i(8), NULL, i(1), NULL, NULL, i(1), i(1), i(32), i(128), i(2), i(128), NULL)
expect_err("lob_gemm_tn_bf16", NULL, i(1), i(8), NULL, i(1), i(8), NULL, i(8), i(8), i(8), i(8), NULL)
expect_err("lob_lstm_dw_bf16", NULL, i(8), NULL, i(8), i(256), NULL, i(8), NULL, NULL, i(2), i(32), i(128), i(2), NULL)
expect_err("lob_lstm_rec_fwd_bf16", NULL, i(1), NULL, NULL, NULL, NULL, i(0), NULL, NULL, f(0.0), u64(0), i(1), i(32), i(128), i(2), i(0), i(0), NULL)
expect_err("lob_lstm_rec_bwd_bf16", NULL, i(1), NULL, i(0), NULL, NULL, NULL, i(0), NULL, NULL, NULL, i(1), i(32), i(128), i(2), NULL)
expect_err("lob_colsum_f32", NULL, i(1), i(1), i(1), NULL, NULL)
expect_err("lob_colsum_bf16", NULL, i(1), i(1), i(1), NULL, NULL)
expect_err("lob_act_f32", NULL, NULL, C.c_int64(1), i(0), NULL)
expect_err("lob_act_bwd_f32", NULL, NULL, NULL, C.c_int64(1), i(0), NULL)
expect_err("lob_layernorm_act_f32", NULL, NULL, NULL, NULL, i(0), i(1), i(128), f(1e-5), i(0), i(0), i(0), i(0), f(0.0), u64(0), NULL)
expect_err("lob_layernorm_act_bwd_f32", NULL, NULL, NULL, NULL, NULL, NULL, NULL, i(1), i(128), f(1e-5), i(0), i(0), i(0), i(0),
           f(0.0), u64(0), NULL, NULL, i(0), i(0), i(0), NULL, NULL)
expect_err("lob_attn_pool_fwd_f32", NULL, i(0), NULL, NULL, NULL, NULL, NULL, i(1), i(1), i(1), i(8), i(4), NULL)
expect_err("lob_attn_pool_bwd_f32", NULL, i(0), NULL, NULL, NULL, NULL, NULL, NULL, i(0), NULL, NULL, i(1), i(1), i(1), i(8), i(4), NULL, NULL)
expect_err("lob_dropout_f32", NULL, NULL, C.c_int64(1), f(0.5), u64(0), NULL)
expect_err("lob_softmax_rows_f32", NULL, NULL, i(0), i(0), NULL)
expect_err("lob_ode_rk4_f64", NULL, NULL, rates, C.c_double(0.5), i(10), C.c_double(0.0), C.c_double(10.0), i(16), NULL, NULL, NULL, i(4), i(0), NULL)
expect_err("lob_prob_to_state_f64", NULL, NULL, i(1), NULL)
expect_err("lob_weighted_ce_f32", NULL, NULL, NULL, NULL, NULL, NULL, i(1), i(2), f(1.0), NULL)
expect_err("lob_sumsq_f32", NULL, C.c_int64(1), NULL, NULL, NULL)
expect_err("lob_clip_scale_f32", NULL, C.c_int64(1), NULL, f(1.0), NULL)
expect_err("lob_adamw_f32", NULL, NULL, NULL, NULL, C.c_int64(1), f(1e-3), f(0.9), f(0.999), f(1e-8), f(0.0), C.c_int64(1), NULL, f(1.0), f(1.0), NULL)
expect_err("lob_pad_cast_bf16", NULL, NULL, C.c_int64(1), i(61), i(64), NULL)
expect_err("lob_abs_colsum_f32", NULL, C.c_int64(1), i(61), f(1.0), NULL, NULL)
expect_err("lob_prep_weights", NULL, i(1), NULL)
# a full descriptor table with one bad entry: the table is copied into the kernel-argument struct before validation ends
class PrepOp(C.Structure):
    _fields_ = [("src", C.c_void_p), ("src2", C.c_void_p), ("dst", C.c_void_p), ("rows", C.c_int), ("cols", C.c_int),
                ("ld_src", C.c_int), ("ld_dst", C.c_int), ("pad_to", C.c_int), ("kind", C.c_int), ("blk0", C.c_int),
                ("reserved", C.c_int)]
ops = (PrepOp * 64)()
for k in range(64):
    ops[k] = PrepOp(0x1000, 0, 0x2000, 4, 4, 4, 4, 0, 0, 0, 0)
ops[63].ld_src = 1                                   # ld_src < cols
expect_err("lob_prep_weights", C.cast(ops, C.c_void_p), i(64), NULL)
expect_err("lob_prep_weights", C.cast(ops, C.c_void_p), i(65), NULL)
L.lob_version.restype = C.c_int
assert L.lob_version() >= 200
L.lob_debug_set_variant.restype = C.c_int
assert L.lob_debug_set_variant(i(999), i(1)) < 0 and L.lob_debug_set_variant(i(-1), i(1)) < 0
if bad:
    print("entry points that did not reject bad arguments:", bad)
    sys.exit(2)
print("asan_abi_check: ok")
'''


def main():
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    clang = os.path.join(os.path.dirname(os.path.realpath(hipcc)), "..", "lib", "llvm", "bin", "clang")
    if not os.path.exists(clang):
        clang = "/opt/rocm/lib/llvm/bin/clang"
    rt = subprocess.run([clang, "-print-file-name=libclang_rt.asan-x86_64.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.exists(rt):
        print("ASan runtime not found: skipped")
        return 77
    tmp = tempfile.mkdtemp(prefix="lob_asan_")
    objs, procs = [], []
    for src in sorted(glob.glob(os.path.join(CSRC, "*.hip"))):
        obj = os.path.join(tmp, os.path.basename(src)[:-4] + ".o")
        objs.append(obj)
        cmd = [hipcc, "--offload-arch=gfx950", "-O1", "-g", "-Xarch_host", "-fsanitize=address", "-Xarch_host",
               "-fno-omit-frame-pointer", "-std=c++17", "-fPIC",
               '-DLOB_BUILD_ID="asan"', "-I", os.path.join(ROOT, "include"), "-I", CSRC, "-c", src, "-o", obj]
        procs.append((cmd, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for cmd, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            print("compile failed:", " ".join(cmd), "\n", out)
            return 1
    lib = os.path.join(tmp, "liblob_asan.so")
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-fsanitize=address", "-o", lib] + objs,
                       capture_output=True, text=True)
    if r.returncode != 0:
        print("link failed:", r.stdout, r.stderr)
        return 1
    child = os.path.join(tmp, "child.py")
    open(child, "w").write(CHILD)
    env = dict(os.environ, LD_PRELOAD=rt, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:exitcode=23")
    r = subprocess.run([sys.executable, child, lib], env=env, capture_output=True, text=True, timeout=300)
    sys.stdout.write(r.stdout)
    if r.returncode != 0 or "ERROR: AddressSanitizer" in r.stderr:
        sys.stdout.write(r.stderr[-4000:])
        return 1
    shutil.rmtree(tmp, ignore_errors=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
