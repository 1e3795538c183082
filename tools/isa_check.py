#!/usr/bin/env python3
"""Static checks on the compiled ISA of the hand-synchronised BPTT kernel (lstm_rec_bwd_h128_bf16_s16_dma_kernel).

That kernel issues its dY loads and reads its LDS-DMA ring through inline asm and counts `s_waitcnt vmcnt(N)` by hand,
so two compiler behaviours would break it silently or slow it down:

1. the registers written by the hand-issued `global_load_dword`s must not be touched (copied, spilled, read) by any
   compiler-generated instruction before the `s_waitcnt vmcnt(18)` that retires them, two half-iterations later;
2. the steady-state loop must contain no `s_waitcnt vmcnt(0)`, no scratch (spill) traffic and no other VMEM wait than
   the hand-written one -- any of these drains the DMA queue every step.

    python tools/isa_check.py            # compiles csrc/lstm_rec_bf16_s16.hip to ISA (hipcc, ~15 s) and checks it
"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "lstm_ode_bci_amd", "csrc", "lstm_rec_bf16_s16.hip")
KERNEL = r"lstm_rec_bwd_h128_bf16_s16_dma_kernelILi%dELb%dELb%dEE"


def compile_to_isa(path=None):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    out = path or os.path.join(tempfile.mkdtemp(prefix="lob_isa_"), "s16.s")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I", os.path.join(ROOT, "include"),
           "-I", os.path.dirname(SRC), "-S", "--cuda-device-only", SRC, "-o", out]
    subprocess.run(cmd, check=True, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    return out


def check_dma_gemms(path=None):
    """The LDS-DMA TN GEMMs (gemm_tn_dma_kernel, lstm_dw_h128_kernel in gemm_bf16.hip) read their ring through inline
    asm because hipcc puts `s_waitcnt vmcnt(0)` in front of every compiler-visible ds_read_b64_tr_b16 of a DMA target
    (that wait covers the DMA issued a moment earlier: the ring degenerates to one tile in flight).  Check that no
    compiler-generated vmcnt(0) and no scratch traffic is left in those kernels."""
    src = os.path.join(ROOT, "lstm_ode_bci_amd", "csrc", "gemm_bf16.hip")
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    out = path or os.path.join(tempfile.mkdtemp(prefix="lob_isa_"), "gemm.s")
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I", os.path.join(ROOT, "include"),
                    "-I", os.path.dirname(src), "-S", "--cuda-device-only", src, "-o", out],
                   check=True, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    lines = open(out).read().split("\n")
    problems = []
    for pat in ("gemm_tn_dma_kernelILi256ELi256E", "lstm_dw_h128_kernelILi256E", "lstm_dw_h128_kernelILi128E"):
        body = _function(lines, pat)
        in_asm = False
        for i, l in enumerate(body):
            if "#ASMSTART" in l:
                in_asm = True
            elif "#ASMEND" in l:
                in_asm = False
            elif not in_asm and "s_waitcnt" in l and "vmcnt(0)" in l:
                problems.append(f"{pat}: compiler-generated vmcnt(0) at line {i}")
            elif "scratch_" in l:
                problems.append(f"{pat}: scratch access at line {i}")
        if not any("global_load_lds" in l for l in body):
            problems.append(f"{pat}: no LDS-DMA found (kernel changed?)")
    return problems


def _check_ws_kernel(srcname, cases, path=None):
    """Shared check of the weight-stationary streaming GEMMs: in the tile loop exactly the LDS-DMA instructions, fragment
    stores and MFMAs the hand-counted `s_waitcnt vmcnt(N)` assumes, no compiler-generated VMEM wait, no spill traffic,
    no register loads (weights re-loaded)."""
    src = os.path.join(ROOT, "lstm_ode_bci_amd", "csrc", srcname)
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    out = path or os.path.join(tempfile.mkdtemp(prefix="lob_isa_"), "ws.s")
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I", os.path.join(ROOT, "include"),
                    "-I", os.path.dirname(src), "-S", "--cuda-device-only", src, "-o", out],
                   check=True, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    lines = open(out).read().split("\n")
    problems = []
    for pat, ndma, nst, nmfma in cases:
        body = _function(lines, pat)
        ins = _instrs(body)
        # the tile loop: the annotated inner loop that holds the MFMAs
        loops = [(a, b) for a, b in _inner_loops(body) if any(x[0] >= a and x[0] <= b and x[1].startswith("v_mfma") for x in ins)]
        if len(loops) != 1:
            problems.append(f"{pat}: tile loop not found")
            continue
        loop = [x for x in ins if loops[0][0] <= x[0] <= loops[0][1]]
        n_dma = sum(1 for _, t, _ in loop if t.startswith("global_load_lds"))
        n_st = sum(1 for _, t, _ in loop if t.startswith("global_store"))
        n_mfma = sum(1 for _, t, _ in loop if t.startswith("v_mfma"))
        if n_dma != ndma or n_st != nst or n_mfma != nmfma:
            problems.append(f"{pat}: tile loop has {n_dma} DMA / {n_st} stores / {n_mfma} MFMAs, the wait counts assume "
                            f"{ndma} / {nst} / {nmfma}")
        for _, t, a in loop:
            if not a and re.search(r"s_waitcnt.*vmcnt\(", t):
                problems.append(f"{pat}: compiler-generated VMEM wait in the tile loop: {t}")
            if t.startswith("scratch_"):
                problems.append(f"{pat}: spill traffic in the tile loop: {t}")
            if not a and t.startswith("global_load") and not t.startswith("global_load_lds"):
                problems.append(f"{pat}: register load in the tile loop (weights re-loaded?): {t}")
    return problems


def check_gate_ws(path=None):
    """gate_gemm_ws.hip (mixed path) and gate_gemm_ws_split.hip (fp32 path): one hand-counted ring wait per row tile."""
    return (_check_ws_kernel("gate_gemm_ws.hip", [("gate_gemm_ws_kernelILi256ELi0ELi128E", 4, 8, 64),
                                                  ("gate_gemm_ws_kernelILi128ELi0ELi128E", 2, 8, 32),
                                                  ("gate_gemm_ws_kernelILi256ELi0ELi256E", 4, 8, 64),
                                                  ("gate_gemm_ws_kernelILi512ELi0ELi256E", 4, 2, 32),
                                                  ("gate_gemm_ws_kernelILi256ELi1ELi128E", 4, 8, 32),
                                                  ("gate_gemm_ws_kernelILi128ELi1ELi128E", 2, 8, 16),
                                                  ("gate_gemm_ws_kernelILi256ELi2ELi128E", 4, 8, 32),
                                                  ("gate_gemm_ws_kernelILi128ELi2ELi128E", 2, 8, 16)], path) +
            _check_ws_kernel("gate_gemm_ws_split.hip", [("gate_gemm_ws_split_kernelILi256E", 4, 4, 48),
                                                        ("gate_gemm_ws_split_kernelILi128E", 2, 4, 24)], path))


def check_dx_ksplit(path=None):
    """dx_ksplit.hip: the A-fragment loads are hand-issued (inline asm) three tiles ahead and retired by one counted
    `s_waitcnt vmcnt(3 KS [+ 3])` per tile.  Inside the tile loop: no compiler-generated VMEM wait, no spill traffic,
    exactly one store per tile body, and the registers of the in-flight loads are touched by nothing but the asm
    statements and the MFMAs that consume them (a compiler copy or spill would read them before the data lands)."""
    src = os.path.join(ROOT, "lstm_ode_bci_amd", "csrc", "dx_ksplit.hip")
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    out = path or os.path.join(tempfile.mkdtemp(prefix="lob_isa_"), "dx.s")
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I", os.path.join(ROOT, "include"),
                    "-I", os.path.dirname(src), "-S", "--cuda-device-only", src, "-o", out],
                   check=True, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    lines = open(out).read().split("\n")
    problems = []
    for KS in (4, 2):
        pat = "dx_ksplit_kernelILi%dE" % KS
        body = _function(lines, pat)
        ins = _instrs(body)
        # the steady-state tile loop = the annotated inner loop that holds MFMAs
        loops = [(x, y) for x, y in _inner_loops(body) if any(x <= z[0] <= y and z[1].startswith("v_mfma") for z in ins)]
        if len(loops) != 1:
            problems.append(f"{pat}: tile loop not found")
            continue
        tgt, last_line = loops[0]
        kend = max(k for k, z in enumerate(ins) if z[0] <= last_line)
        loop = [z for z in ins[:kend + 1] if z[0] >= tgt]
        waits = [t for _, t, a in loop if a and t.startswith("s_waitcnt vmcnt(")]
        if set(waits) != {f"s_waitcnt vmcnt({3 * KS + 3})"}:
            problems.append(f"{pat}: hand-written waits in the loop {sorted(set(waits))}, expected vmcnt({3 * KS + 3})")
        for _, t, a in loop:
            if not a and re.search(r"s_waitcnt.*vmcnt\(", t):
                problems.append(f"{pat}: compiler-generated VMEM wait in the tile loop: {t}")
            if t.startswith("scratch_"):
                problems.append(f"{pat}: spill traffic in the tile loop: {t}")
            if not a and t.startswith("global_load"):
                problems.append(f"{pat}: compiler-visible load in the tile loop: {t}")
        # every hand-issued load group: its destination registers must not be touched by compiler code until the wait
        # that retires it -- the THIRD hand-written wait after it (the loads run three tiles ahead); the three groups of
        # the prologue are retired by the first, second and third wait.  The walk is cyclic inside the loop.
        ls = next(k for k, x in enumerate(ins) if x[0] >= tgt)
        n_pro = 0

        def is_ld(k):
            return ins[k][2] and ins[k][1].startswith("global_load_dwordx4")
        k = 0
        while k <= kend:
            if not is_ld(k) or (k and is_ld(k - 1)):
                k += 1
                continue
            regs, j = set(), k
            while is_ld(j):
                regs |= _regs(ins[j][1].split(",")[0])
                j += 1
            before_first_wait = not any(x[2] and x[1].startswith("s_waitcnt vmcnt(") and "vmcnt(0)" not in x[1]
                                        for x in ins[:k])
            need = (n_pro + 1) if before_first_wait else 3
            if before_first_wait:
                n_pro += 1
            seen, pos, steps = 0, j, 0
            while seen < need and steps < 4 * len(ins):
                _, t, a = ins[pos]
                if a and t.startswith("s_waitcnt vmcnt(") and "vmcnt(0)" not in t:
                    seen += 1
                elif not a and (_regs(t) & regs):
                    problems.append(f"{pat}: register of an in-flight hand-issued load touched before its wait: {t}")
                pos = ls if pos == kend else pos + 1
                steps += 1
            k = j
    return problems


def check_gemm_pp(path=None):
    """gemm_pp.hip (the 8-wave ping-pong GEMMs of the H = 256 step): two wave groups run the same stream one barrier
    apart, so between two s_barriers a wave must do EITHER its 8 MFMAs OR its fragment reads + exactly 2 LDS-DMA
    instructions -- an MFMA that hipcc moved into a read segment would collide with the other group's matrix segment,
    and the one hand-counted `s_waitcnt vmcnt(N)` per k-tile (N = 4: NT, 6: TN) assumes 8 DMA instructions per k-tile and
    no other VMEM wait, spill or register load inside the loop."""
    src = os.path.join(ROOT, "lstm_ode_bci_amd", "csrc", "gemm_pp.hip")
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    out = path or os.path.join(tempfile.mkdtemp(prefix="lob_isa_"), "pp.s")
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I", os.path.join(ROOT, "include"),
                    "-I", os.path.dirname(src), "-S", "--cuda-device-only", src, "-o", out],
                   check=True, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    lines = open(out).read().split("\n")
    problems = []
    for pat, vm in (("gemm_nt_pp_kernelILi0ELi0ELi0EE", 4), ("gemm_nt_pp_kernelILi1ELi0ELi0EE", 4), ("gemm_tn_pp_kernel", 6)):
        body = _function(lines, pat)
        ins = _instrs(body)
        loops = [(a, b) for a, b in _inner_loops(body) if any(a <= x[0] <= b and x[1].startswith("v_mfma") for x in ins)]
        if len(loops) != 1:
            problems.append(f"{pat}: k-tile loop not found")
            continue
        loop = [x for x in ins if loops[0][0] <= x[0] <= loops[0][1]]
        waits = [t for _, t, a in loop if a and t.startswith("s_waitcnt vmcnt(")]
        if set(waits) != {f"s_waitcnt vmcnt({vm})"}:
            problems.append(f"{pat}: hand-written VMEM waits in the loop {sorted(set(waits))}, expected vmcnt({vm})")
        for _, t, a in loop:
            if not a and re.search(r"s_waitcnt.*vmcnt\(", t):
                problems.append(f"{pat}: compiler-generated VMEM wait in the loop: {t}")
            if t.startswith("scratch_"):
                problems.append(f"{pat}: spill traffic in the loop: {t}")
            if not a and t.startswith("global_load") and not t.startswith("global_load_lds"):
                problems.append(f"{pat}: register load in the loop: {t}")
        # segments between barriers, from the first barrier of the loop body that is followed by MFMAs
        segs, cur = [], {"mfma": 0, "dma": 0, "rd": 0, "st": 0}
        for _, t, a in loop:
            if t.startswith("s_barrier"):
                segs.append(cur)
                cur = {"mfma": 0, "dma": 0, "rd": 0, "st": 0}
            elif t.startswith("v_mfma"):
                cur["mfma"] += 1
            elif t.startswith("global_load_lds"):
                cur["dma"] += 1
            elif t.startswith("ds_read"):
                cur["rd"] += 1
            elif t.startswith("global_store") or t.startswith("global_atomic"):
                cur["st"] += 1
        segs.append(cur)
        hot = [sg for sg in segs if not sg["st"]]          # the epilogue's stores sit in segments of their own
        n_m = sum(1 for sg in hot if sg["mfma"])
        for sg in hot:
            if sg["mfma"] and (sg["mfma"] != 8 or sg["dma"] or sg["rd"]):
                problems.append(f"{pat}: a matrix segment holds {sg}")
            if not sg["mfma"] and sg["dma"] not in (0, 2):
                problems.append(f"{pat}: a read segment issues {sg['dma']} DMA instructions (2 expected)")
        if n_m == 0 or n_m % 4:
            problems.append(f"{pat}: {n_m} matrix segments in the loop body (a multiple of 4 expected)")
        n_dma = sum(sg["dma"] for sg in hot)
        if n_dma != 2 * n_m:
            problems.append(f"{pat}: {n_dma} DMA instructions for {n_m} matrix segments (2 per segment expected)")
    return problems


def check_h256_rec(path=None):
    """The H = 256 recurrent kernels (lstm_rec_h256_bf16.hip) sit at 244-256 registers; their schedule depends on what
    must stay true of the compiled code (round 4):
    * no scratch (spill) traffic in any instantiation -- a spill reload is a vector-memory operation in the same in-order
      queue as the W stream;
    * product forward (NQL = 3): between the step's first and last MFMA only W-fragment loads are issued (16-B loads; the
      P loads of the next step -- also 16 B -- come after the last MFMA), and the ring's waits in that stretch are counted
      (`vmcnt(N)`, N >= 6), never `vmcnt(0)`;
    * product BPTT (bf16 cell states): no branch between the last MFMA of a step and its dP stores (a block boundary
      there cost ~25 registers and spilled)."""
    src = os.path.join(ROOT, "lstm_ode_bci_amd", "csrc", "lstm_rec_h256_bf16.hip")
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    out = path or os.path.join(tempfile.mkdtemp(prefix="lob_isa_"), "h256.s")
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I", os.path.join(ROOT, "include"),
                    "-I", os.path.dirname(src), "-S", "--cuda-device-only", src, "-o", out],
                   check=True, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    text = open(out).read()
    lines = text.split("\n")
    problems = []
    names = sorted(set(re.findall(r"^(_ZN\S*lstm_rec_(?:fwd|bwd)_h256_bf16_kernel\S*):", text, re.M)))
    if len(names) < 20:
        problems.append(f"lstm_rec_h256_bf16: only {len(names)} kernels found (file changed?)")
    for n in names:
        body = _function(lines, re.escape(n[3:]))
        if any("scratch_" in l for l in body):
            problems.append(f"{n}: scratch (spill) traffic")
    # product forward: saving, bf16 outputs, dropout, bf16 cell states, NQL = 3
    for pat in ("lstm_rec_fwd_h256_bf16_kernelILb1ELb0ELb1ELb1EDF16bLi3ELi1E", "lstm_rec_fwd_h256_bf16_kernelILb0ELb0ELb1ELb0EfLi3ELi1E"):
        body = [l.strip() for l in _function(lines, pat)]
        mf = [i for i, l in enumerate(body) if l.startswith("v_mfma")]
        if len(mf) != 64:
            problems.append(f"{pat}: {len(mf)} MFMAs in the step, expected 64")
            continue
        seg = body[mf[0]:mf[-1] + 1]
        nload = sum(1 for l in seg if l.startswith("global_load_dwordx4"))
        if nload != 32:
            problems.append(f"{pat}: {nload} 16-byte loads between the first and the last MFMA, expected the 32 streamed W fragments")
        if any(l.startswith(("global_store", "global_load_ushort", "global_load_dword ")) for l in seg):
            problems.append(f"{pat}: an HBM operation inside the MFMA loop (in-order retirement stalls the W stream)")
        for l in seg:
            m = re.match(r"s_waitcnt.*vmcnt\((\d+)\)", l)
            if m and int(m.group(1)) < 6:
                problems.append(f"{pat}: '{l}' inside the MFMA loop (the ring drains)")
        after = body[mf[-1] + 1:mf[-1] + 60]
        if not any(l.startswith("global_load_dwordx4") for l in after):
            problems.append(f"{pat}: no P load right after the MFMA loop")
    pat = "lstm_rec_bwd_h256_bf16_kernelIDF16bDF16bLi2ELi1E"
    body = [l.strip() for l in _function(lines, pat)]
    mf = [i for i, l in enumerate(body) if l.startswith("v_mfma")]
    st = [i for i, l in enumerate(body) if l.startswith("global_store_dwordx4")]
    if len(mf) != 64 or not st:
        problems.append(f"{pat}: {len(mf)} MFMAs / {len(st)} dP stores (kernel changed?)")
    else:
        first_st = min(i for i in st if i > mf[-1])
        if any(l.startswith(("s_cbranch", "s_branch")) for l in body[mf[-1]:first_st]):
            problems.append(f"{pat}: a branch between the MFMA loop and the dP stores")
    return problems


def _inner_loops(body):
    """[(first_line, last_line)] of every loop hipcc annotates: from the first block tagged with the loop's header to the
    last branch that targets one of the loop's own labels."""
    lab = [(i, l) for i, l in enumerate(body) if re.match(r"^\.LBB\d+_\d+:", l)]
    out = []
    for i, l in lab:
        if "Inner Loop Header" not in l:
            continue
        tag = l.split(":")[0].lstrip(".L")
        inloop = {m.split(":")[0]: j for j, m in lab if j == i or re.search(rf"Header={tag}\b", m)}
        first = min(inloop.values())
        last = first
        for j, t in enumerate(body):
            m = re.search(r"s_c?branch\S*\s+(\.LBB\d+_\d+)", t)
            if m and m.group(1) in inloop and j >= first:
                last = max(last, j)
        out.append((first, last))
    return out


def _function(lines, pat):
    st = next(i for i, l in enumerate(lines) if re.match(r"^_ZN.*" + pat + r".*:", l))
    end = next(i for i in range(st + 1, len(lines)) if lines[i].startswith(".Lfunc_end"))
    return lines[st:end]


def _instrs(body):
    """[(index, text, in_asm)] of real instructions."""
    out, in_asm = [], False
    for i, l in enumerate(body):
        t = l.strip()
        if t.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if t.startswith(";;#ASMEND"):
            in_asm = False
            continue
        t = t.split(";")[0].strip()
        if not t or t.endswith(":") or t.startswith("."):
            continue
        out.append((i, t, in_asm))
    return out


def _regs(text):
    """VGPR numbers named by an instruction's operands."""
    regs = set()
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]", text):
        regs.update(range(int(m.group(1)), int(m.group(2)) + 1))
    for m in re.finditer(r"\bv(\d+)\b", text):
        regs.add(int(m.group(1)))
    return regs


def check_kernel(body, vm_loop=18):
    ins = _instrs(body)
    # the steady-state loop: from the first hand-written vmcnt(18) to the last backward branch after it
    labels = {m.group(1): i for i, l in enumerate(body) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m}
    waits = [k for k, (_, t, a) in enumerate(ins) if a and t.startswith(f"s_waitcnt vmcnt({vm_loop})")]
    assert len(waits) == 2, f"expected the two unrolled half-iterations, found {len(waits)} hand-written waits"
    back = [k for k, (i, t, _) in enumerate(ins)
            for m in [re.search(r"s_c?branch\S*\s+(\.LBB\d+_\d+)", t)] if m and labels.get(m.group(1), 1 << 30) < i]
    loop_end = max(k for k in back if k > waits[1])
    loop_start = min(k for k, (i, _, _) in enumerate(ins)
                     if any(labels[m.group(1)] <= i for kk in back if kk >= waits[1]
                            for m in [re.search(r"(\.LBB\d+_\d+)", ins[kk][1])] if m))
    loop = ins[loop_start:loop_end + 1]
    problems = []
    for _, t, a in loop:
        if not a and re.search(r"s_waitcnt.*vmcnt\(", t):
            problems.append(f"compiler-generated VMEM wait in the loop: {t}")
        if t.startswith("scratch_"):
            problems.append(f"spill traffic in the loop: {t}")
    # hazard 1: walk the loop cyclically from each group of hand-issued loads to the SECOND following hand wait
    n = len(loop)
    def is_dy(t):
        return t.startswith("global_load_dword ") or t.startswith("global_load_ushort ")
    starts = [k for k, (_, t, a) in enumerate(loop) if a and is_dy(t)
              and not (k and loop[k - 1][2] and is_dy(loop[k - 1][1]))]
    assert len(starts) == 2, f"expected two groups of hand-issued dY loads, found {len(starts)}"
    for s in starts:
        dst, k = set(), s
        while loop[k][2] and is_dy(loop[k][1]):
            dst |= _regs(loop[k][1].split(",")[0])
            k += 1
        seen_waits = 0
        for step in range(n):
            _, t, a = loop[(k + step) % n]
            if a and t.startswith(f"s_waitcnt vmcnt({vm_loop})"):
                seen_waits += 1
                if seen_waits == 2:
                    break
                continue
            if not a and (_regs(t) & dst):
                problems.append(f"register of an in-flight hand-issued load touched before its wait: {t}")
    return problems


def main(path=None):
    lines = open(path or compile_to_isa()).read().split("\n")
    problems = []
    for D in (1, 2):
        for y16 in (0, 1):
            for c16 in (0, 1):     # bf16 cell states: one DMA instruction fewer per step -> vmcnt(17)
                problems += [f"D={D} dy_bf16={y16} c_bf16={c16}: {p}"
                             for p in check_kernel(_function(lines, KERNEL % (D, y16, c16)), 17 if c16 else 18)]
    return problems


if __name__ == "__main__":
    probs = (main(sys.argv[1] if len(sys.argv) > 1 else None) + check_dma_gemms() + check_gate_ws() + check_dx_ksplit() +
             check_gemm_pp())
    print("\n".join(probs) if probs else "isa_check: ok")
    sys.exit(1 if probs else 0)
