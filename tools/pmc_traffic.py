#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (collected SEPARATELY, as
MI355X_MICROARCH.md prescribes) into per-launch HBM traffic of the bench's hot kernels.

    python tools/pmc_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> <tag> [B] [H] [nsteps] [precision]

nsteps (steps + warm-up of a `bench.py --no-roofline --no-extra --no-cpu-baseline` pass: nothing but the steps ran under
the profiler): adds the HBM bytes of ONE WHOLE STEP (all kernels) as `step|<precision>|B<B>[|H<H>]` -- what bench.py
reports as roofline.step.

gfx950 corrections (MI355X_MICROARCH.md §HBM): FETCH_SIZE counts 64 B per 128-B request, i.e. exactly
half of the bytes of wide coalesced reads (checked here on kernels with a known byte count: the bf16
recurrent kernels read 2.1 GB of P and report 1.05 M KB), so reads = 2 x FETCH_SIZE; WRITE_SIZE is exact.
Both counters are in KB.  Writes profiles/<tag>_pmc_summary.csv and updates profiles/pmc_traffic.json,
which bench.py reads to fill roofline.traffic.  Every entry is stamped with the build id of the sources it was measured on
(`_build_id`: build.source_id() of this tree -- run the tool before touching csrc/ again); bench.py marks an entry whose
stamp differs from the library it runs as stale and then reports no whole-step figure.  The whole-step sum applies the 2x
read correction to EVERY kernel, although it was validated on wide coalesced reads only: an upper bound.
"""
import collections
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# bench label -> (precision, substring of the profiler's kernel name)
KERNELS = {
    "lstm_rec_fwd(save)": [("mixed", "lstm_rec_fwd_h128_bf16_s16_kernelILb1ELb0ELb1ELb1E"), ("fp32", "lstm_rec_fwd_h128_s16_kernel<true")],
    "lstm_rec_fwd": [("mixed", "lstm_rec_fwd_h128_bf16_s16_kernelILb0ELb0"), ("fp32", "lstm_rec_fwd_h128_split_kernel<false")],
    "lstm_rec_bwd": [("mixed", "lstm_rec_bwd_h128_bf16_s16_dma_kernel"), ("fp32", "lstm_rec_bwd_h128_kernel")],
    "gate_gemm_x(K=256)": [("mixed", "gate_gemm_ws_kernel<256, 0, 128>"), ("fp32", "gate_gemm_ws_split_kernel<256>")],
    "gemm_nt(dX)": [("mixed", "dx_ksplit_kernel<4>")],
    "lstm_dw(dW_ih+dW_hh)": [("mixed", "lstm_dw_h128_kernel<256>")],
}
# H = 256 (bench --hidden 256): the kernels of the reference's checkpoint size
KERNELS_H256 = {
    # rocprofv3 half-demangles this one (its DF16b argument defeats the demangler): both spellings
    "lstm_rec_fwd(save)": [("mixed", "lstm_rec_fwd_h256_bf16_kernelILb1ELb0ELb1ELb1E"),
                           ("mixed", "lstm_rec_fwd_h256_bf16_kernel<true, false, true, true")],
    "lstm_rec_bwd": [("mixed", "lstm_rec_bwd_h256_bf16_kernel")],
    "gate_gemm_x(K=512)": [("mixed", "gate_gemm_ws_kernel<512, 0, 256>")],
    "gemm_nt(dX)": [("mixed", "gemm_nt_pp16_kernel")],
    "lstm_dw(dW_ih+dW_hh)": [("mixed", "lstm_dw_pp_kernel<3>")],
}


def load(path, counter):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return agg


def main():
    fetch, write, tag = sys.argv[1], sys.argv[2], sys.argv[3]
    B = int(sys.argv[4]) if len(sys.argv) > 4 else 4096
    H = int(sys.argv[5]) if len(sys.argv) > 5 else 128
    nsteps = int(sys.argv[6]) if len(sys.argv) > 6 else 0
    precision = sys.argv[7] if len(sys.argv) > 7 else "mixed"
    sfx = "" if H == 128 else f"|H{H}"
    sys.path.insert(0, ROOT)
    from lstm_ode_bci_amd import build as _build
    bid = _build.source_id()
    f, w = load(fetch, "FETCH_SIZE"), load(write, "WRITE_SIZE")
    rows = []
    for k in sorted(set(f) | set(w), key=lambda k: -(sum(f.get(k, [0])) + sum(w.get(k, [0])))):
        fv, wv = f.get(k, [0.0]), w.get(k, [0.0])
        fa, wa = sum(fv) / len(fv), sum(wv) / len(wv)
        rows.append({"kernel": k, "launches": max(len(fv), len(wv)), "FETCH_SIZE_KB_avg": round(fa, 1),
                     "WRITE_SIZE_KB_avg": round(wa, 1), "hbm_bytes_per_launch_corrected": int((2 * fa + wa) * 1024)})
    out = os.path.join(ROOT, "profiles", f"{tag}_pmc_summary.csv")
    with open(out, "w", newline="") as fh:
        wr = csv.DictWriter(fh, fieldnames=list(rows[0]))
        wr.writeheader()
        wr.writerows(rows)
    jpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    table = json.load(open(jpath)) if os.path.exists(jpath) else {}
    stamps = table.setdefault("_build_id", {})
    for label, alts in (KERNELS if H == 128 else KERNELS_H256).items():
        for prec, sub in alts:
            if prec != precision:
                continue
            for r in rows:
                if sub in r["kernel"]:
                    table[f"{label}|{prec}|B{B}{sfx}"] = r["hbm_bytes_per_launch_corrected"]
                    stamps[f"{label}|{prec}|B{B}{sfx}"] = bid
                    break
    if nsteps > 0:      # every kernel of the pass, launches x bytes, over the steps that ran
        total = sum((2 * sum(f.get(k, [0.0])) + sum(w.get(k, [0.0]))) * 1024 for k in set(f) | set(w))
        table[f"step|{precision}|B{B}{sfx}"] = int(total / nsteps)
        stamps[f"step|{precision}|B{B}{sfx}"] = bid
    json.dump(table, open(jpath, "w"), indent=1, sort_keys=True)
    print(out)
    print(json.dumps(table, indent=1))


if __name__ == "__main__":
    main()
