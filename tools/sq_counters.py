#!/usr/bin/env python3
"""Aggregate one rocprofv3 --pmc pass of SQ counters (kernel-trace only) into per-kernel fractions of wave cycles.

    rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT \\
              --kernel-trace --output-format csv -d DIR -o run -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline
    python tools/sq_counters.py DIR/run_counter_collection.csv profiles/<tag>_sq_counters.csv

wait_any = waves parked in s_waitcnt / barriers; wait_inst_any = waves with an instruction that cannot issue (both, like
SQ_WAVE_CYCLES, in units of 4 cycles); SQ_VALU_MFMA_BUSY_CYCLES is in cycles and per SIMD: the printed fraction is
busy / (4 x wave cycles), so with n waves per SIMD the pipe's utilisation is n x that."""
import collections
import csv
import sys


def main():
    src, dst = sys.argv[1], sys.argv[2]
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(src)):
        agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    rows = []
    for k, c in agg.items():
        wc = sum(c.get("SQ_WAVE_CYCLES", [0.0]))
        if wc <= 0:
            continue
        n = len(c["SQ_WAVE_CYCLES"])

        def frac(name):
            return round(sum(c[name]) / wc, 4) if name in c else ""
        mf = round(sum(c["SQ_VALU_MFMA_BUSY_CYCLES"]) / (4.0 * wc), 4) if "SQ_VALU_MFMA_BUSY_CYCLES" in c else ""
        rows.append((wc, [k, n, int(wc / n), frac("SQ_WAIT_ANY"), frac("SQ_WAIT_INST_ANY"), mf,
                          frac("SQ_LDS_BANK_CONFLICT")]))
    rows.sort(key=lambda t: -t[0])
    with open(dst, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "launches", "SQ_WAVE_CYCLES_quad_per_launch", "wait_any_frac", "wait_inst_any_frac",
                    "mfma_busy_over_4x_wave_cycles", "lds_bank_conflict_over_wave_cycles"])
        for _, r in rows:
            w.writerow(r)
    for _, r in rows[:14]:
        print(r[0][:60], r[1:])


if __name__ == "__main__":
    main()
