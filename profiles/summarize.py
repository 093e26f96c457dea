#!/usr/bin/env python3
"""Summarise rocprofv3 output directories (gpurun_out/prof_*) into profiles/<tag>_*.{csv,md}.

    python profiles/summarize.py r1a gpurun_out/prof_r1_stats gpurun_out/prof_r1_fetch gpurun_out/prof_r1_write \
        [more_fetch_dir more_write_dir ...]

stats dir : --kernel-trace --stats        -> per-kernel calls / average ns  (copied verbatim)
fetch dir : --pmc FETCH_SIZE --kernel-trace -> KB per dispatch; on gfx950 FETCH_SIZE counts 1/2 of a
            wide (16 B/lane) coalesced read stream (MI355X_MICROARCH.md, HBM) -> bytes = KB*1024*2
write dir : --pmc WRITE_SIZE --kernel-trace -> KB per dispatch, exact for 16 B/lane streaming stores
"""
import collections
import csv
import glob
import os
import shutil
import sys


def pmc(dirname):
    files = glob.glob(os.path.join(dirname, "**", "*_counter_collection.csv"), recursive=True)
    agg = collections.defaultdict(list)
    for f in files:
        for row in csv.DictReader(open(f)):
            agg[(row["Kernel_Name"], row["Counter_Name"], row["Grid_Size"])].append(float(row["Counter_Value"]))
    return agg


def main():
    tag, stats, fetch, write = sys.argv[1:5]
    more = sys.argv[5:]  # further (fetch, write) directory pairs, e.g. the TV scripts
    here = os.path.dirname(os.path.abspath(__file__))
    ks = glob.glob(os.path.join(stats, "**", "*_kernel_stats.csv"), recursive=True)[0]
    shutil.copy(ks, os.path.join(here, f"{tag}_kernel_stats.csv"))
    lines = [f"# rocprofv3 summary {tag}", "", "## kernel-trace --stats (top kernels)", "",
             "| kernel | calls | avg us | total ms | % |", "|---|---|---|---|---|"]
    for row in list(csv.DictReader(open(ks)))[:14]:
        lines.append(f"| `{row['Name'][:90]}` | {row['Calls']} | {float(row['AverageNs']) / 1e3:.1f} | "
                     f"{float(row['TotalDurationNs']) / 1e6:.2f} | {float(row['Percentage']):.2f} |")
    lines += ["", "## HBM traffic per dispatch (PMC, separate passes)", "",
              "FETCH bytes = FETCH_SIZE[KB] x 1024 x 2 (gfx950 wide-read correction); WRITE bytes = WRITE_SIZE[KB] x 1024.",
              "", "| kernel | grid | dispatches | counter | avg MB per dispatch |", "|---|---|---|---|---|"]
    passes = [(fetch, 2.0), (write, 1.0)] + [(d, 2.0 if k % 2 == 0 else 1.0) for k, d in enumerate(more)]
    for d, mult in passes:
        for (name, ctr, grid), vals in sorted(pmc(d).items(), key=lambda kv: -sum(kv[1]))[:12]:
            lines.append(f"| `{name[:70]}` | {grid} | {len(vals)} | {ctr} | "
                         f"{sum(vals) / len(vals) * 1024 * mult / 1e6:.2f} |")
    open(os.path.join(here, f"{tag}_summary.md"), "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
