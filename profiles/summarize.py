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
    # per-kernel bytes per launch (largest grid of each kernel = the full-size legs) + average duration from the
    # stats pass: what bench.py copies into roofline.traffic and DESIGN.md's kernel table cites
    import json
    dur = {row["Name"]: float(row["AverageNs"]) / 1e3 for row in csv.DictReader(open(ks))}
    traffic = {}
    for d, mult, key in ((fetch, 2.0, "fetch_bytes_per_launch"), (write, 1.0, "write_bytes_per_launch")):
        by_kernel = collections.defaultdict(dict)
        for (name, _ctr, grid), vals in pmc(d).items():
            by_kernel[name][int(grid)] = vals
        for name, grids in by_kernel.items():
            g = max(grids, key=lambda k: max(grids[k]))  # the launches that move the most bytes
            top = max(grids[g])
            vals = [v for v in grids[g] if v >= 0.5 * top]  # launches enqueued past a stop / CG convergence are no-ops
            t = traffic.setdefault(name, {})
            t[key] = sum(vals) / len(vals) * 1024 * mult
            t[key.replace("bytes_per_launch", "launches_sampled")] = len(vals)
            t.setdefault("grid", g)
    for name, t in traffic.items():
        if name in dur:
            t["avg_us_all_launches"] = dur[name]
    keep = {k: v for k, v in traffic.items() if k.startswith(("void admm::", "admm::")) and
            v.get("fetch_bytes_per_launch", 0) + v.get("write_bytes_per_launch", 0) > 1e5}
    keep["_note"] = ("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, --kernel-trace only) of the FULL "
                     "bench.py --steps 20 --warmup 2 --no-cpu-baseline on MI355X; FETCH_SIZE[KB]*1024*2 (gfx950 "
                     "wide-read correction, MI355X_MICROARCH.md), WRITE_SIZE[KB]*1024; per kernel the grid with the "
                     "largest traffic (the 100000x10000 / 4096^2 legs); avg_us_all_launches from --kernel-trace --stats "
                     "mixes every grid of that kernel")
    json.dump(keep, open(os.path.join(here, f"{tag}_traffic.json"), "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
