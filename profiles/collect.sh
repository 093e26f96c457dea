#!/bin/bash
# Collects the evidence of one round on the GPU box (run from the repo root through gpurun):
#   bash profiles/collect.sh r1k
# 1. the default bench line, 2. rocprofv3 --kernel-trace --stats of the same command, 3. separate PMC passes
# (FETCH_SIZE / WRITE_SIZE, kernel-trace only) of the FULL bench (every leg: headline, objevals, lad, trsv, cg,
# TV, SVM, consensus) -- r1 had to pass --no-extras here because the counter mode crashed on long unsynchronised
# launch runs; the loops now poll after every batch (engine_run.hip) and the full bench profiles cleanly.
set -o pipefail
tag=${1:-rXX}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
mkdir -p "$out"
cd "$root" || exit 1
timeout -k 10 400 python bench.py > "$out/${tag}_bench.json" 2> "$out/${tag}_bench.err" || exit 1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/prof_${tag}_stats" -- \
  python3 "$root/bench.py" > "$out/prof_${tag}_stats.log" 2>&1 || exit 1
for ctr in FETCH_SIZE WRITE_SIZE; do
  lc=$(echo $ctr | cut -d_ -f1 | tr 'A-Z' 'a-z')
  timeout -k 10 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d "$out/prof_${tag}_${lc}" -- \
    python3 "$root/bench.py" --steps 20 --warmup 2 --no-cpu-baseline > "$out/prof_${tag}_${lc}.log" 2>&1 || exit 1
done
echo collected
