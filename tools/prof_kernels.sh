#!/bin/bash
# usage (on the GPU box, from the repo root): tools/prof_kernels.sh <tag> [bench args...]
# runs bench.py under rocprofv3 --kernel-trace --stats and prints the per-kernel table
tag=$1; shift
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
rm -rf $R/gpurun_out/prof_$tag   # (delete the locally merged copies of earlier runs with this tag too before reading new ones)
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$tag -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extras --inflight 1 "$@" > $R/gpurun_out/prof_$tag.log 2>&1
echo prof_exit=$?
f=$(find $R/gpurun_out/prof_$tag -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:22]:
    print("%-60s calls %4s avg_us %10.1f  pct %5s" % (r["Name"].replace("(anonymous namespace)::", "").split("(")[0][-60:], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
