#!/bin/bash
# usage (GPU box, repo root): tools/prof_pmc.sh <tag>
# two separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) with --kernel-trace only; prints per-kernel means
tag=$1
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
rm -rf $R/gpurun_out/pmc_${tag}_FETCH_SIZE $R/gpurun_out/pmc_${tag}_WRITE_SIZE   # (delete the locally merged copies of earlier runs with this tag too before reading new ones)
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/pmc_${tag}_$c -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --inflight 1 > $R/gpurun_out/pmc_${tag}_$c.log 2>&1
  echo pmc_${c}_exit=$?
done
python3 - $R/gpurun_out $tag <<'PY'
import csv, glob, sys, collections, json
root, tag = sys.argv[1], sys.argv[2]
res = collections.defaultdict(dict)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    fs = glob.glob(f"{root}/pmc_{tag}_{c}/**/*counter_collection.csv", recursive=True)
    acc = collections.defaultdict(list)
    for f in fs:
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") == c:
                acc[r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        res[k][c] = sum(v) / len(v)
        res[k]["launches_" + c] = len(v)
out = {}
for k, d in sorted(res.items(), key=lambda kv: -(kv[1].get("FETCH_SIZE", 0) + kv[1].get("WRITE_SIZE", 0)))[:16]:
    print("%-44s FETCH_SIZE %12.1f  WRITE_SIZE %12.1f (mean counter value per launch)" % (k[-44:], d.get("FETCH_SIZE", 0), d.get("WRITE_SIZE", 0)))
    out[k] = d
json.dump(out, open(f"{root}/pmc_{tag}_summary.json", "w"), indent=1)
PY
