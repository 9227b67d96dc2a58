#!/usr/bin/env python3
"""Turn what tools/prof_kernels.sh / tools/prof_pmc.sh / bench.py left under gpurun_out/ into the tracked files of profiles/.

usage: python tools/collect_profiles.py <tag> [<round>]     (e.g. r01 r01)
Reads   gpurun_out/bench_default.json, gpurun_out/bench_serial.json, gpurun_out/prof_<tag>/**/kernel_stats.csv,
        gpurun_out/pmc_<tag>_{FETCH_SIZE,WRITE_SIZE}/**/counter_collection.csv (newest file of each kind only)
Writes  profiles/bench_<round>.json, bench_<round>_serial.json, <round>_kernel_stats.csv, <round>_kernel_table.txt,
        <round>_pmc_summary.json, traffic_<round>.json
HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: separate --pmc passes, FETCH_SIZE doubled per the gfx950 correction of
MI355X_MICROARCH.md.
"""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
rnd = sys.argv[2] if len(sys.argv) > 2 else tag
G, P = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")


def newest(pattern):
    fs = glob.glob(pattern, recursive=True)
    if not fs:
        raise SystemExit("missing: " + pattern)
    return max(fs, key=os.path.getmtime)


for src, dst in (("bench_default.json", f"bench_{rnd}.json"), ("bench_serial.json", f"bench_{rnd}_serial.json")):
    line = open(os.path.join(G, src)).read().strip().splitlines()[-1]
    json.loads(line)
    open(os.path.join(P, dst), "w").write(line + "\n")

stats = newest(f"{G}/prof_{tag}/**/*kernel_stats.csv")
rows = list(csv.DictReader(open(stats)))
open(os.path.join(P, f"{rnd}_kernel_stats.csv"), "w").write(open(stats).read())
with open(os.path.join(P, f"{rnd}_kernel_table.txt"), "w") as f:
    for r in rows[:22]:
        f.write("%-60s calls %4s avg_us %10.1f  pct %5s\n" % (r["Name"].replace("(anonymous namespace)::", "").split("(")[0][-60:], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))

res = collections.defaultdict(dict)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(newest(f"{G}/pmc_{tag}_{c}/**/*counter_collection.csv"))):
        if r.get("Counter_Name") == c:
            acc[r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        res[k][c] = sum(v) / len(v)
        res[k]["launches_" + c] = len(v)
def find(sub):
    """full kernel name (templates and return type included) of the kernel whose name contains `sub`"""
    exact = [k for k in res if k.split("::")[-1].split("<")[0].strip() == sub]       # k_telea_window, not k_telea_window_retry
    ks = exact or [k for k in res if sub in k]
    if not ks:
        raise SystemExit("no counter rows for a kernel named *%s*" % sub)
    return max(ks, key=lambda k: res[k].get("launches_FETCH_SIZE", 0))


want = [find("k_telea_window_mw"), find("k_robust_polyfit")]
uf = [k for k in res if "k_uf_" in k]                    # the unwrap check is a chain of small kernels: their sum
top = dict(sorted(res.items(), key=lambda kv: -(kv[1].get("FETCH_SIZE", 0) + kv[1].get("WRITE_SIZE", 0)))[:24])
for k in want:
    top[k] = res[k]
json.dump(top, open(os.path.join(P, f"{rnd}_pmc_summary.json"), "w"), indent=1)


def hbm(k):
    return (2 * res[k]["FETCH_SIZE"] + res[k]["WRITE_SIZE"]) * 1024


tp = os.path.join(P, f"traffic_{rnd}.json")
import datetime
import subprocess
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (only for the source fingerprint)
note = ("HBM bytes per launch (per stage: the stage's dominant kernel, x3 for the three polyfit launches) = (2*FETCH_SIZE + WRITE_SIZE) * 1024 from two "
        "separate rocprofv3 --pmc passes; FETCH_SIZE doubled per the gfx950 correction of MI355X_MICROARCH.md; B=256 frames of 224x224.  _csrc_sha is the "
        "fingerprint of the kernel sources these numbers were measured on (bench.py reports them only for the same sources).")
head = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
json.dump({"inpaint (k_telea_window_mw)": hbm(want[0]), "unwrap check (k_unwrap_fast)": sum(hbm(k) for k in uf),
           "detrend (3x IRLS)": 3 * hbm(want[1]), "_note": note, "_head": head + " (+ working tree)", "_date": datetime.date.today().isoformat(),
           "_csrc_sha": bench._csrc_sha(), "_raw": {k: res[k] for k in want + uf}}, open(tp, "w"), indent=1)
b = json.loads(open(os.path.join(P, f"bench_{rnd}.json")).read())
s = json.loads(open(os.path.join(P, f"bench_{rnd}_serial.json")).read())
tel = [r for r in rows if "k_robust_polyfit" in r["Name"]]
print("default %.0f fps %.2f ms | serial %.0f fps %.2f ms | polyfit rocprof %.1f us per launch, dominant stage by events %.3f ms" % (
    b["value"], b["ms_per_step"], s["value"], s["ms_per_step"], float((tel or rows)[0]["AverageNs"]) / 1e3, b["roofline"]["kernel_ms"]))
