#!/usr/bin/env python3
"""Condenses rocprofv3 output directories into the small summaries kept under profiles/.

  prof_summary.py stats <rocprof-dir> <out.csv>       kernel_stats.csv of a --kernel-trace --stats run, trimmed
  prof_summary.py pmc   <out.json> <note> <dir>...    per-kernel averages of every counter of one or more --pmc passes

Counters are averaged per kernel over its launches.  Units and the gfx950 corrections follow
/opt/skills/guides/MI355X_MICROARCH.md (HBM / rocprofv3): FETCH_SIZE and WRITE_SIZE are KiB per launch,
FETCH_SIZE tallies 64 B per 128-B request on gfx950, so read bytes = 2 * FETCH_SIZE * 1024; WRITE_SIZE is
exact; SQ_*_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles.
"""
import collections
import csv
import glob
import json
import os
import sys


def find(d, suffix):
    fs = sorted(glob.glob(os.path.join(d, "**", "*" + suffix), recursive=True), key=os.path.getmtime)
    if not fs:
        raise SystemExit("no *%s under %s" % (suffix, d))
    if len(fs) > 1:
        raise SystemExit("several *%s under %s (stale output of an earlier run?): %s" % (suffix, d, fs))
    return fs[0]


def short(name):
    name = name.replace("mppi::", "").replace("(anonymous namespace)::", "")
    return name.split("(")[0][:96]


def stats(d, out):
    rows = list(csv.DictReader(open(find(d, "kernel_stats.csv"))))
    keep = ["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"]
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(keep)
        for r in rows:
            w.writerow([short(r["Name"])] + [r[k] for k in keep[1:]])
    for r in rows[:6]:
        print("%-70s calls %5s avg %10.1f us" % (short(r["Name"])[:70], r["Calls"], float(r["AverageNs"]) / 1e3))


def pmc(out, note, dirs):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in dirs:
        for r in csv.DictReader(open(find(d, "counter_collection.csv"))):
            acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    res = {"note": note, "units": "averages per launch; FETCH_SIZE/WRITE_SIZE in KiB; SQ *_CYCLES, SQ_WAIT_*, "
                                  "SQ_ACTIVE_INST_* in quad-cycles (MI355X_MICROARCH.md)", "kernels": {}}
    for k, v in acc.items():
        if "rocclr" in k or "fillBuffer" in k:
            continue
        e = {c: sum(x) / len(x) for c, x in v.items()}
        e["launches"] = max(len(x) for x in v.values())
        if "FETCH_SIZE" in e:
            e["read_bytes_corrected"] = 2.0 * e["FETCH_SIZE"] * 1024.0
        if "WRITE_SIZE" in e:
            e["write_bytes"] = e["WRITE_SIZE"] * 1024.0
        if "read_bytes_corrected" in e and "write_bytes" in e:
            e["traffic_bytes_per_launch"] = e["read_bytes_corrected"] + e["write_bytes"]
        res["kernels"][k] = e
    bj = os.environ.get("PROF_BENCH_JSON")
    if bj and os.path.exists(bj):  # the bench line of the --stats pass of the same command: names the workload
        try:
            c = json.loads([l for l in open(bj).read().splitlines() if l.startswith("{")][0])["config"]
            res["workload"] = {"K": c["K"], "T": c["T"], "layers": c["layers"], "rollout_variant": c["rollout_variant"]}
        except (IndexError, KeyError, ValueError):
            pass
    with open(out, "w") as f:
        json.dump(res, f, indent=1, sort_keys=True)
    for k, e in res["kernels"].items():
        print(k[:70], {c: round(x, 1) for c, x in e.items()})


if __name__ == "__main__":
    if len(sys.argv) >= 4 and sys.argv[1] == "stats":
        stats(sys.argv[2], sys.argv[3])
    elif len(sys.argv) >= 5 and sys.argv[1] == "pmc":
        pmc(sys.argv[2], sys.argv[3], sys.argv[4:])
    else:
        raise SystemExit(__doc__)
