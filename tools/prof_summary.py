#!/usr/bin/env python3
"""Condenses rocprofv3 output directories into the small summaries kept under profiles/.

  prof_summary.py stats <rocprof-dir> <out.csv>       kernel_stats.csv of a --kernel-trace --stats run, trimmed
  prof_summary.py pmc   <out.json> <note> <dir>...    per-kernel averages of every counter of one or more --pmc passes
  prof_summary.py timeline <rocprof-dir> <out.json> [note]   dispatch timeline of a --kernel-trace run: how much
                                                      consecutive rollout kernels overlap, gaps between kernels

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
    try:  # the kernel sources this profile was taken on (bench.py refuses a traffic figure of other sources)
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        import bench
        res["kernel_sources_sha16"] = bench.kernel_sources_sha16()
    except Exception as e:  # noqa: BLE001
        res["kernel_sources_sha16"] = "unknown (%s)" % type(e).__name__
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


def timeline(d, out, note=""):
    """From the kernel trace (begin / end stamps of every dispatch): (1) for consecutive rollout dispatches on
    DIFFERENT queues (the two controllers, one stream each) the fraction of the shorter one that ran beside the
    other; (2) the gap between the end of a rollout dispatch and the begin of the next tail dispatch on the same
    queue (the kernel boundary inside a solve); (3) begin-to-begin period of the rollout dispatches."""
    rows = list(csv.DictReader(open(find(d, "kernel_trace.csv"))))
    ev = []
    for r in rows:
        n = short(r["Kernel_Name"])
        if "rocclr" in n or "fillBuffer" in n:
            continue
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n, r.get("Queue_Id", "0"), r.get("Stream_Id", "0")))
    ev.sort()
    roll = [e for e in ev if "rollout" in e[2]]
    tail = [e for e in ev if "solve_tail" in e[2]]
    skip = max(4, len(roll) // 10)  # warm-up dispatches
    roll_s, tail_s = roll[skip:], tail[skip:]
    res = {"note": note, "units": "ns", "dispatches": len(ev), "rollout_dispatches": len(roll),
           "kernels": sorted(set(e[2] for e in ev))}
    dur = [e[1] - e[0] for e in roll_s]
    res["rollout_duration_avg"] = sum(dur) / max(1, len(dur))
    tdur = [e[1] - e[0] for e in tail_s]
    res["tail_duration_avg"] = sum(tdur) / max(1, len(tdur))
    # (1) overlap of neighbouring rollout dispatches from different queues / streams
    ov, pairs = [], 0
    for a, b in zip(roll_s[:-1], roll_s[1:]):
        if (a[3], a[4]) == (b[3], b[4]):
            continue
        pairs += 1
        o = max(0, min(a[1], b[1]) - max(a[0], b[0]))
        ov.append(o / max(1, min(a[1] - a[0], b[1] - b[0])))
    res["cross_queue_rollout_pairs"] = pairs
    res["cross_queue_overlap_fraction_avg"] = (sum(ov) / len(ov)) if ov else None
    res["cross_queue_overlap_fraction_min"] = min(ov) if ov else None
    # (2) rollout end -> next tail begin on the same queue
    gaps = []
    for r in roll_s:
        nxt = [t for t in tail_s if t[0] >= r[1] - 2000 and (t[3], t[4]) == (r[3], r[4])]
        if nxt:
            gaps.append(nxt[0][0] - r[1])
    gaps.sort()
    if gaps:
        res["rollout_end_to_tail_begin"] = {"median": gaps[len(gaps) // 2], "p10": gaps[len(gaps) // 10],
                                            "p90": gaps[(9 * len(gaps)) // 10], "n": len(gaps)}
    # (3) period
    per = sorted(b[0] - a[0] for a, b in zip(roll_s[:-1], roll_s[1:]))
    if per:
        res["rollout_begin_to_begin_median"] = per[len(per) // 2]
    # tail end -> next rollout begin (the host's turn-around between two solves)
    turn = []
    for t in tail_s:
        nxt = [r for r in roll_s if r[0] >= t[0]]
        if nxt:
            turn.append(nxt[0][0] - t[1])
    turn.sort()
    if turn:
        res["tail_end_to_next_rollout_begin_median"] = turn[len(turn) // 2]
    with open(out, "w") as f:
        json.dump(res, f, indent=1, sort_keys=True)
    print(json.dumps({k: v for k, v in res.items() if k != "kernels"}))


if __name__ == "__main__":
    if len(sys.argv) >= 4 and sys.argv[1] == "timeline":
        timeline(sys.argv[2], sys.argv[3], sys.argv[4] if len(sys.argv) > 4 else "")
        sys.exit(0)
    if len(sys.argv) >= 4 and sys.argv[1] == "stats":
        stats(sys.argv[2], sys.argv[3])
    elif len(sys.argv) >= 5 and sys.argv[1] == "pmc":
        pmc(sys.argv[2], sys.argv[3], sys.argv[4:])
    else:
        raise SystemExit(__doc__)
