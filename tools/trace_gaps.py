"""Idle time between consecutive kernels of the same hardware queue in a rocprofv3 --kernel-trace CSV of bench.py (sub-shard
streams: one queue per sub-shard), by (previous kernel, next kernel), and the period of the backward launches per queue.
Usage: trace_gaps.py kernel_trace.csv"""
import csv, sys
from collections import defaultdict
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "aslr::" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
name = lambda n: n.split("aslr::")[1].split("<")[0]
rows = rows[len(rows) // 4:]  # steady state
last, gaps, dur, starts = {}, defaultdict(list), defaultdict(list), defaultdict(list)
for r in rows:
    q, n, s, e = r.get("Queue_Id", "?"), name(r["Kernel_Name"]), int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if q in last:
        gaps[(last[q][0], n)].append((s - last[q][1]) / 1e3)
    last[q] = (n, e)
    dur[n].append((e - s) / 1e3)
    if n == "backward_kernel": starts[q].append(s)
for k, v in sorted(gaps.items(), key=lambda kv: -sum(kv[1]) / len(kv[1])):
    print("%-26s -> %-26s n %3d  mean gap %7.1f us  max %7.1f" % (k[0], k[1], len(v), sum(v) / len(v), max(v)))
for k, v in dur.items():
    print("%-26s n %3d  mean duration %7.1f us" % (k, len(v), sum(v) / len(v)))
for q, v in starts.items():
    d = [(b - a) / 1e3 for a, b in zip(v, v[1:])]
    if d: print("queue %s: period of the backward launches %.1f us (%d)" % (q, sum(d) / len(d), len(d)))
