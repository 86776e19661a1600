"""Timeline of a few steady-state iterations from a rocprofv3 --kernel-trace CSV of bench.py: per kernel launch its queue,
start (us, relative) and duration, for the last N launches.  Usage: trace_timeline.py kernel_trace.csv [N]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
N = int(sys.argv[2]) if len(sys.argv) > 2 else 60
rows = [r for r in rows if "aslr::" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-N:]
t0 = int(rows[0]["Start_Timestamp"])
short = lambda n: n.split("aslr::")[1].split("<")[0]
for r in rows:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print("queue %s  %-18s start %9.1f us  dur %7.1f us  grid %s" % (r.get("Queue_Id", "?"), short(r["Kernel_Name"]), s / 1e3, (e - s) / 1e3, r.get("Grid_Size_X", r.get("Grid_Size", "?"))))
