#!/usr/bin/env python3
"""Timeline of the library's kernels from a rocprofv3 --kernel-trace CSV: start, duration, queue of every trace and
resolve kernel and of the fills between them.   python tools/trace_timeline.py <kernel_trace.csv> [last N rows]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
last = int(sys.argv[2]) if len(sys.argv) > 2 else 60
t0 = min(int(r["Start_Timestamp"]) for r in rows)
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
sel = [r for r in rows if "dray" in r["Kernel_Name"] or "fill" in r["Kernel_Name"].lower()]
for r in sel[-last:]:
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:34]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%-34s q%s  start %9.3f  end %9.3f  dur %7.3f" % (name, r["Queue_Id"], (s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6))
