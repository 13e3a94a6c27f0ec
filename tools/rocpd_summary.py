#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd database (the *_results.db it writes) into a small CSV:
per-kernel call count, total and average duration (microseconds), share of GPU time,
plus launch geometry and register counts of each distinct kernel.

    python tools/rocpd_summary.py gpurun_out/prof/x_results.db profiles/rNN/name.csv "command line that was profiled"
"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"\(.*$", "", name)
    return name[:80]


def main():
    db_path, out_path = sys.argv[1], sys.argv[2]
    cmd = sys.argv[3] if len(sys.argv) > 3 else ""
    cur = sqlite3.connect(db_path).cursor()
    rows = list(cur.execute("select name,total_calls,total_duration,average,percentage from top_kernels"))
    geo = {}
    for r in cur.execute("select name, grid_x, workgroup_x, vgpr_count, accum_vgpr_count, sgpr_count, lds_size, scratch_size "
                         "from kernels group by name"):
        geo[r[0]] = r[1:]
    with open(out_path, "w") as f:
        f.write("# rocprofv3 --kernel-trace --stats -- %s\n" % cmd)
        f.write("# source: top_kernels / kernels views of the rocpd database; durations in microseconds\n")
        f.write("kernel,calls,total_us,average_us,percent,grid_threads,workgroup,vgpr,agpr,sgpr,lds_bytes,scratch_bytes\n")
        for name, calls, total, avg, pct in rows:
            g = geo.get(name, ("",) * 7)
            f.write('"%s",%d,%.3f,%.3f,%.4f,%s\n' % (short(name), calls, total, avg, pct, ",".join(str(x) for x in g)))
    print(open(out_path).read())


if __name__ == "__main__":
    main()
