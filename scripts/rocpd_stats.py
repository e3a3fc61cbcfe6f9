#!/usr/bin/env python
"""Kernel-stats summary (name, calls, total/avg/min/max ns, %) from a rocprofv3 rocpd .db file
(rocprofv3 --kernel-trace --stats on ROCm 7.2 writes SQLite unless --output-format csv is given)."""
import sqlite3
import sys


def main(path, out=None):
    db = sqlite3.connect(path)
    rows = db.execute(
        "select name, count(*), sum(duration), avg(duration), min(duration), max(duration), max(vgpr_count), "
        "max(lds_size), max(grid_x), max(workgroup_x) from kernels group by name order by sum(duration) desc").fetchall()
    tot = sum(r[2] for r in rows) or 1
    lines = ['"Name","Calls","TotalDurationNs","AverageNs","MinNs","MaxNs","Percentage","VGPRs","LDS","GridX","WorkgroupX"']
    for r in rows:
        lines.append('"%s",%d,%d,%.1f,%d,%d,%.3f,%s,%s,%s,%s' % (r[0], r[1], r[2], r[3], r[4], r[5], 100.0 * r[2] / tot,
                                                                 r[6], r[7], r[8], r[9]))
    text = "\n".join(lines) + "\n"
    if out:
        open(out, "w").write(text)
    else:
        sys.stdout.write(text)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
