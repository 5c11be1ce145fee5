#!/usr/bin/env python3
"""tools/rocpd_stats.py <results.db> [top] -- per-kernel totals from a rocprofv3 rocpd database (the default output of
`rocprofv3 --kernel-trace --stats` on this image), as a CSV like the --output-format csv kernel_stats file."""
import re
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    cur = db.cursor()
    rows = cur.execute("select name, count(*), sum(end - start), avg(end - start), min(end - start), max(end - start) "
                       "from kernels group by name order by 3 desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    print('"Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs"')
    for name, calls, tot, avg, mn, mx in rows[:top]:
        short = re.sub(r"\(anonymous namespace\)::", "", name)
        short = re.sub(r"\(.*$", "", short)[:110]
        print('"%s",%d,%d,%.1f,%.2f,%d,%d' % (short, calls, tot, avg, 100.0 * tot / total, mn, mx))
    print('"TOTAL",%d,%d,,100.0,,' % (sum(r[1] for r in rows), total))


if __name__ == "__main__":
    main()
