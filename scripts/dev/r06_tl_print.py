#!/usr/bin/env python3
"""Print a timeline.csv of scripts/dev/r06_timeline.sh (kernel names may contain commas).  usage: r06_tl_print.py file [from] [to]"""
import sys
rows = []
for l in open(sys.argv[1]).read().splitlines()[1:]:
    p = l.split(",")
    rows.append((int(p[0]), ",".join(p[1:-5]), float(p[-5]), float(p[-4]), float(p[-3]), p[-2], p[-1]))
a = int(sys.argv[2]) if len(sys.argv) > 2 else 0
b = int(sys.argv[3]) if len(sys.argv) > 3 else len(rows)
for r in rows[a:b]:
    print("%3d %-44s st %8.2f dur %7.2f gap %6.2f end %8.2f grid %s" % (r[0], r[1][:44], r[2], r[3], r[4], r[2] + r[3], r[5]))
print("span %.1f us; sum of durations %.1f" % (rows[-1][2] + rows[-1][3], sum(r[3] for r in rows)))
