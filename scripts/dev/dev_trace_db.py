"""Per-kernel totals and the timeline of the LAST registration out of a rocprofv3 (rocpd sqlite) kernel trace.
usage: dev_trace_db.py results.db [n_registrations] [timeline: 0/1]"""
import sqlite3, sys, collections
db = sqlite3.connect(sys.argv[1])
nreg = int(sys.argv[2]) if len(sys.argv) > 2 else 13
rows = list(db.execute("select name, start, end, stream from kernels order by start"))
def short(n):
    n = n.replace("void ", "")
    if "rocprim" in n:
        for k in ("radix_sort_onesweep", "onesweep_histograms", "scan_impl", "partition", "select", "lookback", "init_"):
            if k in n:
                return "rocprim:" + k
        return "rocprim:other"
    return n.split("(")[0][:48]
tot = collections.defaultdict(lambda: [0, 0.0])
for n, s, e, st in rows:
    t = tot[short(n)]
    t[0] += 1
    t[1] += (e - s) / 1e3
print("%-50s %8s %10s" % ("kernel", "calls/reg", "us/reg"))
for n, (c, us) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
    print("%-50s %8.1f %10.1f" % (n, c / nreg, us / nreg))
print("sum of kernel time per registration: %.1f us" % (sum(v[1] for v in tot.values()) / nreg))
if len(sys.argv) > 3 and sys.argv[3] == "1":
    # the last registration: from the last big gap (> 200 us) before the end... simpler: the last 260 kernels
    tail = rows[-int(sys.argv[4]) if len(sys.argv) > 4 else -230:]
    t0 = tail[0][1]
    prev_end = t0
    for n, s, e, st in tail:
        print("%9.1f  +%7.1f gap %6.1f  %-10s %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, st, short(n)))
        prev_end = max(prev_end, e)
