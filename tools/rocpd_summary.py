#!/usr/bin/env python3
"""Summarise rocprofv3 rocpd databases (*_results.db): kernel-trace statistics and PMC counters of the library's
kernels (names containing `phyhip`), one line per kernel / counter.
usage: python tools/rocpd_summary.py DIR_OR_DB [...]"""
import glob, os, re, sqlite3, sys
from collections import defaultdict


def short(name):
    name = re.sub(r"^void ", "", name).split("(")[0].replace("phyhip::", "")
    return name


def dbs(paths):
    for p in paths:
        if os.path.isdir(p):
            yield from sorted(glob.glob(os.path.join(p, "**", "*_results.db"), recursive=True))
        else:
            yield p


for path in dbs(sys.argv[1:]):
    db = sqlite3.connect(path)
    cur = db.cursor()
    print("## %s" % path)
    rows = list(cur.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels "
                            "where name like '%phyhip%' group by name order by sum(duration) desc"))
    total = cur.execute("select sum(duration) from kernels").fetchone()[0] or 1
    if rows:
        print("kernel-trace: name, calls, total_us, avg_us, min_us, max_us, pct_of_all_kernel_time")
        for n, c, s, a, lo, hi in rows:
            print("  %-70s %4d %12.1f %10.1f %10.1f %10.1f %6.2f" % (short(n), c, s / 1e3, a / 1e3, lo / 1e3, hi / 1e3, 100.0 * s / total))
    acc = defaultdict(list)
    try:
        for n, cn, v in cur.execute("select kernel_name, counter_name, value from counters_collection where kernel_name like '%phyhip%'"):
            acc[(short(n), cn)].append(v)
    except sqlite3.Error:
        pass
    if acc:
        print("pmc: kernel, counter, mean per dispatch, dispatches")
        for (n, cn), vs in sorted(acc.items()):
            print("  %-70s %-24s %16.6g %4d" % (n, cn, sum(vs) / len(vs), len(vs)))
