#!/usr/bin/env python3
"""Kernel timeline of the LAST call of a traced program (rocprofv3 --kernel-trace, rocpd database): the dispatches behind the last gap of more than
GAP us, each with its start relative to the first, its duration and the idle time in front of it -- what one library call costs on the device
and how much of it is the spaces between its kernels.
usage: python tools/measure/call_timeline.py DIR_OR_DB [GAP_US=30] [CALLS=3]"""
import glob, os, re, sqlite3, sys


def short(name):
    return re.sub(r"^void ", "", name).split("(")[0].replace("phyhip::", "")[:60]


path = sys.argv[1]
gap_us = float(sys.argv[2]) if len(sys.argv) > 2 else 30.0
calls = int(sys.argv[3]) if len(sys.argv) > 3 else 3
if os.path.isdir(path):
    path = sorted(glob.glob(os.path.join(path, "**", "*_results.db"), recursive=True))[-1]
cur = sqlite3.connect(path).cursor()
rows = list(cur.execute("select name, start, end from kernels order by start"))
groups, g = [], []
for n, s, e in rows:
    if g and (s - g[-1][2]) / 1e3 > gap_us:
        groups.append(g)
        g = []
    g.append((n, s, e))
if g:
    groups.append(g)
print("## %s: %d dispatches in %d calls (gap > %.0f us)" % (path, len(rows), len(groups), gap_us))
for g in groups[-calls:]:
    t0, busy = g[0][1], 0.0
    print("call of %d kernels, %.1f us from first start to last end" % (len(g), (g[-1][2] - t0) / 1e3))
    prev = None
    for n, s, e in g:
        busy += (e - s) / 1e3
        print("  +%8.1f us  %7.1f us  (idle before: %6.1f)  %s" % ((s - t0) / 1e3, (e - s) / 1e3, 0.0 if prev is None else (s - prev) / 1e3, short(n)))
        prev = e
    print("  kernels busy %.1f us, between kernels %.1f us" % (busy, (g[-1][2] - t0) / 1e3 - busy))
