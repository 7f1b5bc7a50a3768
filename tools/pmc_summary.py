#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files: per kernel name, mean counter value per dispatch.
usage: python tools/pmc_summary.py gpurun_out/pmc_*  (directories)"""
import csv, glob, os, sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(list))
for d in sys.argv[1:]:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        per = defaultdict(lambda: defaultdict(float))
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"]
            if "phyhip" not in k:
                continue
            k = k.split("(")[0].replace("void phyhip::", "")
            per[(k, row["Dispatch_Id"])][row["Counter_Name"]] += float(row["Counter_Value"])
        for (k, _), cs in per.items():
            for c, v in cs.items():
                acc[k][c].append(v)
for k, cs in acc.items():
    print(k)
    for c, vs in sorted(cs.items()):
        print("   %-24s mean %.6g  (n=%d)" % (c, sum(vs) / len(vs), len(vs)))
