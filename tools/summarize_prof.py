"""Summarise rocprofv3 output dirs written by tools/profile_bench.sh: per-kernel count, mean
duration (kernel trace) and mean counter value per dispatch (PMC passes)."""
import csv
import glob
import os
import re
import sys
from collections import defaultdict

out = sys.argv[1]


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return name.split("(")[0][:48]


def find(d, pat):
    return sorted(glob.glob(os.path.join(d, "**", pat), recursive=True))


for f in find(os.path.join(out, "trace"), "*kernel_stats.csv"):
    print("== kernel stats (%s)" % os.path.relpath(f, out))
    with open(f) as fh:
        rows = list(csv.DictReader(fh))
    for r in rows[:12]:
        print("  %-48s calls %6s  avg_ns %12s  total_ns %14s  %5s%%" % (
            short(r["Name"]), r["Calls"], r["AverageNs"], r["TotalDurationNs"], r["Percentage"]))

for d in sorted(glob.glob(os.path.join(out, "pmc_*"))):
    if not os.path.isdir(d):
        continue
    for f in find(d, "*counter_collection.csv"):
        acc = defaultdict(lambda: defaultdict(list))
        with open(f) as fh:
            for r in csv.DictReader(fh):
                acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
        print("== %s" % os.path.relpath(f, out))
        for k, cs in acc.items():
            if not k.startswith("k_"):
                continue
            print("  %-48s " % k + "  ".join("%s mean %.4g (n=%d)" % (c, sum(v) / len(v), len(v))
                                             for c, v in cs.items()))
