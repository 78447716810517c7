"""Derive per-kernel HBM bytes per launch from the separate FETCH_SIZE / WRITE_SIZE PMC passes of
tools/profile_bench.sh and write profiles/pmc_traffic.json (read by bench.py's roofline.traffic).
usage: python tools/make_pmc_traffic.py gpurun_out/<tag> profiles/<dir-with-summary>"""
import csv
import glob
import json
import os
import re
import sys

out, prof = sys.argv[1], sys.argv[2]


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    return re.sub(r"^void ", "", name).split("(")[0]


res = {}
for cname in ("FETCH_SIZE", "WRITE_SIZE"):
    found = glob.glob(os.path.join(out, "pmc_" + cname, "**", "*counter_collection.csv"), recursive=True)
    if found:
        acc = {}
        for r in csv.DictReader(open(found[0])):
            acc.setdefault(short(r["Kernel_Name"]), []).append(float(r["Counter_Value"]))
        for k, v in acc.items():
            res.setdefault(k, {})[cname + "_KB_mean"] = sum(v) / len(v)
            res[k]["launches"] = len(v)
        continue
    # profile_bench.sh drops csv files above 2 MB (the plan builder's torch kernels fill them):
    # the per-kernel means are in its summary.txt
    for line in open(os.path.join(out, "summary.txt")):
        m = re.match(r"\s+(\S.*?)\s+%s mean (\S+) \(n=(\d+)\)" % cname, line)
        if m:
            k = short(m.group(1))
            res.setdefault(k, {})[cname + "_KB_mean"] = float(m.group(2))
            res[k]["launches"] = int(m.group(3))
for v in res.values():
    v["hbm_bytes_per_launch"] = (2 * v.get("FETCH_SIZE_KB_mean", 0) + v.get("WRITE_SIZE_KB_mean", 0)) * 1024
doc = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on `python3 bench.py "
                 "--steps 5 --warmup 2 --no-cpu-baseline`, MI355X; summary in %s/summary.txt" % prof,
       "correction": "MI355X_MICROARCH.md HBM section: FETCH_SIZE counts 64 B per 128-B request on gfx950 "
                     "-> doubled; WRITE_SIZE exact; both in KB",
       "kernels": res}
json.dump(doc, open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles",
                                 "pmc_traffic.json"), "w"), indent=1)
for k, v in res.items():
    if k.startswith("k_"):
        print("%-36s %.1f MB" % (k, v["hbm_bytes_per_launch"] / 1e6))
