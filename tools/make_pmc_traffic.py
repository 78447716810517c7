"""Per-kernel HBM bytes per launch (and the vector-ALU busy fraction) from the separate rocprofv3 --pmc
passes of tools/profile_bench.sh -> profiles/pmc_traffic.json, which bench.py reads for `roofline.traffic`
(c3) and the `c5` sub-record.

usage: python tools/make_pmc_traffic.py gpurun_out/<tag> profiles/<dir-with-summary> [section]
section: c3 (default; top level of the file), c5_f32, c5_bf16 (own sections; the others are kept)

FETCH_SIZE is doubled (MI355X_MICROARCH.md, HBM: it counts 64 B per 128-B request on gfx950), WRITE_SIZE
is exact, both in KB.  valu_busy = SQ_ACTIVE_INST_VALU x 4 (quad-cycles -> cycles) / 1024 SIMDs, over the
kernel's cycles = SQ_BUSY_CYCLES / 32 shader engines."""
import json
import os
import re
import sys

out, prof = sys.argv[1], sys.argv[2]
section = sys.argv[3] if len(sys.argv) > 3 else "c3"
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles", "pmc_traffic.json")


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    return re.sub(r"^void ", "", name).split("(")[0]


# a c5 run times the other arithmetic beside the one it is about: keep the kernels of the section's own
# (k_iter_wx<F, D, LAST, XP, EX>: EX = true is the exact fp32 form)
DROP = {"c5_f32": r"k_iter_wx<.*, false>$|k_iter_w<.*, false>$|k_input4_bf|k_pack16",
        "c5_bf16": r"k_iter_wx<.*, true>$|k_iter_w<.*, true>$|k_input4_x|k_pack32"}.get(section)
res, sq = {}, {}
for line in open(os.path.join(out, "summary.txt")):
    m = re.match(r"\s+(k_\S.*?)\s{2,}(\S.*)$", line)
    if not m:
        continue
    k = short(m.group(1))
    if DROP and re.search(DROP, k):
        continue
    for c, v, n in re.findall(r"(\w+) mean (\S+) \(n=(\d+)\)", m.group(2)):
        if c in ("FETCH_SIZE", "WRITE_SIZE"):
            res.setdefault(k, {})[c + "_KB_mean"] = float(v)
            res[k]["launches"] = int(n)
        elif c in ("SQ_ACTIVE_INST_VALU", "SQ_BUSY_CYCLES"):
            sq.setdefault(k, {})[c] = float(v)
for v in res.values():
    v["hbm_bytes_per_launch"] = (2 * v.get("FETCH_SIZE_KB_mean", 0) + v.get("WRITE_SIZE_KB_mean", 0)) * 1024
# vector-ALU busy fraction per kernel NAME (template variants pooled by their launch-weighted mean)
valu = {}
for k, c in sq.items():
    if c.get("SQ_BUSY_CYCLES"):
        valu.setdefault(k.split("<")[0], []).append(c.get("SQ_ACTIVE_INST_VALU", 0.0) / (8.0 * c["SQ_BUSY_CYCLES"]))
rec = {"source": "rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE / SQ_* each in a run of its own) of "
                 "tools/profile_bench.sh on MI355X; summary in %s/summary.txt" % prof,
       "correction": "MI355X_MICROARCH.md HBM section: FETCH_SIZE counts 64 B per 128-B request on gfx950 "
                     "-> doubled; WRITE_SIZE exact; both in KB",
       "kernels": res,
       "valu_busy": {k: sum(v) / len(v) for k, v in valu.items()}}
doc = {}
if os.path.exists(path):
    with open(path) as f:
        doc = json.load(f)
if section == "c3":
    keep = {k: v for k, v in doc.items() if k.startswith("c5_")}
    doc = dict(rec, **keep)
else:
    doc[section] = rec
with open(path, "w") as f:
    json.dump(doc, f, indent=1)
for k, v in res.items():
    if k.startswith("k_"):
        print("%-40s %.1f MB per launch" % (k, v["hbm_bytes_per_launch"] / 1e6))
print("valu_busy", rec["valu_busy"])
