#!/bin/bash
# Per-kernel times of a training probe (run on the GPU box): tools/train_scale_profile.sh <tag> [G]
# PROBE="tools/train_wide_probe.py 64 6 1" selects another probe (default: tools/train_scale_probe.py G 20)
set -u
TAG=${1:-train}; G=${2:-32}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
PROBE=${PROBE:-tools/train_scale_probe.py $G 20}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 $ROOT/$PROBE > "$OUT/probe.log" 2>&1 || tail -5 "$OUT/probe.log"
f=$(find "$OUT/trace" -name "*kernel_stats.csv" | head -1)
cp "$f" "$OUT/kernel_stats.csv"
find "$OUT/trace" -name "*.csv" -size +1M -delete
python3 - "$OUT/kernel_stats.csv" <<'PY'
import csv, sys
rows = list(csv.reader(open(sys.argv[1])))[1:]
for r in rows[:16]:
    n = r[0]
    n = n.split("::")[1] if "anonymous" in n and "::" in n else n
    print("%-44s calls %4s avg %8.1f us %6.2f %%" % (n.split("(")[0][:44], r[1], float(r[3]) / 1e3, float(r[4])))
PY
grep "training step" "$OUT/probe.log" | cut -c1-300
