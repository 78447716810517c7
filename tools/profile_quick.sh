#!/bin/bash
# Quick counter passes on bench.py (run on the GPU box via gpurun): tools/profile_quick.sh <tag> <kernel regex> [bench args...]
set -u
TAG=${1:-q}; KRE=${2:-k_}; shift 2 || true
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 3 --warmup 1 --no-cpu-baseline --no-train --no-pruned $*"
for pass in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INST_CYCLES_VMEM"; do
  name=$(echo $pass | tr ' ' '_')
  rocprofv3 --pmc $pass --kernel-trace --output-format csv -d "$OUT/pmc_$name" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/pmc_$name.log" 2>&1 || { echo "pmc $pass failed"; tail -5 "$OUT/pmc_$name.log"; }
done
python3 "$ROOT/tools/summarize_prof.py" "$OUT" 2>&1 | grep -E "^==|$KRE" | cut -c1-420 > "$OUT/summary.txt"
cat "$OUT/summary.txt"
for f in $(find "$OUT" -name "*counter_collection.csv"); do
  { head -1 "$f"; grep -E '"[^"]*(k_iter|k_edge|k_input|k_pack|k_event|k_node|k_pq)' "$f"; } > "$f.tmp" && mv "$f.tmp" "$f"
done
find "$OUT" -name "*.csv" -size +2M -delete
