#!/bin/bash
# Run on the GPU box (via gpurun): kernel-trace stats + separate PMC passes for bench.py.
# Usage: tools/profile_bench.sh <tag> [bench args...]
set -u
TAG=${1:-prof}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 5 --warmup 2 --no-cpu-baseline --no-train --no-pruned --no-c5 $*"
TRACE_ARGS="--steps 30 --warmup 5 --no-cpu-baseline --no-train --no-pruned --no-c5 $*"   # long enough that the clock ramp of the first steps does not dominate the averages
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$ROOT/bench.py" $TRACE_ARGS > "$OUT/trace.log" 2>&1 || { echo trace failed; tail -5 "$OUT/trace.log"; exit 1; }
for pass in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INST_CYCLES_VMEM" "GRBM_GUI_ACTIVE"; do
  name=$(echo $pass | tr ' ' '_')
  rocprofv3 --pmc $pass --kernel-trace --output-format csv -d "$OUT/pmc_$name" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/pmc_$name.log" 2>&1 || { echo "pmc $pass failed"; tail -5 "$OUT/pmc_$name.log"; }
done
python3 "$ROOT/tools/summarize_prof.py" "$OUT" > "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"
# counter files: keep the rows of this library's kernels (the plan builder's torch kernels are thousands of rows)
for f in $(find "$OUT" -name "*counter_collection.csv"); do
  { head -1 "$f"; grep -E '"[^"]*(k_iter|k_edge|k_input|k_pack|k_event|k_node|k_pq|k_exp_bound)' "$f"; } > "$f.tmp" && mv "$f.tmp" "$f"
done
# keep only small files (csv traces of 100k+ dispatches are not needed)
find "$OUT" -name "*.csv" -size +2M -delete
