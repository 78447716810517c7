#!/bin/bash
# Counter passes on a training probe (run on the GPU box): tools/train_scale_pmc.sh <tag> <kernel regex>
# PROBE="tools/train_wide_probe.py 64 6 1" selects another probe (default: tools/train_scale_probe.py 32 6)
set -u
TAG=${1:-trainpmc}; KRE=${2:-k_seg_bwd4}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
PROBE=${PROBE:-tools/train_scale_probe.py 32 6}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES" "SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS" "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "FETCH_SIZE" "GRBM_GUI_ACTIVE"; do
  name=$(echo $pass | tr ' ' '_')
  rocprofv3 --pmc $pass --kernel-trace --output-format csv -d "$OUT/pmc_$name" -- python3 $ROOT/$PROBE > "$OUT/pmc_$name.log" 2>&1 || { echo "pmc $pass failed"; tail -3 "$OUT/pmc_$name.log"; }
done
python3 "$ROOT/tools/summarize_prof.py" "$OUT" 2>&1 | grep -E "^==|$KRE" | cut -c1-700
find "$OUT" -name "*.csv" -size +1M -delete
