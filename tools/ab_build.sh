#!/bin/bash
# A / B on ONE box: the in-tree library against a build of sell_pipeline.hip with one more -D switch (e.g.
# -DGNN_NO_PIPE_B), three alternating bench runs each.  Run through gpurun: the BOX's copy of the library is swapped
# and put back; nothing is written to the repository.
# usage: bash tools/ab_build.sh -DGNN_NO_PIPE_B [bench args]
set -e
FLAG=${1:?switch}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
LIB=gnn-fpga_amd/libgnn_hip.so
cp $LIB /tmp/ab_a.so
trap 'cp /tmp/ab_a.so $LIB' EXIT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude $FLAG -c -o /tmp/sell_ab.o gnn-fpga_amd/csrc/sell_pipeline.hip 2>/dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/ab_b.so build/gnn_kernels.o /tmp/sell_ab.o build/backward.o build/plan_build.o build/csr_build.o
for i in 1 2 3; do
  for v in a b; do
    cp /tmp/ab_$v.so $LIB
    timeout -k 10 200 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-train --no-pruned --no-c5 "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$v' == 'a' and 'in-tree ' or '$FLAG', round(d['ms_per_step'],4), 'first', round(r['launch_ms_first'],4), 'middle', round(r['launch_ms_middle'],4), 'last', round(r['launch_ms_last'],4), 'other', round(r['other_kernels_ms'],4))"
  done
done
