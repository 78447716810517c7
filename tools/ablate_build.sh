# timing-only builds of sell_pipeline.hip with extra -D flags (results invalid), e.g.
#   VARIANTS="|-DGNN_TIMING_EXPPROD" bash tools/ablate_build.sh
# The variants are linked to a TEMPORARY copy of the library; the in-tree libgnn_hip.so is saved first
# and put back on exit, whatever happens (a timing build must never be left in the tree).
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
LIB=gnn-fpga_amd/libgnn_hip.so
cp $LIB /tmp/libgnn_hip.keep.so
trap 'cp /tmp/libgnn_hip.keep.so $LIB' EXIT
IFS='|' read -ra VS <<< "${VARIANTS:-|-DGNN_ABLATE_TRANS|-DGNN_ABLATE_LDS}"
for variant in "${VS[@]}"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -DGNN_DIAG $variant -c -o /tmp/sell_ab.o gnn-fpga_amd/csrc/sell_pipeline.hip 2>/dev/null
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $LIB build/gnn_kernels.o /tmp/sell_ab.o build/backward.o build/plan_build.o
  timeout -k 10 200 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-train --no-pruned --no-c5 ${BENCH_ARGS:-} 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('variant [$variant]', d['ms_per_step'], d['roofline']['kernel_ms'])"
done
