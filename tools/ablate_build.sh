# timing-only builds: GNN_ABLATE_TRANS (cheap activation), GNN_ABLATE_LDS (no record reads)
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
for variant in "" "-DGNN_ABLATE_TRANS" "-DGNN_ABLATE_LDS" "-DGNN_ABLATE_TRANS -DGNN_ABLATE_LDS"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude $variant -c -o /tmp/sell_ab.o gnn-fpga_amd/csrc/sell_pipeline.hip 2>/dev/null
  cp gnn-fpga_amd/libgnn_hip.so /tmp/libgnn_hip.orig.so 2>/dev/null || true
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o gnn-fpga_amd/libgnn_hip.so build/gnn_kernels.o /tmp/sell_ab.o build/backward.o
  for a in 0 25; do GNN_ABLATE=$a timeout -k 10 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('variant [$variant] ablate',$a, d['roofline']['kernel_ms'])"; done
done
