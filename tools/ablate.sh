# GNN_ABLATE bit masks are honoured by -DGNN_DIAG builds only (tools/ablate_build.sh); the shipped library ignores them
for a in ${ABLATE_LIST:-0 8 16 24 25 9}; do GNN_ABLATE=$a timeout -k 10 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-train --no-pruned 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('ablate',$a, d['roofline']['kernel_ms'])"; done
