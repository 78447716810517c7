# GPU box: wide-shape parity tests, then the c5 line (exact fp32 timed, bf16 beside it)
set -e
python -m pytest tests -m gpu -x -q -k "golden or wide or c5 or bf16 or matrix" > gpurun_out/c5q_tests.log 2>&1 || { tail -30 gpurun_out/c5q_tests.log; exit 1; }
tail -2 gpurun_out/c5q_tests.log
python bench.py --workload c5 --dtype f32 --steps 20 --warmup 5 --no-train --no-cpu-baseline > gpurun_out/c5q_f32.json 2>gpurun_out/c5q.err
python - <<P
import json
d=json.load(open("gpurun_out/c5q_f32.json")); r=d["roofline"]
print("f32 ms", round(d["ms_per_step"],4), "other(bf16)", round(d["other_dtype"]["ms_per_step"],4), r["launch_sequence_ms"])
P
