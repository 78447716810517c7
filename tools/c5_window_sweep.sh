set -e
python -m pytest tests -m gpu -x -q > gpurun_out/r3_t1.log 2>&1 || { tail -30 gpurun_out/r3_t1.log; exit 1; }
tail -2 gpurun_out/r3_t1.log
for kb in 3072 2048 4096 1048576; do
  GNN_WIDE_WINDOW_KB=$kb python bench.py --workload c5 --dtype f32 --steps 20 --warmup 5 --no-train --no-cpu-baseline > gpurun_out/r3_c5_f32_$kb.json 2>gpurun_out/r3_c5.err
  python - <<P
import json
d=json.load(open("gpurun_out/r3_c5_f32_$kb.json")); r=d["roofline"]
print("f32 window_kb=$kb ms", round(d["ms_per_step"],4), "other(bf16)", round(d["other_dtype"]["ms_per_step"],4), r["launch_sequence_ms"])
P
done
