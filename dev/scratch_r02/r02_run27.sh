#!/bin/bash
mkdir -p gpurun_out/r02
timeout -k 10 400 python -m pytest tests/test_gpu_ops.py tests/test_gpu_models.py tests/test_gpu_fullsize.py tests/test_gpu_sparse.py -q -m gpu -k "din or DIN or dien or DIEN or afm or attention" 2>&1 | tail -4
for w in "din" "din --sparse" "dien"; do
  tag=$(echo $w | tr -d ' -')
  timeout -k 10 300 python bench.py --workload $w --no-gather-leg --no-cpu-baseline --steps 20 --warmup 5 > gpurun_out/r02/bench_${tag}.json 2> gpurun_out/r02/bench_$tag.err || tail -5 gpurun_out/r02/bench_$tag.err
python - <<PY
import json
d = json.load(open("gpurun_out/r02/bench_${tag}.json"))
print("$tag:", round(d["value"]/1e6, 2), "M/s ms", round(d["ms_per_step"], 4), "full", d["full_step"] and round(d["full_step"]["ms_per_step"], 3), {k: v["avg_us"] for k, v in list(d["kernels"].items())[:6]})
PY
done
