#!/bin/bash
mkdir -p gpurun_out/r02
timeout -k 10 500 python -m pytest tests/test_gpu_ops.py tests/test_gpu_models.py tests/test_gpu_fullsize.py tests/test_gpu_sparse.py -q -m gpu -k "gru or dien or DIEN" 2>&1 | tail -6
for fg in 1 0; do
  CTR_DIEN_FUSED_GRU=$fg timeout -k 10 300 python bench.py --workload dien --no-gather-leg --no-cpu-baseline --steps 20 --warmup 5 > gpurun_out/r02/bench_dien_fg$fg.json 2> gpurun_out/r02/bench_dien.err || tail -5 gpurun_out/r02/bench_dien.err
python - <<PY
import json
d = json.load(open("gpurun_out/r02/bench_dien_fg$fg.json"))
print("fused gru $fg:", round(d["value"]/1e6, 2), "M/s ms", round(d["ms_per_step"], 4), "full", d["full_step"] and round(d["full_step"]["ms_per_step"], 3), {k: v["avg_us"] for k, v in list(d["kernels"].items())[:10]})
PY
done
