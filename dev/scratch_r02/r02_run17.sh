#!/bin/bash
mkdir -p gpurun_out/r02
timeout -k 10 300 python -m pytest tests/test_gpu_ops.py tests/test_gpu_models.py -q -m gpu -k "allpairs or pnn" 2>&1 | tail -4
CTR_PAIRS_WAVES=2 timeout -k 10 300 python -m pytest tests/test_gpu_ops.py -q -m gpu -k "allpairs" 2>&1 | tail -2
for wv in 1 2; do
CTR_PAIRS_WAVES=$wv timeout -k 10 300 python bench.py --workload pnn26 --no-gather-leg --no-cpu-baseline --steps 20 --warmup 5 > gpurun_out/r02/bench_pnn26_w$wv.json 2> gpurun_out/r02/bench_pnn26.err || tail -5 gpurun_out/r02/bench_pnn26.err
python - <<PY
import json
d = json.load(open("gpurun_out/r02/bench_pnn26_w$wv.json"))
print("waves $wv:", round(d["value"]/1e6, 2), "M/s ms", round(d["ms_per_step"], 4), {k: v["avg_us"] for k, v in d["kernels"].items() if "allpairs" in k})
PY
done
