#!/bin/bash
# round 3: the reference script's NeuralCF shape -- parity tests, then the bench leg with the table-row path and without
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/r03
cd $R
timeout -k 10 300 python -m pytest tests/test_gpu_models.py -m gpu -x -q -k "any_tower" 2>&1 | tail -3
for p in 1 0; do
  CTR_NCF_PROJ=$p timeout -k 10 200 python bench.py --workload neuralcf_script --steps 30 --warmup 5 --no-gather-leg --no-cpu-baseline > gpurun_out/r03/script_$p.json 2> gpurun_out/r03/script.err || tail -3 gpurun_out/r03/script.err
  python - <<PY
import json
d=json.load(open("$R/gpurun_out/r03/script_$p.json"))
print("CTR_NCF_PROJ=$p:", round(d["value"]/1e6,1), "M/s", round(d["ms_per_step"]*1e3,1), "us/step; top kernels", list(d["kernels"].items())[:6])
PY
done
