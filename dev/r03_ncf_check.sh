#!/bin/bash
# NeuralCF parity tests, the per-kernel probe under two id patterns, the default bench line
set -e
mkdir -p gpurun_out/r03
timeout -k 10 600 python -m pytest tests/test_gpu_models.py tests/test_gpu_fullsize.py -q -m gpu -x -k "neuralcf or ncf or graph or recommendation" > gpurun_out/r03/ncf_tests.txt 2>&1 || { tail -40 gpurun_out/r03/ncf_tests.txt; exit 1; }
tail -2 gpurun_out/r03/ncf_tests.txt
bash dev/r03_ids_pattern.sh random sorted | grep "== ids\|ncfp_\|bce"
timeout -k 10 300 python bench.py --no-gather-leg --no-cpu-baseline > gpurun_out/r03/${1:-chk}_bench.json 2> gpurun_out/r03/${1:-chk}_bench.err
python - ${1:-chk} <<'P'
import json,sys
d=json.loads(open(f"gpurun_out/r03/{sys.argv[1]}_bench.json").read().strip().splitlines()[-1])
print("bench", round(d["value"]/1e6,1),"M/s", round(d["ms_per_step"]*1e3,2),"us", "loss", d["loss"])
P
