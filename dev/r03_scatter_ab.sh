#!/bin/bash
# round 3: the cfg3b scatter, plain kernel against the hot-row kernel at several workgroup counts (bench.py gather leg)
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/r03
cd $R
timeout -k 10 300 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_ops.py -m gpu -x -q -k "embed or scatter or stage" 2>&1 | tail -2
for cfg in "0 0 9 256" "1 256 11 1024" "1 256 10 1024" "1 512 10 512" "1 512 11 512"; do
  set -- $cfg
  CTR_EMBED_HOT=$1 CTR_EMBED_HOT_WGS=$2 CTR_EMBED_HOT_SLOTS=$3 CTR_EMBED_HOT_THREADS=$4 timeout -k 10 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r03/sc_$1_$2_$3.json 2> gpurun_out/r03/sc.err || tail -3 gpurun_out/r03/sc.err
  python - <<PY
import json
d=json.load(open("$R/gpurun_out/r03/sc_$1_$2_$3.json"))["gather_roofline"]["scatter_bwd"]
print("hot $1 wgs $2 slots 2^$3 threads $4: uniform %.1f us  zipf %.1f us" % (d["uniform"]["avg_us"], d["zipf"]["avg_us"]))
PY
done
