#!/bin/bash
# round 3: NeuralCF headline step -- bench line + rocprofv3 kernel stats of the same command (eager launches)
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-x}
mkdir -p $R/gpurun_out/r03
cd $R
timeout -k 10 200 python bench.py --steps 100 --warmup 10 --no-gather-leg --no-cpu-baseline > gpurun_out/r03/${TAG}_bench.json 2> gpurun_out/r03/${TAG}_bench.err || tail -5 gpurun_out/r03/${TAG}_bench.err
python - <<PY
import json
d=json.load(open("$R/gpurun_out/r03/${TAG}_bench.json"))
print("value", round(d["value"]/1e6,1), "M/s  ms", round(d["ms_per_step"]*1e3,1), "us; kernels", {k:v["avg_us"] for k,v in d["kernels"].items()})
PY
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r03/${TAG}_prof -- python3 $R/bench.py --no-gather-leg --no-cpu-baseline --no-graph --steps 20 > /dev/null 2> $R/gpurun_out/r03/${TAG}_prof.err
f=$(ls $R/gpurun_out/r03/${TAG}_prof/*/*kernel_stats.csv | head -1)
cp $f $R/gpurun_out/r03/${TAG}_kernel_stats.csv
head -14 $f | cut -c1-160
