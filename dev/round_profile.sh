#!/bin/bash
# end-of-round evidence: GPU tests, default bench line, rocprofv3 kernel stats of the same command, PMC traffic
# usage: dev/round_profile.sh <tag>      e.g. r02_a   -> gpurun_out/<tag>_*  (copy what is to be judged into profiles/)
T=${1:-r02_x}
R=$PWD; mkdir -p gpurun_out; rm -rf gpurun_out/${T}_prof gpurun_out/traffic_FETCH_SIZE gpurun_out/traffic_WRITE_SIZE
python -m pytest tests -q -m gpu 2>&1 | tail -2 | tee gpurun_out/${T}_gpu_tests.txt
python bench.py > gpurun_out/${T}_bench_neuralcf.json 2>gpurun_out/${T}_bench.err; tail -c 1500 gpurun_out/${T}_bench_neuralcf.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${T}_prof -- python3 $R/bench.py --no-cpu-baseline > $R/gpurun_out/${T}_bench_under_rocprof.json 2>$R/gpurun_out/${T}_prof.err
cd $R
cp gpurun_out/${T}_prof/*/*kernel_stats.csv gpurun_out/${T}_neuralcf_kernel_stats.csv
head -14 gpurun_out/${T}_neuralcf_kernel_stats.csv | cut -c1-150
bash dev/pmc_traffic.sh ${T}_neuralcf --steps 10 --warmup 3
