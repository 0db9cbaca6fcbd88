#!/bin/bash
# end-of-round evidence: GPU tests, default bench line, rocprofv3 kernel stats of the same command, PMC traffic
R=$PWD; rm -rf gpurun_out/r01_prof gpurun_out/traffic_FETCH_SIZE gpurun_out/traffic_WRITE_SIZE
python -m pytest tests -q -m gpu 2>&1 | tail -2 | tee gpurun_out/r01_gpu_tests.txt
python bench.py > gpurun_out/r01_bench_neuralcf.json 2>gpurun_out/r01_bench.err; tail -c 1500 gpurun_out/r01_bench_neuralcf.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r01_prof -- python3 $R/bench.py --no-cpu-baseline > $R/gpurun_out/r01_bench_under_rocprof.json 2>$R/gpurun_out/r01_prof.err
cd $R
cp gpurun_out/r01_prof/*/*kernel_stats.csv gpurun_out/r01_neuralcf_kernel_stats.csv
head -12 gpurun_out/r01_neuralcf_kernel_stats.csv | cut -c1-150
bash dev/pmc_traffic.sh r01_neuralcf --steps 10 --warmup 3
