#!/bin/bash
# end of round 3: smoke, full GPU suite, per-workload sweep
set -e
mkdir -p gpurun_out/r03
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout -k 10 1000 python -m pytest tests -q -m gpu > gpurun_out/r03/final_tests.txt 2>&1 || { tail -40 gpurun_out/r03/final_tests.txt; exit 1; }
tail -2 gpurun_out/r03/final_tests.txt
bash dev/r03_sweep.sh > gpurun_out/r03/final_sweep.log 2>&1
grep "^==" gpurun_out/r03s/summary.txt | cut -c1-120
