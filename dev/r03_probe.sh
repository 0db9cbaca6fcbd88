#!/bin/bash
# rocprofv3 kernel trace of dev/ncfp_probe.py, averaged per phase
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-p}
mkdir -p $R/gpurun_out/r03
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r03/${TAG}_probe -- python3 $R/dev/ncfp_probe.py > /dev/null 2> $R/gpurun_out/r03/${TAG}_probe.err
f=$(ls $R/gpurun_out/r03/${TAG}_probe/*/*kernel_trace.csv | head -1)
python3 $R/dev/ncfp_probe.py --parse $f | tee $R/gpurun_out/r03/${TAG}_probe.txt
rm -rf $R/gpurun_out/r03/${TAG}_probe
