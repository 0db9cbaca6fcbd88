#!/bin/bash
# round 3: grid sizes of the table-row NeuralCF kernels (CTR_NCFP_FWD_WGS / CTR_NCFP_BWD_WGS), probe under rocprofv3
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/r03
cd $R && timeout -k 10 200 python -m pytest tests/test_gpu_models.py -m gpu -x -q -k "table_row" 2>&1 | tail -3
for cfg in "256 256" "512 256" "768 512"; do
  set -- $cfg
  echo "== fwd wgs $1, bwd wgs $2"
  CTR_NCFP_FWD_WGS=$1 CTR_NCFP_BWD_WGS=$2 bash $R/dev/r03_probe.sh g_$1_$2 | grep "C step"
done
