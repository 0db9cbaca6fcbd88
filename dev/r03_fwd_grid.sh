#!/bin/bash
# forward grid (workgroups per CU) against the rank split, now that the per-sample loop has no atomics
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/r03
cd $R && timeout -k 10 300 python -m pytest tests/test_gpu_models.py -q -m gpu -x -k "unusual_call_orders or table_row" 2>&1 | tail -2
for cfg in "256 50" "512 100" "512 50" "384 100" "256 100"; do
  set -- $cfg
  echo "== fwd wgs $1, rank split $2"
  CTR_NCFP_FWD_WGS=$1 CTR_NCFP_RANK_SPLIT=$2 bash $R/dev/r03_probe.sh fg_$1_$2 | grep "A eval fwd   ncfp_fwd\|C step       ncfp_prep\|C step       ncfp_fwd"
  CTR_NCFP_FWD_WGS=$1 CTR_NCFP_RANK_SPLIT=$2 timeout -k 10 300 python $R/bench.py --no-gather-leg --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   bench', round(d['value']/1e6,1),'M/s', round(d['ms_per_step']*1e3,2),'us')"
done
