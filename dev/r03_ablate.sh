#!/bin/bash
# round 3: ablation of ncfp_fwd_kernel, kernel durations under rocprofv3 (CTR_NCFP_DBG bits: 2 no layers, 4 no y stores,
# 8 computed ids (no id loads), 16 no prob store, 32 no row fetch, 64 no ranks store)
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/r03
for d in "$@"; do
  echo "== dbg $d"
  CTR_NCFP_DBG=$d bash $R/dev/r03_probe.sh ab_$d | grep "A eval fwd   ncfp_fwd"
done
