#!/bin/bash
# does the forward's time depend on the id PATTERN (loaded from memory in every case)?
R=${GRAFT_REPO_ROOT:-/root/repo}
for pat in ${@:-random regular sorted}; do
  echo "== ids $pat"; PROBE_IDS=$pat bash $R/dev/r03_probe.sh pat_$pat | grep "ncfp_fwd\|C step       ncfp_bwd\|C step       ncfp_segsum"
done
