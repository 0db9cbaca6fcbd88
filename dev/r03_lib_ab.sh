#!/bin/bash
# A/B of two builds of the library on one box: the in-tree one against dev/timing/libctrhip_old.so, alternating
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/r03
for rep in 1 2; do
  echo "== new ($rep)"; bash $R/dev/r03_probe.sh ab_new$rep | grep "C step"
  echo "== old ($rep)"; CTRHIP_LIB=$R/dev/timing/libctrhip_old.so bash $R/dev/r03_probe.sh ab_old$rep | grep "C step"
done
