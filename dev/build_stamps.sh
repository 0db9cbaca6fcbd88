#!/bin/bash
# second build of the library with cycle stamps in ncf_proj.hip (and mlp_mfma16.hip): dev/timing/libctrhip_stamps.so
set -e
R=$(cd $(dirname $0)/.. && pwd)
C=$R/deeplearningrecommendationsystem_amd/csrc
make -C $C -j8 > /dev/null
mkdir -p $C/build_st $R/dev/timing
for f in ncf_proj mlp_mfma16; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -DCTR_STAMPS -c $C/$f.hip -o $C/build_st/$f.o
done
objs=$(ls $C/build/*.o | grep -v "ncf_proj.o\|mlp_mfma16.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/dev/timing/libctrhip_stamps.so $objs $C/build_st/ncf_proj.o $C/build_st/mlp_mfma16.o
echo built $R/dev/timing/libctrhip_stamps.so
