#!/bin/bash
# dev: A/B builds of the id-gather fast path (unroll x grid cap) -> dev/build/libctrhip_<tag>.so
cd "$(dirname "$0")/../deeplearningrecommendationsystem_amd/csrc"
for v in "u1g8192:-DCTR_FAST_UNROLL=1 -DCTR_FAST_GRID=8192" "u1g16384:-DCTR_FAST_UNROLL=1 -DCTR_FAST_GRID=16384" \
         "u2g8192:-DCTR_FAST_UNROLL=2 -DCTR_FAST_GRID=8192" "u2g4096:-DCTR_FAST_UNROLL=2 -DCTR_FAST_GRID=4096" \
         "u4g8192:-DCTR_FAST_UNROLL=4 -DCTR_FAST_GRID=8192" "u8g2048:-DCTR_FAST_UNROLL=8 -DCTR_FAST_GRID=2048"; do
  tag=${v%%:*}; flags=${v#*:}
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics $flags -c embed.hip -o ../../dev/build/embed_$tag.o || exit 1
  objs=$(ls build/*.o | grep -v "build/embed.o")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../dev/build/libctrhip_$tag.so $objs ../../dev/build/embed_$tag.o || exit 1
done
ls -la ../../dev/build/
