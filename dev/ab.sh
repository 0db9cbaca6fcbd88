for v in nl ns nls u8 u2 u8nls; do echo "== $v"; CTRHIP_LIB=$PWD/dev/variants/lib_$v.so python dev/gather_bench.py 2>&1 | grep embed_fwd; done
echo "== base"; python dev/gather_bench.py 2>&1 | grep embed_fwd
