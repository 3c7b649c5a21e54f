#!/bin/bash
# usage: tools/try_variants.sh [mode] -> per-variant DDC kernel time from the library's own event
# timing (swaps the product .so on the GPU box only; tools/variants/*.so are built by hand)
R=$GRAFT_REPO_ROOT
cp $R/webradio_amd/lib/libwebradio_amd.so /tmp/orig.so
for v in $R/tools/variants/*.so; do
  cp $v $R/webradio_amd/lib/libwebradio_amd.so
  echo "$(basename $v .so): $(python $R/tools/quick_time.py 256 ${1:-rotate} 2>/dev/null | tr '\n' ' ')"
done
cp /tmp/orig.so $R/webradio_amd/lib/libwebradio_amd.so
