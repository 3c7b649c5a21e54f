#!/bin/bash
# usage: tools/try_variants_k.sh [mode] -> per-variant kernel medians from rocprofv3 kernel trace
R=$GRAFT_REPO_ROOT
cp $R/webradio_amd/lib/libwebradio_amd.so /tmp/orig.so
for v in $R/tools/variants/*.so; do
  cp $v $R/webradio_amd/lib/libwebradio_amd.so
  echo "== $(basename $v .so)"; bash $R/tools/kstats.sh python $R/tools/quick_time.py 256 ${1:-rotate} | cut -c1-75
done
cp /tmp/orig.so $R/webradio_amd/lib/libwebradio_amd.so
