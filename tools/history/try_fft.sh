#!/bin/bash
# usage: tools/try_fft.sh -> k_fft64k_pass1/2 durations (C3 off 12 resident blocks, HBM-cold) under every tools/variants/*.so
R=$GRAFT_REPO_ROOT
cp $R/webradio_amd/lib/libwebradio_amd.so /tmp/orig.so
for v in $R/tools/variants/*.so; do
  cp $v $R/webradio_amd/lib/libwebradio_amd.so
  echo "== $(basename $v .so)"
  bash $R/tools/kstats.sh python $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --settle-ms 30 | grep fft
done
cp /tmp/orig.so $R/webradio_amd/lib/libwebradio_amd.so
