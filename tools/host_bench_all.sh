#!/bin/bash
# usage: tools/host_bench_all.sh [blocks=40] -> tests/cxx/host_bench, f32 and u8 sources, audio on time / one / two blocks late
R=$GRAFT_REPO_ROOT
cd $R
export LD_LIBRARY_PATH=$R/webradio_amd/lib:$R/tests/cxx:$LD_LIBRARY_PATH
B=${1:-40}
for src in f32 u8; do
  for late in 0 1 2; do
    WEBRADIO_QUIET=1 WEBRADIO_AUDIO_LATE=$late tests/cxx/host_bench 256 $B 4000000 $src 2>&1 | tail -2
  done
done
