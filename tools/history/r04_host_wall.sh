#!/bin/bash
# r04: where a Radio::run() of the drop-in path spends its WALL time (WEBRADIO_WALL=1: every process() bracketed with
# the monotonic clock), byte-format and float sources, audio on time and late.
cd $(dirname $0)/../../tests/cxx
for src in u8 f32; do for late in 0 1; do
  echo "== $src late=$late"
  WEBRADIO_QUIET=1 WEBRADIO_WALL=1 WR_HOST_BENCH_PROFILE=1 WEBRADIO_AUDIO_LATE=$late ./host_bench 256 200 4000000 $src 2>&1 | cut -c1-400
  WEBRADIO_QUIET=1 WEBRADIO_AUDIO_LATE=$late ./host_bench 256 200 4000000 $src 2>&1 | cut -c1-200
done; done
