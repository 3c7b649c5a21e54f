#!/bin/bash
# r04: the drop-in path (tests/cxx/host_bench: Radio::run, 256 receivers, block in HOST memory) -- sparse staging
# (WEBRADIO_SPARSE, default on) against whole-block staging in one piece and in parts (WEBRADIO_PIECES), audio on time and late
cd $(dirname $0)/../../tests/cxx
for src in u8 f32; do
  for cfg in "WEBRADIO_SPARSE=1" "WEBRADIO_SPARSE=0 WEBRADIO_PIECES=1" "WEBRADIO_SPARSE=0 WEBRADIO_PIECES=2" "WEBRADIO_SPARSE=1 WEBRADIO_AUDIO_LATE=1" "WEBRADIO_SPARSE=0 WEBRADIO_AUDIO_LATE=1"; do
    echo "== $src $cfg"
    env WEBRADIO_QUIET=1 $cfg ./host_bench 256 200 4000000 $src 2>&1 | cut -c1-330
  done
done
